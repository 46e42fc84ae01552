"""GPU tests of round 3's fixes in the streaming paths: no host copy beyond the caller's
buffer, inflate() hands back only what it took in, the 32 KiB window is carried from call to call (streams of another zlib that reach back across
sync flushes decode piece by piece), readers drain what is decoded before they feed more."""
import ctypes as C
import gzip
import os
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import cases, corpus_py as CP  # noqa: E402
import zhost as Z  # noqa: E402


@pytest.fixture(scope="module")
def eng():
    import zlib_amd
    e = zlib_amd.Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def L():
    return Z.lib()


def test_inflate_host_never_writes_behind_the_callers_buffer(eng):
    """4100 chunks (the batch-by-batch path to the host), a length that is no multiple of the chunk size: the caller's buffer is exactly as long as
    the data, the bytes behind it stay untouched; a buffer that is too small is an error and is not overrun either."""
    from zlib_amd import gpu
    n = 4100 * 65536 - 12345
    data = np.tile(CP.chunks(CP.KIND_LOGTEXT, 5, 64), 65)[:n]
    z, offs = eng.deflate_host(data, 4, flags=gpu.F_FINAL, want_offsets=True)  # a raw body, the last chunk with the final block
    arr = np.frombuffer(z, dtype=np.uint8)
    offs = np.ascontiguousarray(offs, dtype=np.uint64)
    guard = 1 << 20
    for cap in (n, n - 70000):
        out = np.full(n + guard, 0xA5, dtype=np.uint8)
        res = gpu.InflateResult()
        rc = eng.L.zgpu_inflate_host(eng.h, arr.ctypes.data, arr.size, offs.ctypes.data, len(offs) - 1, 65536, out.ctypes.data, cap, C.byref(res))
        assert (out[cap:] == 0xA5).all(), "bytes behind the caller's %d-byte buffer were written" % cap
        if cap == n:
            assert rc == 0 and res.out_bytes == n and np.array_equal(out[:n], data)
        else:
            assert rc == -5  # Z_BUF_ERROR


def _inflate_loop(L, z, wbits, in_step, out_step, total_cap, trailing=b""):
    """Feed z + trailing in pieces of in_step with fresh next_in every call, drain through a small buffer.  Returns (rc, bytes, bytes left unread,
    [(total_in, total_out) after every call])."""
    s = Z.ZStream()
    assert L.inflateInit2_(C.byref(s), wbits, b"1.2.3", C.sizeof(Z.ZStream)) == Z.Z_OK
    blob = z + trailing
    src = C.create_string_buffer(blob, max(len(blob), 1))
    obuf = C.create_string_buffer(out_step)
    got, ipos, rc, trace = bytearray(), 0, Z.Z_OK, []
    for _ in range(10_000_000):
        step = min(in_step, len(blob) - ipos)
        s.next_in = C.addressof(src) + ipos
        s.avail_in = step
        s.next_out = C.addressof(obuf)
        s.avail_out = out_step
        rc = L.inflate(C.byref(s), Z.Z_NO_FLUSH)
        assert C.addressof(src) + ipos <= (s.next_in or 0) <= C.addressof(src) + ipos + step, "next_in left the piece it was given"
        assert (s.next_in or 0) - (C.addressof(src) + ipos) == step - s.avail_in
        ipos += step - s.avail_in
        got += obuf.raw[: out_step - s.avail_out]
        trace.append((s.total_in, s.total_out))
        assert len(got) <= total_cap
        if rc != Z.Z_OK:
            break
    L.inflateEnd(C.byref(s))
    return rc, bytes(got), len(blob) - ipos, trace


def test_end_reached_in_an_earlier_call_leaves_next_in_alone(L):
    """A stream whose end is reached while output is still undelivered: the calls that drain it bring fresh input (the bytes behind the stream), which
    must stay with the caller -- next_in never moves backwards out of the piece, and exactly the trailing bytes are left."""
    data = cases.make("mix", 3 * 65536 + 777, 5)
    z = zlib.compress(data, 6)
    trailing = bytes(range(256)) * 40
    for in_step, out_step in ((len(z) + len(trailing), 1000), (len(z) // 3 + 1, 4096), (100000, 70000)):
        rc, got, left, trace = _inflate_loop(L, z, 15, in_step, out_step, len(data), trailing)
        assert rc == Z.Z_STREAM_END and got == data
        assert trace[-1] == (len(z), len(data))


def test_sync_flushed_stream_of_another_zlib_in_pieces(L):
    """The system zlib with Z_SYNC_FLUSH every 20 KiB: matches reach back across the markers.  Fed in pieces the window has to come along."""
    data = CP.chunks(CP.KIND_LOGTEXT, 77, 24).tobytes()[: 24 * 65536 - 999]
    for wbits, trailing in ((15, b""), (-15, b""), (31, b"tail")):
        co = zlib.compressobj(6, zlib.DEFLATED, wbits)
        z = b"".join(co.compress(data[i: i + 20000]) + co.flush(zlib.Z_SYNC_FLUSH) for i in range(0, len(data), 20000)) + co.flush()
        for in_step, out_step in ((8192, 65536), (50000, 3000), (len(z), 1 << 20)):
            rc, got, left, trace = _inflate_loop(L, z, wbits, in_step, out_step, len(data), trailing)
            assert rc == Z.Z_STREAM_END, (wbits, in_step, rc)
            assert got == data and left == len(trailing)
            assert all(a[0] <= b[0] and a[1] <= b[1] for a, b in zip(trace, trace[1:]))


def test_large_foreign_stream_in_slices_is_delivered_as_it_arrives(L):
    """128 MiB through the system zlib as ONE stream (no flush points), fed in 1 MiB slices with a 1 MiB output buffer: the bytes are the original's,
    total_in / total_out only grow, and output arrives long before the input has ended (the library holds a bounded backlog, not the stream)."""
    data = np.tile(CP.chunks(CP.KIND_SILESIA, 900, 256), 8).tobytes()
    z = zlib.compress(data, 6)
    s = Z.ZStream()
    assert L.inflateInit2_(C.byref(s), 15, b"1.2.3", C.sizeof(Z.ZStream)) == Z.Z_OK
    src = C.create_string_buffer(z, len(z))
    step = 1 << 20
    obuf = C.create_string_buffer(step)
    ipos, opos, rc, backlog_max, half_out = 0, 0, Z.Z_OK, 0, None
    prev = (0, 0)
    view = memoryview(data)
    while rc == Z.Z_OK:
        n = min(step, len(z) - ipos)
        s.next_in = C.addressof(src) + ipos
        s.avail_in = n
        s.next_out = C.addressof(obuf)
        s.avail_out = step
        rc = L.inflate(C.byref(s), Z.Z_NO_FLUSH)
        ipos += n - s.avail_in
        k = step - s.avail_out
        assert obuf.raw[:k] == view[opos: opos + k], "bytes differ at %d" % opos
        opos += k
        assert (s.total_in, s.total_out) >= prev and s.total_in == ipos and s.total_out == opos
        prev = (s.total_in, s.total_out)
        if half_out is None and ipos >= len(z) // 2:
            half_out = opos
    L.inflateEnd(C.byref(s))
    assert rc == Z.Z_STREAM_END and opos == len(data)
    assert half_out is not None and half_out >= len(data) // 4, "with half of the input in, only %d of %d bytes had come out" % (half_out, len(data))


def test_stream_taken_up_at_a_bit_offset_when_the_pieces_decline(L):
    """a stream without flush points is taken up again inside a byte; when the decode in pieces says "not this way" there for a harmless
    reason (no scratch room, too many repairs of the chain, the resolve flag -- forced here with ZGPU_SPEC_DECLINE_AT_BIT) the one-workgroup decoder's result on
    the shifted copy stands: the stream must come out whole, not end in Z_DATA_ERROR."""
    import os
    data = CP.chunks(CP.KIND_SILESIA, 31, 96).tobytes()[: 96 * 65536 - 4321]
    z = zlib.compress(data, 6)
    trailing = b"behind the stream"
    os.environ["ZGPU_SPEC_DECLINE_AT_BIT"] = "1"
    try:
        for in_step, out_step in ((700000, 1 << 20), (400000, 300000)):
            rc, got, left, trace = _inflate_loop(L, z, 15, in_step, out_step, len(data), trailing)
            assert rc == Z.Z_STREAM_END and got == data and left == len(trailing), (in_step, rc, len(got))
    finally:
        del os.environ["ZGPU_SPEC_DECLINE_AT_BIT"]


def test_gzread_two_members_larger_than_the_file_buffer_small_reads(L, tmp_path):
    """Two gzip members, each larger than the reader's 1 MiB file buffer, one of them sync-flushed by the system zlib, read 3000 bytes at a time."""
    a = CP.chunks(CP.KIND_SILESIA, 40, 40).tobytes()
    b = CP.chunks(CP.KIND_LOGTEXT, 41, 90).tobytes()[:-5]
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    zb = b"".join(co.compress(b[i: i + 300000]) + co.flush(zlib.Z_SYNC_FLUSH) for i in range(0, len(b), 300000)) + co.flush()
    path = os.path.join(str(tmp_path), "two.gz")
    with open(path, "wb") as f:
        f.write(gzip.compress(a, 6) + zb)
    L.gzopen.restype = C.c_void_p
    L.gzopen.argtypes = [C.c_char_p, C.c_char_p]
    L.gzread.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
    L.gzclose.argtypes = [C.c_void_p]
    g = L.gzopen(path.encode(), b"rb")
    assert g
    buf = C.create_string_buffer(3000)
    got = bytearray()
    while True:
        k = L.gzread(g, buf, 3000)
        assert k >= 0
        if k == 0:
            break
        got += buf.raw[:k]
    assert L.gzclose(g) == 0
    assert bytes(got) == a + b
