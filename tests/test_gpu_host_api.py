"""The zlib-compatible host library (libzamd_z.so) driven the way /root/reference/qcsrc/example.c drives zlib:
one-shot compress/uncompress, streaming with tiny buffers, flushes, error returns.  Output bytes are compared with the
CPU oracle's mode-B stream (= the reference's bytes for the same chunking)."""
import hashlib

import pytest

pytestmark = pytest.mark.gpu

from oracle import cases, corpus_py as CP, oracle_py as O  # noqa: E402
import zhost as Z  # noqa: E402


def chunks_of(data, flush_points):
    """oracle stream for input cut at the given flush points and, between them, every 64 KiB"""
    pieces, lo = [], 0
    for hi in list(flush_points) + [len(data)]:
        k = lo
        while True:
            e = min(k + 65536, hi)
            pieces.append(data[k:e])
            k = e
            if k >= hi:
                break
        lo = hi
    return pieces


def test_version_and_errors():
    L = Z.lib()
    assert L.zlibVersion() == b"1.2.3"
    assert L.zError(-3) == b"data error" and L.zError(1) == b"stream end"
    import ctypes as C
    s = Z.ZStream()
    assert L.deflateInit_(C.byref(s), 6, b"2.0", C.sizeof(Z.ZStream)) == Z.Z_VERSION_ERROR      # deflate.c:236-239
    assert L.deflateInit_(C.byref(s), 6, b"1.2.3", 100) == Z.Z_VERSION_ERROR
    assert L.deflateInit_(C.byref(s), 10, b"1.2.3", C.sizeof(Z.ZStream)) == Z.Z_STREAM_ERROR
    assert L.deflateInit2_(C.byref(s), 6, 8, 7, 8, 0, b"1.2.3", C.sizeof(Z.ZStream)) == Z.Z_STREAM_ERROR   # deflate.c:258-261: windowBits 8..15,
    assert L.deflateInit2_(C.byref(s), 6, 8, 15, 10, 0, b"1.2.3", C.sizeof(Z.ZStream)) == Z.Z_STREAM_ERROR # memLevel 1..9 (all served: tests/test_gpu_geometry.py)
    assert L.deflateInit2_(C.byref(s), 6, 8, 15, 0, 0, b"1.2.3", C.sizeof(Z.ZStream)) == Z.Z_STREAM_ERROR
    assert L.deflateInit2_(C.byref(s), 6, 8, 14, 9, 0, b"1.2.3", C.sizeof(Z.ZStream)) == Z.Z_OK
    assert L.deflateEnd(C.byref(s)) == Z.Z_OK
    assert L.deflateInit2_(C.byref(s), 6, 8, 31, 8, 0, b"1.2.3", C.sizeof(Z.ZStream)) == Z.Z_OK            # gzip wrapper (tests/test_gpu_gzip.py)
    assert L.deflateEnd(C.byref(s)) == Z.Z_OK
    assert L.deflateInit_(C.byref(s), 6, b"1.2.3", C.sizeof(Z.ZStream)) == Z.Z_OK
    out = C.create_string_buffer(16)
    s.next_out = C.addressof(out); s.avail_out = 0
    assert L.deflate(C.byref(s), Z.Z_NO_FLUSH) == Z.Z_BUF_ERROR                                  # deflate.c:570
    s.avail_out = 16
    assert L.deflate(C.byref(s), 7) == Z.Z_STREAM_ERROR
    assert L.deflateEnd(C.byref(s)) == Z.Z_OK                                                    # nothing written yet


@pytest.mark.parametrize("level", [-1, 0, 1, 6, 9])
def test_compress2_uncompress_roundtrip(level, golden):
    """BASELINE config 1 (1 MiB of 'hello, hello! ') through compress2()/uncompress(), and the 14-byte example.c string."""
    kat = golden("kat.json")
    rc, z = Z.compress2(cases.HELLO, level)
    assert rc == 0 and z.hex() == kat["hello"][str(level)]  # a single chunk: identical to the reference's compress2
    rc, back = Z.uncompress(z, 100)
    assert rc == 0 and back == cases.HELLO
    big = cases.hello_1mib()
    rc, z = Z.compress2(big, level)
    # ONE continuous stream since round 4: the bytes of the reference's compress2() (kat.json: 2071 bytes at levels 6 and 9, 5632 at level 1)
    assert rc == 0 and z == O.cont_stream(big, level)
    if level != 0:
        ref = kat["hello_1mib"][str(6 if level == -1 else level)]
        assert len(z) == ref["compress2_len"] and hashlib.sha256(z).hexdigest() == ref["compress2_sha256"]
    assert Z.lib().compressBound(len(big)) == kat["compressBound"]["1048576"] == 1048907  # compress.c:75-79
    rc, back = Z.uncompress(z, len(big))
    assert rc == 0 and back == big


def test_compress2_buffer_too_small_and_uncompress_errors():
    data = cases.make("rand", 200000, 4)
    rc, _ = Z.compress2(data, 6, cap=1000)
    assert rc == Z.Z_BUF_ERROR                                       # compress.c:50-53
    rc, z = Z.compress2(data, 6)
    assert rc == 0
    assert Z.uncompress(z, len(data) - 1)[0] == Z.Z_BUF_ERROR         # output too small
    assert Z.uncompress(z[:-7], len(data))[0] == Z.Z_DATA_ERROR      # input ends early (uncompr.c:53-55)
    bad = bytearray(z); bad[-1] ^= 1
    assert Z.uncompress(bytes(bad), len(data))[0] == Z.Z_DATA_ERROR  # incorrect data check
    bad = bytearray(z); bad[0] ^= 0x10
    assert Z.uncompress(bytes(bad), len(data))[0] == Z.Z_DATA_ERROR  # incorrect header check


def test_streaming_deflate_tiny_buffers():
    """example.c:169-205 feeds and drains one byte at a time; the stream must not depend on the buffering."""
    data = CP.chunks(CP.KIND_SILESIA, 11, 3).tobytes()[:150000]
    want = O.cont_stream(data, 6)
    z, codes, info = Z.deflate_stream(data, 6, [(len(data), Z.Z_FINISH)], in_step=None, out_step=7)
    assert z == want and codes[-1] == Z.Z_STREAM_END
    assert info["total_in"] == len(data) and info["total_out"] == len(want) and info["adler"] == O.adler32(data)
    z, codes, info = Z.deflate_stream(data[:70000], 1, [(70000, Z.Z_FINISH)], in_step=1000, out_step=None)
    assert z == O.cont_stream(data[:70000], 1) and info["end_rc"] == Z.Z_OK
    # input that ends exactly on a chunk boundary, fed in slices and finished with an empty call (minizip's zipWriteInFileInZip /
    # zipCloseFileInZip, qcsrc/zip.c:969-1062): the last chunk carries the final bit, as in the one-call stream
    for n in (65536, 131072):
        for lvl in (1, 6):
            z, codes, info = Z.deflate_stream(data[:n], lvl, [(n, Z.Z_NO_FLUSH), (0, Z.Z_FINISH)], in_step=16384, out_step=16384)
            assert z == O.cont_stream(data[:n], lvl), (n, lvl)
    small = cases.HELLO
    z, codes, info = Z.deflate_stream(small, 9, [(len(small), Z.Z_FINISH)], in_step=1, out_step=1)
    assert z == O.deflate_stream(small, 9)
    assert info["data_type"] == 0  # "hello, hello!\0" holds a NUL: Z_BINARY (trees.c:1126-1139)


def test_full_flush_points_become_chunk_boundaries():
    data = cases.make("text", 200000, 8)
    cuts = [3, 70000, 70001, 150000]
    plan = [(3, Z.Z_FULL_FLUSH), (69997, Z.Z_FULL_FLUSH), (1, Z.Z_SYNC_FLUSH), (79999, Z.Z_FULL_FLUSH), (50000, Z.Z_FINISH)]
    z, codes, info = Z.deflate_stream(data, 6, plan)
    # what the reference writes for these calls: Z_FULL_FLUSH forgets the history (deflate.c:817), Z_SYNC_FLUSH keeps it (the window stays)
    want = O.cont_stream(data, 6, [(3, Z.Z_FULL_FLUSH), (70000, Z.Z_FULL_FLUSH), (70001, Z.Z_SYNC_FLUSH), (150000, Z.Z_FULL_FLUSH)])
    assert z == want
    # and the library reads its own flushed stream back, fed one byte at a time
    rc, back, msg, adler = Z.inflate_stream(z[:5000], 10, in_step=1)  # incomplete prefix: every byte is accepted, then a call
    assert rc == Z.Z_BUF_ERROR and back == data[:3]                   # without input or progress gets Z_BUF_ERROR (inflate.c:1150-1151);
                                                                      # what the complete segments in front hold has come out (here: the first flush point)
    rc, back, msg, adler = Z.inflate_stream(z, len(data), in_step=4096)
    assert rc == Z.Z_STREAM_END and back == data and adler == O.adler32(data)


def test_duplicate_flush_and_finish_rules():
    import ctypes as C
    L = Z.lib()
    s = Z.ZStream()
    assert L.deflateInit_(C.byref(s), 6, b"1.2.3", C.sizeof(Z.ZStream)) == Z.Z_OK
    src = C.create_string_buffer(b"abcabcabc")
    out = C.create_string_buffer(1000)
    s.next_in = C.addressof(src); s.avail_in = 9; s.next_out = C.addressof(out); s.avail_out = 1000
    assert L.deflate(C.byref(s), Z.Z_FULL_FLUSH) == Z.Z_OK
    assert L.deflate(C.byref(s), Z.Z_FULL_FLUSH) == Z.Z_BUF_ERROR    # same flush again, no input (deflate.c:774-777)
    assert L.deflateEnd(C.byref(s)) == Z.Z_DATA_ERROR                # stream abandoned while busy (deflate.c:886)
    assert L.deflateInit_(C.byref(s), 6, b"1.2.3", C.sizeof(Z.ZStream)) == Z.Z_OK
    s.next_in = C.addressof(src); s.avail_in = 9; s.next_out = C.addressof(out); s.avail_out = 1000
    assert L.deflate(C.byref(s), Z.Z_FINISH) == Z.Z_STREAM_END
    assert L.deflate(C.byref(s), Z.Z_FINISH) == Z.Z_STREAM_END       # harmless repeat (deflate.c:770-773)
    s.avail_in = 3; s.next_in = C.addressof(src)
    assert L.deflate(C.byref(s), Z.Z_FINISH) == Z.Z_BUF_ERROR        # input after the end (deflate.c:780-782)
    assert L.deflate(C.byref(s), Z.Z_NO_FLUSH) == Z.Z_STREAM_ERROR
    assert L.deflateEnd(C.byref(s)) == Z.Z_OK
    assert out.raw[: s.total_out if s.total_out else 0] is not None


def test_inflate_one_byte_at_a_time_like_example_c():
    """example.c:210-243: avail_in = avail_out = 1 until Z_STREAM_END."""
    z = O.deflate_stream(cases.HELLO, 6)
    rc, back, msg, adler = Z.inflate_stream(z, 100, in_step=1, out_step=1)
    assert rc == Z.Z_STREAM_END and back == cases.HELLO
    # a raw stream (windowBits -15), as zip.c/unzip.c use it
    zr, _, _ = Z.deflate_stream(b"raw deflate for zip members " * 50, 6, [(1400, Z.Z_FINISH)], window_bits=-15)
    assert zr == O.deflate_stream(b"raw deflate for zip members " * 50, 6)[2:-4]
    rc, back, msg, adler = Z.inflate_stream(zr, 2000, window_bits=-15, flush=Z.Z_FINISH)
    assert rc == Z.Z_STREAM_END and back == b"raw deflate for zip members " * 50


def test_foreign_single_segment_stream():
    """A plain zlib stream (no flush markers) that decodes to <= 64 KiB is one segment for the engine."""
    import zlib
    data = cases.make("text", 60000, 2)
    z = zlib.compress(data, 6)
    rc, back = Z.uncompress(z, len(data))
    assert rc == 0 and back == data
    # stored data that contains the marker pattern 00 00 FF FF inside a chunk must not be split there
    tricky = (b"\x00\x00\xff\xff" * 100 + bytes(range(256))) * 20
    rc, z = Z.compress2(tricky, 0)
    assert rc == 0
    rc, back = Z.uncompress(z, len(tricky))
    assert rc == 0 and back == tricky


def test_foreign_streams_of_any_size():
    """Streams of another producer (here the system zlib): matches reach across 64 KiB, no full-flush markers, or markers of
    Z_SYNC_FLUSH that do not reset the window.  They do not split into independent segments; one workgroup decodes them from
    end to end (ZGPU_WHOLE_STREAM) and inflate()/uncompress() deliver the same bytes as for any other stream."""
    import zlib
    data = cases.make("mix", 700000, 9)
    for level in (1, 6, 9):
        z = zlib.compress(data, level)
        rc, back = Z.uncompress(z, len(data))
        assert rc == 0 and back == data, level
    # sync-flush markers in the middle (the window is kept: the pieces depend on each other)
    co = zlib.compressobj(6)
    z = co.compress(data[:100000]) + co.flush(zlib.Z_SYNC_FLUSH) + co.compress(data[100000:300000]) + co.flush(zlib.Z_SYNC_FLUSH) + co.compress(data[300000:]) + co.flush()
    rc, back = Z.uncompress(z, len(data))
    assert rc == 0 and back == data
    # a gzip member, fed in pieces, output drained in pieces
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    gz = co.compress(data) + co.flush()
    rc, back, _, _ = Z.inflate_stream(gz, len(data), in_step=50000, out_step=30000, window_bits=47)
    assert rc == Z.Z_STREAM_END and back == data
    # destination too small: Z_BUF_ERROR like the reference's uncompress (uncompr.c:53-57)
    rc, _ = Z.uncompress(zlib.compress(data, 6), len(data) - 1)
    assert rc == Z.Z_BUF_ERROR
    # damage in the middle of the stream: a data error, as the system zlib finds one too
    bad = bytearray(zlib.compress(data, 6)); bad[len(bad) // 2] ^= 0x55
    with pytest.raises(zlib.error):
        zlib.decompress(bytes(bad))
    rc, _ = Z.uncompress(bytes(bad), len(data))
    assert rc == Z.Z_DATA_ERROR


def test_level0_block_structure():
    """deflate_stored's cuts (deflate.c:1390-1439) for chunk lengths around MAX_DIST and the 65531-byte block limit, finished or
    flushed: the host library's framing against the oracle (which is pinned to the reference for level 0)."""
    from oracle import cases, oracle_py as O
    for n in (0, 1, 1000, 32505, 32506, 32507, 40000, 65531, 65532, 65535, 65536, 65536 + 32506, 2 * 65536 + 40000):
        data = cases.make("text", n, 3)
        z, codes, info = Z.deflate_stream(data, 0, [(n, Z.Z_FINISH)])
        assert z == O.cont_stream(data, 0), n
        if n:
            z2, codes, info = Z.deflate_stream(data, 0, [(n, Z.Z_FULL_FLUSH), (0, Z.Z_FINISH)])
            assert z2 == O.cont_stream(data, 0, [(n, Z.Z_FULL_FLUSH)]), n
            z3, codes, info = Z.deflate_stream(data, 0, [(n, Z.Z_FINISH)], in_step=20000)  # (level 0 cuts its blocks by what each call brings)
            assert z3 == O.cont_stream(data, 0, [(q, Z.Z_NO_FLUSH) for q in range(20000, n, 20000)]), n


def test_host_buffers_in_batches_with_copies_under_the_kernels():
    """zgpu_deflate_host over more than one batch (ZGPU_BATCH_CHUNKS makes batches small): the input arrives batch by batch on the copy stream and a
    second host thread takes finished batches' bytes home while later ones are compressed.  The stream must be the one-batch device stream."""
    import hashlib
    import os
    import torch
    import zlib_amd
    from zlib_amd import gpu
    e = zlib_amd.Engine(0)
    n = 8192 + 700  # more than the 8192 chunks below which the host entry point does not bother to overlap
    src = torch.empty(n * 65536, dtype=torch.uint8, device="cuda")
    e.corpus_fill_device(0, 0x5EED5117, 3, n, src.data_ptr())
    cap = e.L.zgpu_deflate_bound(n * 65536, 65536)
    dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
    res = e.deflate_device(src.data_ptr(), n * 65536, 6, dst.data_ptr(), cap)
    want = hashlib.sha256(dst[: res.out_bytes].cpu().numpy().tobytes()).hexdigest()
    host = src.cpu().numpy()
    old = os.environ.get("ZGPU_BATCH_CHUNKS")
    try:
        for batch in ("1500", None):
            if batch is None:
                os.environ.pop("ZGPU_BATCH_CHUNKS", None)
            else:
                os.environ["ZGPU_BATCH_CHUNKS"] = batch
            z = e.deflate_host(host, 6)
            assert len(z) == res.out_bytes and hashlib.sha256(z).hexdigest() == want, batch
    finally:
        if old is None:
            os.environ.pop("ZGPU_BATCH_CHUNKS", None)
        else:
            os.environ["ZGPU_BATCH_CHUNKS"] = old
    e.close()


def test_zamd_devices_fans_one_call_out_over_engines():
    """ZAMD_DEVICES names the GPUs one deflate()/compress2() call may use (SURVEY.md 8e through the C boundary: contiguous chunk ranges, one engine
    and one host thread per device, the ranges' streams laid end to end).  Here both names are device 0 -- two engines side by side on the one
    GPU of the test box; the stream must be the one-engine stream, byte for byte, zlib and gzip wrapper."""
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = r"""
import ctypes as C, hashlib, os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import zhost as Z
from oracle import corpus_py as CP
data = CP.chunks(0, 7, 1200).tobytes()[: 1200 * 65536 - 12345]   # ~75 MiB: above the 64 MiB from which a call fans out
rc, z = Z.compress2(data, 6)
assert rc == 0
print("ZLIB", hashlib.sha256(z).hexdigest(), len(z))
zs, codes, info = Z.deflate_stream(data, 4, [(len(data), Z.Z_FINISH)], window_bits=31)
print("GZIP", hashlib.sha256(zs).hexdigest(), len(zs), info["adler"])
"""
    outs = []
    for devs in (None, "0,0", "0,0,0"):
        env = dict(os.environ)
        env.pop("ZAMD_DEVICES", None)
        if devs:
            env["ZAMD_DEVICES"] = devs
        p = subprocess.run([sys.executable, "-c", child % (ROOT, ROOT)], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append([ln for ln in p.stdout.splitlines() if ln.startswith(("ZLIB", "GZIP"))])
    assert len(outs[0]) == 2 and outs[1] == outs[0] and outs[2] == outs[0], outs


def test_threads_share_a_pool_of_engines():
    """ZAMD_ENGINES=3: calls of different threads run side by side on engines of their own (with the default they take turns).  Six threads compress and
    decompress different buffers at once; every result must be what the same call gives alone."""
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = r"""
import hashlib, os, sys, threading
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import zhost as Z
from oracle import corpus_py as CP
bufs = [CP.chunks(k & 1, 50 * k, 200 + 13 * k).tobytes()[: (200 + 13 * k) * 65536 - 1000 * k] for k in range(6)]
alone = []
for b in bufs:
    rc, z = Z.compress2(b, 6); assert rc == 0
    alone.append(hashlib.sha256(z).hexdigest())
got, back = [None] * 6, [None] * 6
def work(i):
    for _ in range(3):
        rc, z = Z.compress2(bufs[i], 6); assert rc == 0
        got[i] = hashlib.sha256(z).hexdigest()
        rc, out = Z.uncompress(z, len(bufs[i])); assert rc == 0
        back[i] = out == bufs[i]
th = [threading.Thread(target=work, args=(i,)) for i in range(6)]
[t.start() for t in th]; [t.join() for t in th]
assert got == alone and all(back), (got, alone, back)
print("POOL OK")
"""
    for n in ("1", "3"):
        env = dict(os.environ, ZAMD_ENGINES=n)
        p = subprocess.run([sys.executable, "-c", child % (ROOT, ROOT)], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0 and "POOL OK" in p.stdout, p.stderr[-2000:]
