"""The API rows added in round 2, through libzamd_z.so on the GPU, against vectors of the compiled reference (tests/golden/api_kat.json,
oracle/gen_golden_api.py): deflateTune, deflateSetHeader / inflateGetHeader, bytes behind the end of a stream, prefixes of flushed
streams, deflateCopy / inflateCopy, inflateSync, inflatePrime, the gz* file functions, inflateBack, and a hostile segment table."""
import ctypes as C
import hashlib
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import zhost as Z  # noqa: E402
from oracle import cases, oracle_py as O  # noqa: E402


class GzHeader(C.Structure):
    _fields_ = [("text", C.c_int), ("time", C.c_ulong), ("xflags", C.c_int), ("os", C.c_int), ("extra", C.c_void_p), ("extra_len", C.c_uint),
                ("extra_max", C.c_uint), ("name", C.c_void_p), ("name_max", C.c_uint), ("comment", C.c_void_p), ("comm_max", C.c_uint),
                ("hcrc", C.c_int), ("done", C.c_int)]


def h16(b):
    return hashlib.sha256(b).hexdigest()[:16]


@pytest.fixture(scope="module")
def L():
    lib = Z.lib()
    P = C.POINTER(Z.ZStream)
    lib.deflateTune.argtypes = [P, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.deflateSetHeader.argtypes = [P, C.POINTER(GzHeader)]
    lib.inflateGetHeader.argtypes = [P, C.POINTER(GzHeader)]
    lib.deflateCopy.argtypes = [P, P]
    lib.inflateCopy.argtypes = [P, P]
    lib.inflateSync.argtypes = [P]
    lib.inflateSyncPoint.argtypes = [P]
    lib.inflatePrime.argtypes = [P, C.c_int, C.c_int]
    lib.deflatePrime.argtypes = [P, C.c_int, C.c_int]
    lib.get_crc_table.restype = C.POINTER(C.c_ulong)
    lib.crc32.argtypes = [C.c_ulong, C.c_char_p, C.c_uint]
    lib.crc32.restype = C.c_ulong
    for n in ("gzopen", "gzdopen"):
        getattr(lib, n).restype = C.c_void_p
    lib.gzopen.argtypes = [C.c_char_p, C.c_char_p]
    lib.gzread.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
    lib.gzwrite.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
    lib.gzclose.argtypes = [C.c_void_p]
    lib.gzseek.argtypes = [C.c_void_p, C.c_long, C.c_int]
    lib.gzseek.restype = C.c_long
    lib.gztell.argtypes = [C.c_void_p]
    lib.gztell.restype = C.c_long
    for n in ("gzeof", "gzdirect", "gzrewind", "gzgetc"):
        getattr(lib, n).argtypes = [C.c_void_p]
    lib.gzflush.argtypes = [C.c_void_p, C.c_int]
    lib.gzsetparams.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.gzputs.argtypes = [C.c_void_p, C.c_char_p]
    lib.gzgets.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.gzgets.restype = C.c_void_p
    lib.gzputc.argtypes = [C.c_void_p, C.c_int]
    lib.gzungetc.argtypes = [C.c_int, C.c_void_p]
    lib.gzerror.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    lib.gzerror.restype = C.c_char_p
    lib.gzclearerr.argtypes = [C.c_void_p]
    return lib


def one_shot_deflate(L, data, level, wbits, flush, before=None):
    s = Z.ZStream()
    assert L.deflateInit2_(C.byref(s), level, 8, wbits, 8, 0, b"1.2.3", C.sizeof(Z.ZStream)) == Z.Z_OK
    keep = before(s) if before else None
    cap = len(data) + (len(data) >> 8) + 1024
    out = C.create_string_buffer(cap); src = C.create_string_buffer(data, max(len(data), 1))
    s.next_in = C.addressof(src); s.avail_in = len(data); s.next_out = C.addressof(out); s.avail_out = cap
    rc = L.deflate(C.byref(s), flush)
    assert rc == (Z.Z_STREAM_END if flush == Z.Z_FINISH else Z.Z_OK), rc
    z = out.raw[: s.total_out]
    L.deflateEnd(C.byref(s))
    del keep
    return z


def test_deflate_tune_vs_reference(L, golden):
    kat = golden("api_kat.json")["tune"]
    assert len(kat) >= 200
    for key, want in kat.items():
        kind, n, seed, lvl, tune = key.split("/")
        d = cases.make(kind, int(n), int(seed))
        t = tuple(int(x) for x in tune.split("-"))
        for last, w in zip((False, True), want):
            z = one_shot_deflate(L, d, int(lvl[1:]), -15, Z.Z_FINISH if last else Z.Z_FULL_FLUSH, before=lambda s: L.deflateTune(C.byref(s), *t))
            assert h16(z) == w, (key, last)


def test_gzip_header_fields_vs_reference(L, golden):
    import json
    from oracle import gen_golden_api as G
    for row in golden("api_kat.json")["gzhead"]:
        h = G.GZHEADS[row["head"]]
        d = cases.make(row["kind"], row["n"], row["seed"])
        bufs = {k: (C.create_string_buffer(h[k], len(h[k]) + 1) if h[k] is not None else None) for k in ("extra", "name", "comment")}

        def set_head(s, h=h, bufs=bufs):
            g = GzHeader(text=h["text"], time=h["time"], os=h["os"], extra=C.addressof(bufs["extra"]) if bufs["extra"] is not None else None,
                         extra_len=len(h["extra"] or b""), name=C.addressof(bufs["name"]) if bufs["name"] is not None else None,
                         comment=C.addressof(bufs["comment"]) if bufs["comment"] is not None else None, hcrc=h["hcrc"])
            assert L.deflateSetHeader(C.byref(s), C.byref(g)) == Z.Z_OK
            return g
        z = one_shot_deflate(L, d, row["level"], 31, Z.Z_FINISH, before=set_head)
        assert z.hex() == row["member"], (row["head"], row["level"])
        # and back: the fields arrive in a gz_header, the data is the data
        s = Z.ZStream()
        assert L.inflateInit2_(C.byref(s), 47, b"1.2.3", C.sizeof(Z.ZStream)) == Z.Z_OK
        ex, nm, cm = C.create_string_buffer(64), C.create_string_buffer(64), C.create_string_buffer(64)
        g = GzHeader(extra=C.addressof(ex), extra_max=64, name=C.addressof(nm), name_max=64, comment=C.addressof(cm), comm_max=64)
        assert L.inflateGetHeader(C.byref(s), C.byref(g)) == Z.Z_OK and g.done == 0
        src = C.create_string_buffer(z, len(z)); out = C.create_string_buffer(len(d) + 16)
        s.next_in = C.addressof(src); s.avail_in = len(z); s.next_out = C.addressof(out); s.avail_out = len(d) + 16
        assert L.inflate(C.byref(s), Z.Z_FINISH) == Z.Z_STREAM_END and out.raw[: s.total_out] == d
        assert g.done == 1 and g.text == h["text"] and g.time == h["time"] and g.os == h["os"] and g.hcrc == h["hcrc"]
        if h["extra"] is not None:
            assert g.extra_len == len(h["extra"]) and ex.raw[: g.extra_len] == h["extra"]
        else:
            assert not g.extra
        assert (nm.value == h["name"]) if h["name"] is not None else (not g.name)
        assert (cm.value == h["comment"]) if h["comment"] is not None else (not g.comment)
        L.inflateEnd(C.byref(s))


def verdict(L, z, wbits, cap, flush=Z.Z_NO_FLUSH):
    s = Z.ZStream()
    assert L.inflateInit2_(C.byref(s), wbits, b"1.2.3", C.sizeof(Z.ZStream)) == Z.Z_OK
    src = C.create_string_buffer(z, max(len(z), 1)); out = C.create_string_buffer(cap)
    s.next_in = C.addressof(src); s.avail_in = len(z); s.next_out = C.addressof(out); s.avail_out = cap
    rc = L.inflate(C.byref(s), flush)
    res = [rc, int(s.total_in), int(s.total_out), int(s.avail_in), h16(out.raw[: s.total_out])]
    L.inflateEnd(C.byref(s))
    return res


def test_input_behind_the_end_of_a_stream_stays_with_the_caller(L, golden):
    """inflate.c:1114: DONE returns Z_STREAM_END and leaves what follows in next_in / avail_in, total_in counts the stream alone."""
    for row in golden("api_kat.json")["trailing"]:
        assert row["stream"] is not None
        assert verdict(L, bytes.fromhex(row["stream"]), row["wbits"], 200000) == row["verdict"], row["name"]
    # two gzip members, one after the other through inflateReset, and uncompress() with sourceLen beyond the stream
    d = cases.make("text", 30000, 11)
    kat = {r["name"]: r for r in golden("api_kat.json")["trailing"]}
    gz2 = bytes.fromhex(kat["gzip+gzip"]["stream"])
    s = Z.ZStream()
    assert L.inflateInit2_(C.byref(s), 31, b"1.2.3", C.sizeof(Z.ZStream)) == Z.Z_OK
    src = C.create_string_buffer(gz2, len(gz2)); out = C.create_string_buffer(70000)
    s.next_in = C.addressof(src); s.avail_in = len(gz2); s.next_out = C.addressof(out); s.avail_out = 70000
    assert L.inflate(C.byref(s), Z.Z_NO_FLUSH) == Z.Z_STREAM_END and s.avail_in == len(gz2) // 2
    assert L.inflateReset(C.byref(s)) == Z.Z_OK
    assert L.inflate(C.byref(s), Z.Z_NO_FLUSH) == Z.Z_STREAM_END and s.avail_in == 0
    assert out.raw[:60000] == cases.make("text", 100000, 11)[:30000] * 2
    L.inflateEnd(C.byref(s))
    zl = bytes.fromhex(kat["zlib+garbage"]["stream"])
    rc, back = Z.uncompress(zl, 100000)
    assert rc == Z.Z_OK and back == cases.make("text", 100000, 11)


def test_prefix_of_a_flushed_stream_delivers_its_complete_chunks(L, golden):
    d = cases.make("text", 100000, 11)
    parts = [O.deflate_chunk(d[i * 30000:(i + 1) * 30000], 6, i == 3) for i in range(4)]
    zb = O.deflate_stream(b"", 6)[:2] + b"".join(parts)
    for row in golden("api_kat.json")["prefix"]:
        assert row["cut"] == 2 + sum(len(p) for p in parts[: row["chunks"]])
        assert verdict(L, zb[: row["cut"]], 15, 200000) == row["verdict"], row
    # sync-flushed input that arrives piece by piece: every piece's data is out before the next piece is in
    s = Z.ZStream()
    assert L.inflateInit2_(C.byref(s), 15, b"1.2.3", C.sizeof(Z.ZStream)) == Z.Z_OK
    out = C.create_string_buffer(200000); src = C.create_string_buffer(zb, len(zb))
    s.next_out = C.addressof(out); s.avail_out = 200000
    pos = 0
    for k in range(3):
        cut = 2 + sum(len(p) for p in parts[: k + 1])
        s.next_in = C.addressof(src) + pos; s.avail_in = cut - pos; pos = cut
        assert L.inflate(C.byref(s), Z.Z_SYNC_FLUSH) == Z.Z_OK and s.total_out == 30000 * (k + 1) and L.inflateSyncPoint(C.byref(s)) == 1
    L.inflateEnd(C.byref(s))


def test_copies_are_independent_streams(L):
    d = cases.make("mix", 150000, 12)
    s, c = Z.ZStream(), Z.ZStream()
    assert L.deflateInit_(C.byref(s), 6, b"1.2.3", C.sizeof(Z.ZStream)) == Z.Z_OK
    src = C.create_string_buffer(d, len(d)); o1, o2 = C.create_string_buffer(200000), C.create_string_buffer(200000)
    s.next_in = C.addressof(src); s.avail_in = 80000; s.next_out = C.addressof(o1); s.avail_out = 200000
    assert L.deflate(C.byref(s), Z.Z_NO_FLUSH) == Z.Z_OK
    n1 = s.total_out
    assert L.deflateCopy(C.byref(c), C.byref(s)) == Z.Z_OK
    C.memmove(o2, o1, n1)
    c.next_out = C.addressof(o2) + n1; c.avail_out = 200000 - n1
    for st in (s, c):
        st.next_in = C.addressof(src) + 80000; st.avail_in = len(d) - 80000
        assert L.deflate(C.byref(st), Z.Z_FINISH) == Z.Z_STREAM_END
    z1, z2 = o1.raw[: s.total_out], o2.raw[: c.total_out]
    assert z1 == z2 and Z.uncompress(z1, 200000) == (Z.Z_OK, d)
    assert L.deflateEnd(C.byref(s)) == Z.Z_OK and L.deflateEnd(C.byref(c)) == Z.Z_OK
    # inflateCopy in the middle of the input
    s, c = Z.ZStream(), Z.ZStream()
    assert L.inflateInit_(C.byref(s), b"1.2.3", C.sizeof(Z.ZStream)) == Z.Z_OK
    zsrc = C.create_string_buffer(z1, len(z1)); half = len(z1) // 2
    s.next_in = C.addressof(zsrc); s.avail_in = half; s.next_out = C.addressof(o1); s.avail_out = 200000
    assert L.inflate(C.byref(s), Z.Z_NO_FLUSH) == Z.Z_OK
    got = s.total_out
    assert L.inflateCopy(C.byref(c), C.byref(s)) == Z.Z_OK
    C.memmove(o2, o1, got)
    c.next_out = C.addressof(o2) + got
    for st in (s, c):
        st.next_in = C.addressof(zsrc) + half; st.avail_in = len(z1) - half
        assert L.inflate(C.byref(st), Z.Z_FINISH) == Z.Z_STREAM_END
    assert o1.raw[: len(d)] == d and o2.raw[: len(d)] == d
    assert L.inflateEnd(C.byref(s)) == Z.Z_OK and L.inflateEnd(C.byref(c)) == Z.Z_OK


def test_inflate_sync_skips_a_damaged_chunk(L):
    d = cases.make("text", 180000, 13)
    # a stream with full-flush points every 64 KiB (plain compress2() writes ONE continuous stream since round 4: nothing to sync to)
    z = bytearray(Z.deflate_stream(d, 6, [(65536, Z.Z_FULL_FLUSH), (65536, Z.Z_FULL_FLUSH), (len(d) - 131072, Z.Z_FINISH)])[0])
    z[700] ^= 0x55  # inside the first piece
    s = Z.ZStream()
    assert L.inflateInit_(C.byref(s), b"1.2.3", C.sizeof(Z.ZStream)) == Z.Z_OK
    src = C.create_string_buffer(bytes(z), len(z)); out = C.create_string_buffer(200000)
    s.next_in = C.addressof(src); s.avail_in = len(z); s.next_out = C.addressof(out); s.avail_out = 200000
    rc = L.inflate(C.byref(s), Z.Z_FINISH)
    assert rc == Z.Z_DATA_ERROR
    L.inflateEnd(C.byref(s))
    s = Z.ZStream()
    assert L.inflateInit_(C.byref(s), b"1.2.3", C.sizeof(Z.ZStream)) == Z.Z_OK
    s.next_in = C.addressof(src); s.avail_in = 2; s.next_out = C.addressof(out); s.avail_out = 200000
    L.inflate(C.byref(s), Z.Z_NO_FLUSH)                      # the header
    s.avail_in = len(z) - 2
    assert L.inflateSync(C.byref(s)) == Z.Z_OK               # to the first flush point: the start of the second chunk
    assert L.inflate(C.byref(s), Z.Z_FINISH) == Z.Z_DATA_ERROR and s.msg == b"incorrect data check"  # the trailer covers all chunks
    assert out.raw[: s.total_out] == d[65536:]
    L.inflateEnd(C.byref(s))
    s = Z.ZStream()
    assert L.inflateInit_(C.byref(s), b"1.2.3", C.sizeof(Z.ZStream)) == Z.Z_OK
    junk = C.create_string_buffer(b"no marker in here" * 10, 170)
    s.next_in = C.addressof(junk); s.avail_in = 170
    assert L.inflateSync(C.byref(s)) == Z.Z_DATA_ERROR and s.avail_in == 0
    assert L.inflateSync(C.byref(s)) == Z.Z_BUF_ERROR
    L.inflateEnd(C.byref(s))


def test_inflate_prime_reads_a_stream_that_starts_inside_a_byte(L):
    d = cases.make("text", 70000, 14)
    raw = O.deflate_chunk(d[:65536], 6, False) + O.deflate_chunk(d[65536:], 6, True)
    for k in (1, 3, 7):
        n = int.from_bytes(raw, "little")
        value, rest = n & ((1 << k) - 1), n >> k
        shifted = rest.to_bytes(len(raw), "little")
        s = Z.ZStream()
        assert L.inflateInit2_(C.byref(s), -15, b"1.2.3", C.sizeof(Z.ZStream)) == Z.Z_OK
        assert L.inflatePrime(C.byref(s), k, value) == Z.Z_OK
        src = C.create_string_buffer(shifted, len(shifted)); out = C.create_string_buffer(80000)
        s.next_in = C.addressof(src); s.avail_in = len(shifted); s.next_out = C.addressof(out); s.avail_out = 80000
        assert L.inflate(C.byref(s), Z.Z_FINISH) == Z.Z_STREAM_END and out.raw[: s.total_out] == d
        L.inflateEnd(C.byref(s))
    s = Z.ZStream()
    assert L.deflateInit_(C.byref(s), 6, b"1.2.3", C.sizeof(Z.ZStream)) == Z.Z_OK
    assert L.deflatePrime(C.byref(s), 0, 0) == Z.Z_OK and L.deflatePrime(C.byref(s), 17, 5) == Z.Z_STREAM_ERROR  # (tests/test_gpu_prime.py has the rest)
    L.deflateEnd(C.byref(s))


def test_crc_table_and_utilities(L):
    t = L.get_crc_table()
    assert t[0] == 0 and t[1] == 0x77073096 and t[255] == 0x2D02EF8D
    assert L.crc32(0, b"123456789", 9) == 0xCBF43926


def test_gz_file_functions(L, tmp_path):
    d = cases.make("text", 300000, 15)
    path = str(tmp_path / "t.gz").encode()
    f = L.gzopen(path, b"wb6")
    assert f
    src = C.create_string_buffer(d, len(d))
    assert L.gzwrite(f, src, 100000) == 100000 and L.gztell(f) == 100000
    assert L.gzflush(f, Z.Z_FULL_FLUSH) == Z.Z_OK
    assert L.gzsetparams(f, 9, 1) == Z.Z_OK
    assert L.gzwrite(f, C.addressof(src) + 100000, 200000) == 200000
    assert L.gzputs(f, b"tail\n") == 5 and L.gzputc(f, ord("!")) == ord("!")
    assert L.gzseek(f, 7, 1) == 300013  # seven zero bytes
    assert L.gzclose(f) == Z.Z_OK
    import gzip
    whole = d + b"tail\n!" + bytes(7)
    assert gzip.decompress(open(path.decode(), "rb").read()) == whole  # any gzip reader takes the file
    f = L.gzopen(path, b"rb")
    buf = C.create_string_buffer(400000)
    assert L.gzdirect(f) == 0
    assert L.gzread(f, buf, 150000) == 150000 and buf.raw[:150000] == whole[:150000] and L.gztell(f) == 150000 and L.gzeof(f) == 0
    assert L.gzseek(f, -50000, 1) == 100000 and L.gzgetc(f) == whole[100000]
    assert L.gzungetc(whole[100000], f) == whole[100000] and L.gztell(f) == 100000
    assert L.gzread(f, buf, 400000) == len(whole) - 100000 and buf.raw[: len(whole) - 100000] == whole[100000:]
    assert L.gzread(f, buf, 10) == 0 and L.gzeof(f) == 1
    assert L.gzrewind(f) == 0 and L.gzseek(f, 300000, 0) == 300000
    line = C.create_string_buffer(64)
    assert L.gzgets(f, line, 64) and line.value == b"tail\n"
    assert L.gzclose(f) == Z.Z_OK
    # two members in one file, then garbage: both are read, the garbage is ignored (gzio.c:293-298, 459-476)
    two = str(tmp_path / "two.gz").encode()
    open(two.decode(), "wb").write(gzip.compress(b"first member, ") + gzip.compress(b"second member") + b"\x00\x01garbage")
    f = L.gzopen(two, b"rb")
    assert L.gzread(f, buf, 1000) == 27 and buf.raw[:27] == b"first member, second member"
    assert L.gzclose(f) == Z.Z_OK
    # not a gzip file: handed through
    plain = str(tmp_path / "plain.txt").encode()
    open(plain.decode(), "wb").write(b"plain text, no magic")
    f = L.gzopen(plain, b"rb")
    assert L.gzread(f, buf, 1000) == 20 and buf.raw[:20] == b"plain text, no magic" and L.gzdirect(f) == 1
    assert L.gzclose(f) == Z.Z_OK
    # a damaged member is a data error with a message
    bad = bytearray(open(path.decode(), "rb").read()); bad[5000] ^= 0xFF
    open(two.decode(), "wb").write(bytes(bad))
    f = L.gzopen(two, b"rb")
    n, err = L.gzread(f, buf, 400000), C.c_int(0)
    msg = L.gzerror(f, C.byref(err))
    assert n <= 0 or err.value == Z.Z_DATA_ERROR or L.gzread(f, buf, 10) == -1
    assert err.value == Z.Z_DATA_ERROR and msg.startswith(two) or L.gzerror(f, C.byref(err)) and err.value == Z.Z_DATA_ERROR
    assert L.gzclose(f) in (Z.Z_OK, Z.Z_DATA_ERROR)
    assert L.gzopen(str(tmp_path / "missing.gz").encode(), b"rb") is None


def test_inflate_back(L):
    d = cases.make("mix", 200000, 16)
    raw = b"".join(O.deflate_chunk(d[i:i + 65536], 6, i + 65536 >= len(d)) for i in range(0, len(d), 65536)) + b"TRAILER!"
    IN = C.CFUNCTYPE(C.c_uint, C.c_void_p, C.POINTER(C.c_void_p))
    OUT = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint)
    src = C.create_string_buffer(raw, len(raw))
    state = {"pos": 0, "out": bytearray()}

    def pull(_, nextp):
        n = min(30000, len(raw) - state["pos"])
        nextp[0] = C.addressof(src) + state["pos"]
        state["pos"] += n
        return n

    def push(_, buf, n):
        state["out"] += C.string_at(buf, n)
        assert n <= 32768
        return 0
    L.inflateBackInit_.argtypes = [C.POINTER(Z.ZStream), C.c_int, C.c_void_p, C.c_char_p, C.c_int]
    L.inflateBack.argtypes = [C.POINTER(Z.ZStream), IN, C.c_void_p, OUT, C.c_void_p]
    L.inflateBackEnd.argtypes = [C.POINTER(Z.ZStream)]
    window = C.create_string_buffer(32768)
    s = Z.ZStream()
    assert L.inflateBackInit_(C.byref(s), 15, window, b"1.2.3", C.sizeof(Z.ZStream)) == Z.Z_OK
    s.next_in = None; s.avail_in = 0
    cb_in, cb_out = IN(pull), OUT(push)
    assert L.inflateBack(C.byref(s), cb_in, None, cb_out, None) == Z.Z_STREAM_END
    assert bytes(state["out"]) == d
    rest = C.string_at(s.next_in, s.avail_in) + raw[state["pos"]:]
    assert rest == b"TRAILER!"  # the input behind the stream is the caller's
    assert L.inflateBackEnd(C.byref(s)) == Z.Z_OK


def test_segment_table_from_anywhere_is_checked():
    """The chunk table travels with the data: entries that point outside the input or run backwards are an error of that call, not an
    access outside the buffer."""
    import numpy as np
    import zlib_amd
    eng = zlib_amd.Engine(0)
    d = cases.make("text", 4 * 65536, 17)
    z, offs = eng.deflate_host(d, 6, want_offsets=True)
    good = np.array(offs, dtype=np.uint64)
    assert eng.inflate_host(z, good, 65536) == d
    for k, v in ((1, 10 ** 12), (2, 0), (4, len(z) + 5000)):
        bad = good.copy(); bad[k] = v
        with pytest.raises(zlib_amd.EngineError):
            eng.inflate_host(z, bad, 65536)
    assert eng.inflate_host(z, good, 65536) == d
    eng.close()
