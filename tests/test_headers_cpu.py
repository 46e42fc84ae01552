"""The host library's parsers of untrusted headers, without a GPU: gzip and zlib headers through inflate() (zamd_zlib.c parse_header), gz files
through gzopen / gzread (zamd_gzio.c gz_header_in).  Damaged headers must be refused or accepted -- never a crash, never a read behind the
input.  No GPU is needed: a header that passes ends in Z_MEM_ERROR here (no engine), which is as far as these tests go.  tests/test_sanitizers_cpu.py
runs this file again with the ASan + UBSan build of the library."""
import ctypes as C
import gzip
import os
import random
import struct

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get("ZAMD_Z_LIB") or os.path.join(ROOT, "zlib_amd", "libzamd_z.so")


class ZStream(C.Structure):
    _fields_ = [("next_in", C.c_void_p), ("avail_in", C.c_uint), ("total_in", C.c_ulong), ("next_out", C.c_void_p),
                ("avail_out", C.c_uint), ("total_out", C.c_ulong), ("msg", C.c_char_p), ("state", C.c_void_p),
                ("zalloc", C.c_void_p), ("zfree", C.c_void_p), ("opaque", C.c_void_p), ("data_type", C.c_int),
                ("adler", C.c_ulong), ("reserved", C.c_ulong)]


class GzHeader(C.Structure):
    _fields_ = [("text", C.c_int), ("time", C.c_ulong), ("xflags", C.c_int), ("os", C.c_int), ("extra", C.c_void_p), ("extra_len", C.c_uint),
                ("extra_max", C.c_uint), ("name", C.c_void_p), ("name_max", C.c_uint), ("comment", C.c_void_p), ("comm_max", C.c_uint),
                ("hcrc", C.c_int), ("done", C.c_int)]


@pytest.fixture(scope="module")
def L():
    if not os.path.exists(LIB):
        pytest.skip("host library not built")
    lib = C.CDLL(LIB)
    P = C.POINTER(ZStream)
    lib.inflateInit2_.argtypes = [P, C.c_int, C.c_char_p, C.c_int]
    lib.inflate.argtypes = [P, C.c_int]
    lib.inflateEnd.argtypes = [P]
    lib.inflateGetHeader.argtypes = [P, C.POINTER(GzHeader)]
    lib.gzopen.restype = C.c_void_p
    lib.gzopen.argtypes = [C.c_char_p, C.c_char_p]
    lib.gzread.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
    lib.gzclose.argtypes = [C.c_void_p]
    s = ZStream()
    if lib.inflateInit2_(C.byref(s), 15, b"1.2.3", C.sizeof(ZStream)) != 0:
        pytest.skip("this build of the library refuses to start without a GPU (the product does; the sanitizer build, -DZAMD_SAN_NO_ENGINE, does not)")
    lib.inflateEnd(C.byref(s))
    return lib


def gzip_header(rnd):
    flg = rnd.choice([0, 2, 4, 8, 16, 4 | 8, 4 | 8 | 16 | 2, 8 | 16, 1 | 4])
    h = bytearray(b"\x1f\x8b\x08" + bytes([flg]) + struct.pack("<IBB", rnd.randrange(1 << 32), rnd.choice([0, 2, 4]), rnd.choice([3, 255])))
    if flg & 4:
        x = bytes(rnd.randrange(256) for _ in range(rnd.randrange(0, 40)))
        h += struct.pack("<H", len(x)) + x
    if flg & 8:
        h += bytes(rnd.randrange(1, 256) for _ in range(rnd.randrange(0, 30))) + b"\0"
    if flg & 16:
        h += bytes(rnd.randrange(1, 256) for _ in range(rnd.randrange(0, 30))) + b"\0"
    if flg & 2:
        import zlib
        h += struct.pack("<H", zlib.crc32(bytes(h)) & 0xffff)
    return bytes(h)


def test_damaged_gzip_and_zlib_headers_through_inflate(L):
    rnd = random.Random(31337)
    seen = set()
    for it in range(4000):
        wbits = rnd.choice([31, 47, 15])
        raw = bytearray(gzip_header(rnd) if wbits != 15 else bytes([0x78, rnd.choice([0x01, 0x9c, 0xda, 0xbb, 0x20])]) + b"\x00\x01\x02\x03")
        raw += b"\x03\x00" + bytes(8)  # an empty final block and room for a trailer
        mode = it % 4
        if mode == 1:
            for _ in range(rnd.randint(1, 3)):
                raw[rnd.randrange(len(raw))] = rnd.randrange(256)
        elif mode == 2:
            raw = raw[: rnd.randrange(0, len(raw))]
        elif mode == 3 and len(raw) > 12:
            at = rnd.randrange(10, len(raw) - 2)
            raw[at: at + 2] = rnd.choice([b"\xff\xff", b"\x00\x00", b"\xff\x7f"])
        s = ZStream()
        assert L.inflateInit2_(C.byref(s), wbits, b"1.2.3", C.sizeof(ZStream)) == 0
        name = C.create_string_buffer(16)
        extra = C.create_string_buffer(8)
        comm = C.create_string_buffer(4)
        gh = GzHeader()
        gh.name, gh.name_max, gh.extra, gh.extra_max, gh.comment, gh.comm_max = C.addressof(name), 16, C.addressof(extra), 8, C.addressof(comm), 4
        if wbits != 15 and it % 2:
            L.inflateGetHeader(C.byref(s), C.byref(gh))
        src = C.create_string_buffer(bytes(raw), max(len(raw), 1))
        out = C.create_string_buffer(64)
        pos, rc = 0, 0
        step = rnd.choice([1, 3, 7, len(raw) or 1])
        for _ in range(len(raw) + 4):  # byte by byte or all at once
            n = min(step, len(raw) - pos)
            s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src) + pos, n, C.addressof(out), 64
            rc = L.inflate(C.byref(s), 0)
            pos += n - s.avail_in
            if rc not in (0, -5) or n == 0:
                break
        seen.add(rc)
        assert rc in (0, 1, 2, -3, -4, -5), rc  # OK / END / NEED_DICT / DATA / MEM (no engine here) / BUF
        L.inflateEnd(C.byref(s))
    assert -3 in seen and (-4 in seen or 1 in seen)


def test_damaged_gz_files_through_gzread(L, tmp_path):
    rnd = random.Random(777)
    good = gzip.compress(b"hello, hello! " * 50, 6)
    path = str(tmp_path / "d.gz").encode()
    buf = C.create_string_buffer(256)
    for it in range(1500):
        raw = bytearray(gzip_header(rnd) + good[10:]) if it % 2 else bytearray(good)
        mode = it % 3
        if mode == 0:
            for _ in range(rnd.randint(1, 3)):
                raw[rnd.randrange(min(len(raw), 60))] = rnd.randrange(256)
        elif mode == 1:
            raw = raw[: rnd.randrange(0, min(len(raw), 80))]
        with open(path, "wb") as f:
            f.write(bytes(raw))
        g = L.gzopen(path, b"rb")
        assert g
        for _ in range(4):
            if L.gzread(g, buf, 256) <= 0:
                break
        L.gzclose(g)
