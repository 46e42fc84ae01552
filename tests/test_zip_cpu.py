"""The PKZIP fixture (tests/golden/zip_kat.json, archives written by the reference's minizip) against an independent reader:
Python's zipfile must list and extract every member with the inputs oracle/cases.py regenerates.  Pins the fixture the GPU tests
(tests/test_gpu_zip.py) compare the product's archives with."""
import base64
import io
import json
import os
import zipfile

import pytest

from oracle import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KAT = json.load(open(os.path.join(ROOT, "tests", "golden", "zip_kat.json")))


@pytest.mark.parametrize("arc", KAT["archives"], ids=lambda a: "L%d" % a["level"])
def test_fixture_archives_hold_the_case_inputs(arc):
    zf = zipfile.ZipFile(io.BytesIO(base64.b64decode(arc["zip_b64"])))
    assert zf.testzip() is None
    assert [i.filename for i in zf.infolist()] == [m[0] for m in KAT["members"]]
    for (name, kind, n, seed), info in zip(KAT["members"], zf.infolist()):
        assert zf.read(name) == cases.make(kind, n, seed)
        assert info.compress_type == (zipfile.ZIP_DEFLATED if arc["level"] else zipfile.ZIP_STORED)
        assert info.date_time == (2005, 7, 18, 12, 34, 56)
        want_flag = {8: 2, 9: 2, 2: 4, 1: 6}.get(arc["level"], 0)  # zip.c:760-766
        assert info.flag_bits == want_flag


def test_directory_reader_survives_damaged_archives(tmp_path):
    """zamd_unzip_open / count / stat / locate parse a directory that may come from anywhere: 3 000 damaged copies of a fixture archive
    (bytes of the end record and the central directory overwritten, the file cut short) must be refused or listed -- never a crash, never a
    member that points outside the file.  These entry points do not touch the GPU, so this runs in the CPU suite (a crash would end the run)."""
    import ctypes as C
    import random
    lib = os.environ.get("ZAMD_Z_LIB") or os.path.join(ROOT, "zlib_amd", "libzamd_z.so")  # (ZAMD_Z_LIB: the sanitizer build)
    if not os.path.exists(lib):
        pytest.skip("host library not built")
    L = C.CDLL(lib)

    class Entry(C.Structure):
        _fields_ = [("name", C.c_char * 512), ("crc32", C.c_ulong), ("compressed_size", C.c_ulong), ("uncompressed_size", C.c_ulong),
                    ("dos_date", C.c_ulong), ("local_header_offset", C.c_ulong), ("method", C.c_int), ("flag", C.c_int), ("internal_fa", C.c_int)]  # include/zamd_zip.h
    L.zamd_unzip_open.restype = C.c_void_p
    L.zamd_unzip_open.argtypes = [C.c_char_p]
    L.zamd_unzip_count.argtypes = [C.c_void_p]
    L.zamd_unzip_stat.argtypes = [C.c_void_p, C.c_int, C.POINTER(Entry)]
    L.zamd_unzip_locate.argtypes = [C.c_void_p, C.c_char_p]
    L.zamd_unzip_close.argtypes = [C.c_void_p]
    good = base64.b64decode(KAT["archives"][2]["zip_b64"])
    cd = good.find(b"PK\x01\x02")
    rnd = random.Random(20051)
    path = tmp_path / "d.zip"
    opened = refused = 0
    for it in range(3000):
        raw = bytearray(good)
        mode = it % 3
        if mode == 0:  # a few bytes of the directory / end record
            for _ in range(rnd.randint(1, 4)):
                raw[rnd.randrange(cd, len(raw))] = rnd.randrange(256)
        elif mode == 1:  # a 16- or 32-bit field set to an extreme
            at = rnd.randrange(cd, len(raw) - 4)
            raw[at:at + 4] = rnd.choice([b"\xff\xff\xff\xff", b"\x00\x00\x00\x00", b"\xff\xff\x00\x00", b"\x00\x00\x00\x80"])
        else:  # cut short
            raw = raw[: rnd.randrange(0, len(raw))]
        path.write_bytes(bytes(raw))
        u = L.zamd_unzip_open(str(path).encode())
        if not u:
            refused += 1
            continue
        opened += 1
        n = L.zamd_unzip_count(u)
        assert 0 <= n <= 0xFFFF
        e = Entry()
        for i in range(min(n, 8)):
            assert L.zamd_unzip_stat(u, i, C.byref(e)) == 0
            assert len(e.name) < 512
        L.zamd_unzip_locate(u, b"no such member")
        assert L.zamd_unzip_stat(u, n, C.byref(e)) != 0
        L.zamd_unzip_close(u)
    assert opened > 100 and refused > 100
