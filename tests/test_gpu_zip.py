"""SURVEY.md 8f N3 -- the PKZIP container around the engine.
  * zamd_zip_* (include/zamd_zip.h) must write the archive the reference's minizip writes for the same members, level and file
    time: byte for byte against tests/golden/zip_kat.json (members of at most one chunk, where the reference's single stream and the
    chunked stream coincide);
  * the reference's minizip / miniunz, compiled UNMODIFIED from the mount and linked with the product (oracle/_ref/minizip_zamd,
    miniunz_zamd), must produce that same archive, and -- for members of many chunks -- the archive zamd_zip_* writes;
  * zamd_unzip_* must read archives written by the reference, by the product and by a foreign writer (Python's zipfile over the system zlib),
    and refuse a member whose CRC-32 does not match."""
import base64
import ctypes as C
import io
import json
import os
import subprocess
import zipfile

import pytest

from oracle import cases

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KAT = json.load(open(os.path.join(ROOT, "tests", "golden", "zip_kat.json")))
ENV = dict(os.environ, TZ="UTC", LD_LIBRARY_PATH=os.path.join(ROOT, "zlib_amd") + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))


class Entry(C.Structure):
    _fields_ = [("name", C.c_char * 512), ("crc32", C.c_ulong), ("compressed_size", C.c_ulong), ("uncompressed_size", C.c_ulong), ("dos_date", C.c_ulong),
                ("local_header_offset", C.c_ulong), ("method", C.c_int), ("flag", C.c_int), ("internal_fa", C.c_int)]


@pytest.fixture(scope="module")
def L():
    from tests import zhost
    lib = zhost.lib()
    lib.zamd_zip_open.restype = C.c_void_p
    lib.zamd_zip_open.argtypes = [C.c_char_p]
    lib.zamd_zip_add.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_ulong, C.c_int, C.c_ulong, C.c_char_p]
    lib.zamd_zip_close.argtypes = [C.c_void_p, C.c_char_p]
    lib.zamd_unzip_open.restype = C.c_void_p
    lib.zamd_unzip_open.argtypes = [C.c_char_p]
    lib.zamd_unzip_count.argtypes = [C.c_void_p]
    lib.zamd_unzip_stat.argtypes = [C.c_void_p, C.c_int, C.POINTER(Entry)]
    lib.zamd_unzip_locate.argtypes = [C.c_void_p, C.c_char_p]
    lib.zamd_unzip_read.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_ulong]
    lib.zamd_unzip_read.restype = C.c_long
    lib.zamd_unzip_close.argtypes = [C.c_void_p]
    return lib


def write_zip(L, path, members, level, dos_date, comment=None):
    z = L.zamd_zip_open(str(path).encode())
    assert z
    for name, data in members:
        assert L.zamd_zip_add(z, name.encode(), data, len(data), level, dos_date, None) == 0
    assert L.zamd_zip_close(z, comment) == 0
    return open(path, "rb").read()


def read_zip(L, path):
    u = L.zamd_unzip_open(str(path).encode())
    assert u
    out = []
    for i in range(L.zamd_unzip_count(u)):
        e = Entry()
        assert L.zamd_unzip_stat(u, i, C.byref(e)) == 0
        buf = C.create_string_buffer(max(e.uncompressed_size, 1))
        n = L.zamd_unzip_read(u, i, buf, e.uncompressed_size)
        out.append((e.name.decode(), n, buf.raw[: max(n, 0)], e))
    L.zamd_unzip_close(u)
    return out


@pytest.mark.parametrize("arc", KAT["archives"], ids=lambda a: "L%d" % a["level"])
def test_bulk_writer_matches_the_reference_minizip(L, arc, tmp_path):
    members = [(name, cases.make(kind, n, seed)) for name, kind, n, seed in KAT["members"]]
    got = write_zip(L, tmp_path / "t.zip", members, arc["level"], arc["dos_date"])
    assert got == base64.b64decode(arc["zip_b64"])


@pytest.mark.parametrize("arc", KAT["archives"], ids=lambda a: "L%d" % a["level"])
def test_bulk_reader_reads_the_reference_archives(L, arc, tmp_path):
    p = tmp_path / "ref.zip"
    p.write_bytes(base64.b64decode(arc["zip_b64"]))
    got = read_zip(L, p)
    assert [g[0] for g in got] == [m[0] for m in KAT["members"]]
    for (name, kind, n, seed), (_, rc, data, e) in zip(KAT["members"], got):
        assert rc == n and data == cases.make(kind, n, seed), name
        assert e.dos_date == arc["dos_date"] and e.method == (8 if arc["level"] else 0)


def _minizip(tmp_path, level, members, exe="minizip_zamd"):
    path = os.path.join(ROOT, "oracle", "_ref", exe)
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/%s was not built (no reference mount at build time)" % exe)
    for name, data in members:
        p = tmp_path / name
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_bytes(data)
        os.utime(p, (KAT["mtime"], KAT["mtime"]))
    r = subprocess.run([path, "-o", "-%d" % level, "mz.zip"] + [m[0] for m in members], cwd=str(tmp_path), env=ENV, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode(errors="replace")
    return (tmp_path / "mz.zip").read_bytes()


@pytest.mark.parametrize("level", [1, 6])
def test_reference_minizip_linked_with_the_product_writes_the_reference_archive(level, tmp_path):
    arc = next(a for a in KAT["archives"] if a["level"] == level)
    members = [(name, cases.make(kind, n, seed)) for name, kind, n, seed in KAT["members"]]
    assert _minizip(tmp_path, level, members) == base64.b64decode(arc["zip_b64"])


def test_many_chunk_members_minizip_and_bulk_writer_agree_and_everyone_reads_them(L, tmp_path):
    members = [("big.txt", cases.make("text", 3 * 1024 * 1024 + 17, 21)), ("big.mix", cases.make("mix", 1024 * 1024 + 65536, 22)), ("tiny", b"x")]
    mz = _minizip(tmp_path, 6, members)
    dos = int.from_bytes(mz[10:14], "little")
    bulk = write_zip(L, tmp_path / "bulk.zip", members, 6, dos)
    assert bulk == mz
    # foreign reader
    zf = zipfile.ZipFile(io.BytesIO(bulk))
    assert zf.testzip() is None
    for name, data in members:
        assert zf.read(name) == data
    # the product's reader, and the reference's miniunz over the product
    got = read_zip(L, tmp_path / "bulk.zip")
    assert [(g[0], g[2]) for g in got] == members
    exe = os.path.join(ROOT, "oracle", "_ref", "miniunz_zamd")
    out = tmp_path / "out"
    out.mkdir()
    r = subprocess.run([exe, "-o", str(tmp_path / "bulk.zip")], cwd=str(out), env=ENV, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode(errors="replace")
    for name, data in members:
        assert (out / name).read_bytes() == data


def test_foreign_archive_and_crc_damage(L, tmp_path):
    members = [("f1", cases.make("text", 300000, 31)), ("f2", cases.make("rand", 70000, 32)), ("f3", b""), ("f4", cases.make("runs", 200000, 33))]
    p = tmp_path / "py.zip"
    with zipfile.ZipFile(p, "w", zipfile.ZIP_DEFLATED, compresslevel=7) as zf:
        for i, (name, data) in enumerate(members):
            zf.writestr(zipfile.ZipInfo(name, (2005, 7, 18, 12, 34, 56)), data, zipfile.ZIP_STORED if i == 1 else zipfile.ZIP_DEFLATED)
    got = read_zip(L, p)
    assert [(g[0], g[2]) for g in got] == members
    u = L.zamd_unzip_open(str(p).encode())
    assert L.zamd_unzip_locate(u, b"f4") == 3 and L.zamd_unzip_locate(u, b"nope") == -102
    small = C.create_string_buffer(10)
    assert L.zamd_unzip_read(u, 0, small, 10) == -102
    L.zamd_unzip_close(u)
    # flip one bit of the CRC field in the central directory of f1 and of the stored member f2
    raw = bytearray(p.read_bytes())
    cd = raw.find(b"PK\x01\x02")
    raw[cd + 16] ^= 1
    nxt = raw.find(b"PK\x01\x02", cd + 4)
    raw[nxt + 16] ^= 1
    bad = tmp_path / "bad.zip"
    bad.write_bytes(bytes(raw))
    got = read_zip(L, bad)
    assert got[0][1] == -105 and got[1][1] == -105 and got[2][1] == 0 and got[3][2] == members[3][1]
    # a directory that claims more compressed bytes than the file holds: refused before anything is read or allocated
    raw = bytearray(p.read_bytes())
    cd = raw.find(b"PK\x01\x02")
    raw[cd + 20:cd + 24] = (0xFFFFFF00).to_bytes(4, "little")
    liar = tmp_path / "liar.zip"
    liar.write_bytes(bytes(raw))
    got = read_zip(L, liar)
    assert got[0][1] == -103 and got[3][2] == members[3][1]
    # a directory that announces fewer bytes than the member decodes to (with room for exactly what it announces): a bad file, not a hang
    raw = bytearray(p.read_bytes())
    cd = raw.find(b"PK\x01\x02")
    raw[cd + 24:cd + 28] = (100).to_bytes(4, "little")
    short = tmp_path / "short.zip"
    short.write_bytes(bytes(raw))
    got = read_zip(L, short)
    assert got[0][1] in (-103, -105) and got[3][2] == members[3][1]
    # not an archive
    junk = tmp_path / "junk.zip"
    junk.write_bytes(b"PK" + bytes(100))
    assert not L.zamd_unzip_open(str(junk).encode())
