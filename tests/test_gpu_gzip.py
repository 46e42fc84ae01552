"""GPU: the gzip wrapper and CRC-32 through the engine's C ABI and through libzamd_z.so, against the
oracle, the reference's golden vectors (tests/golden/gzip_kat.json) and the reference's error messages."""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import cases, corpus_py as CP, oracle_py as O  # noqa: E402


@pytest.fixture(scope="module")
def eng():
    import zlib_amd
    e = zlib_amd.Engine(0)
    yield e
    e.close()


MULTI = {"corpus0x5": lambda: CP.chunks(0, 7, 5).tobytes(), "corpus1x3-ragged": lambda: CP.chunks(1, 2, 3).tobytes()[:-4321],
         "hello": lambda: cases.hello_1mib()[:300000]}


def test_engine_gzip_member_matches_reference_vectors(eng, golden):
    from zlib_amd import gpu
    kat = golden("gzip_kat.json")
    for key, (ln, sha, crc) in kat["multi"].items():
        name, lvl = key.split("/")
        data = MULTI[name]()
        z = eng.deflate_host(data, int(lvl), flags=gpu.F_FINAL | gpu.F_GZIP_WRAP)
        assert (len(z), hashlib.sha256(z).hexdigest()[:16]) == (ln, sha), key
        assert eng.last.crc32 == crc
    for key, want in kat["single"].items():
        kind, n, lvl = key.split("/")
        if int(lvl) == 0:
            continue  # level 0 is framing done by the host library (below)
        z = eng.deflate_host(cases.make(kind, int(n), 9), int(lvl), flags=gpu.F_FINAL | gpu.F_GZIP_WRAP)
        assert (z.hex() if len(z) <= 64 else [len(z), hashlib.sha256(z).hexdigest()[:16]]) == want, key


def test_crc32_device_and_flag(eng):
    import torch
    from zlib_amd import gpu
    g = cases.Lcg(5)
    for n in (0, 1, 255, 256, 257, 65535, 65536, 65537, 3 * 65536 + 17, 40 * 65536):
        data = cases.make(cases.KINDS[g.below(len(cases.KINDS))], n, g.below(1000))
        t = torch.from_numpy(np.frombuffer(data + b"\0", dtype=np.uint8).copy()).cuda()
        assert eng.crc32_device(t.data_ptr(), n) == O.crc32(data), n
        if n:
            eng.deflate_host(data, 6, flags=gpu.F_FINAL | gpu.F_CRC32)  # the checksum alone, no wrapper
            assert eng.last.crc32 == O.crc32(data)
            eng.deflate_host(data, 6, flags=gpu.F_FINAL, chunk_size=4096)  # flag absent: field is 0
            assert eng.last.crc32 == 0


def test_host_api_gzip(golden):
    import zhost as Z
    kat = golden("gzip_kat.json")
    # one-piece deflate(Z_FINISH) with windowBits 31, all levels incl. 0: byte-identical to the reference's gzip member on one chunk
    for key, want in kat["single"].items():
        kind, n, lvl = key.split("/")
        data = cases.make(kind, int(n), 9)
        z, codes, info = Z.deflate_stream(data, int(lvl), [(len(data), Z.Z_FINISH)], window_bits=31)
        assert (z.hex() if len(z) <= 64 else [len(z), hashlib.sha256(z).hexdigest()[:16]]) == want, key
        assert info["adler"] == O.crc32(data)  # strm->adler carries the CRC-32 of a gzip stream (deflate.c:577, 968)
    data = CP.chunks(0, 11, 4).tobytes()[:-777]
    # streamed in pieces with flushes: the reference's inflate is not here on the GPU box; our own inflate must agree, in both modes
    z, codes, info = Z.deflate_stream(data, 6, [(100000, Z.Z_NO_FLUSH), (50000, Z.Z_FULL_FLUSH), (len(data) - 150000, Z.Z_FINISH)], in_step=30000, out_step=7777,
                                      window_bits=31)
    assert z[:10] == O.gzip_header(6) and z[-8:] == O.crc32(data).to_bytes(4, "little") + len(data).to_bytes(4, "little")
    for wb in (31, 47):
        rc, out, msg, adler = Z.inflate_stream(z, len(data) + 10, in_step=50000, out_step=60000, window_bits=wb)
        assert (rc, out, adler) == (Z.Z_STREAM_END, data, O.crc32(data)), (wb, rc, msg)
    zz = O.deflate_stream(data, 6)  # a zlib stream through the detecting mode, and refused by the gzip-only mode
    assert Z.inflate_stream(zz, len(data) + 10, window_bits=47)[:2] == (Z.Z_STREAM_END, data)
    rc, out, msg, adler = Z.inflate_stream(zz, len(data) + 10, window_bits=31)
    assert (rc, msg) == (Z.Z_DATA_ERROR, "incorrect header check")
    # a gzip member with name, comment, extra field and header CRC (inflate.c:634-759)
    import zlib as pyz
    body = O.deflate_stream(data, 9)[2:-4]
    hdr = bytearray([31, 139, 8, 2 | 4 | 8 | 16, 1, 2, 3, 4, 2, 3]) + bytes([5, 0]) + b"extra" + b"name.txt\0" + b"a comment\0"
    hdr += (pyz.crc32(bytes(hdr)) & 0xFFFF).to_bytes(2, "little")
    member = bytes(hdr) + body + O.crc32(data).to_bytes(4, "little") + len(data).to_bytes(4, "little")
    assert Z.inflate_stream(member, len(data) + 10, window_bits=31)[:2] == (Z.Z_STREAM_END, data)
    # the reference's verdicts on damaged members
    bad = bytearray(member); bad[len(hdr) - 1] ^= 1
    assert Z.inflate_stream(bytes(bad), len(data) + 10, window_bits=31)[0::2] == (Z.Z_DATA_ERROR, "header crc mismatch")
    bad = bytearray(member); bad[-5] ^= 0x10
    assert Z.inflate_stream(bytes(bad), len(data) + 10, window_bits=31)[0::2] == (Z.Z_DATA_ERROR, "incorrect data check")
    bad = bytearray(member); bad[-1] ^= 0x10
    assert Z.inflate_stream(bytes(bad), len(data) + 10, window_bits=31)[0::2] == (Z.Z_DATA_ERROR, "incorrect length check")
    bad = bytearray(member); bad[3] |= 0x20
    assert Z.inflate_stream(bytes(bad), len(data) + 10, window_bits=31)[0::2] == (Z.Z_DATA_ERROR, "unknown header flags set")
    bad = bytearray(member); bad[2] = 7
    assert Z.inflate_stream(bytes(bad), len(data) + 10, window_bits=31)[0::2] == (Z.Z_DATA_ERROR, "unknown compression method")
    # crc32() / crc32_combine() of the library
    L = Z.lib()
    import ctypes as C
    L.crc32.restype = C.c_ulong; L.crc32.argtypes = [C.c_ulong, C.c_char_p, C.c_uint]
    L.crc32_combine.restype = C.c_ulong; L.crc32_combine.argtypes = [C.c_ulong, C.c_ulong, C.c_long]
    assert L.crc32(0, data, len(data)) == O.crc32(data)
    for c1, c2, ln, want in kat["combine"]:
        assert L.crc32_combine(c1, c2, ln) == want
