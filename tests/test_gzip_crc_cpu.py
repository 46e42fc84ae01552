"""CPU: the CRC-32 / gzip row.  The oracle's restatements against the reference's golden vectors (tests/golden/gzip_kat.json,
made by oracle/gen_golden_gzip.py) and, when oracle/_ref is built, against the reference itself."""
import hashlib

import pytest

from oracle import cases, corpus_py as CP, oracle_py as O, refzlib as R


def enc(z):
    return z.hex() if len(z) <= 64 else [len(z), hashlib.sha256(z).hexdigest()[:16]]


def test_crc32_golden(golden):
    kat = golden("gzip_kat.json")
    for key, want in kat["crc32"].items():
        kind, n = key.split("/")
        data = cases.make(kind, int(n), 5)
        assert O.crc32(data) == want, key
        if len(data) > 10:  # incremental use, crc32.c:219
            assert O.crc32(data[7:], O.crc32(data[:7])) == want
    for c1, c2, ln, want in kat["combine"]:
        assert O.crc32_combine(c1, c2, ln) == want


def test_gzip_members_golden(golden):
    kat = golden("gzip_kat.json")
    for key, want in kat["single"].items():
        kind, n, lvl = key.split("/")
        data = cases.make(kind, int(n), 9)
        if int(lvl) == 0:
            continue  # level 0 is host-side framing (zamd_zlib.c), not a path of the oracle's deflate_stream
        assert enc(O.deflate_stream_gzip(data, int(lvl))) == want, key
    multi = {"corpus0x5": CP.chunks(0, 7, 5).tobytes(), "corpus1x3-ragged": CP.chunks(1, 2, 3).tobytes()[:-4321], "hello": cases.hello_1mib()[:300000]}
    for key, (ln, sha, crc) in kat["multi"].items():
        name, lvl = key.split("/")
        z = O.deflate_stream_gzip(multi[name], int(lvl))
        assert (len(z), hashlib.sha256(z).hexdigest()[:16]) == (ln, sha), key
        assert O.crc32(multi[name]) == crc


@pytest.mark.skipif(not R.available(), reason="oracle/_ref/libzref.so not built")
def test_crc32_and_gzip_against_reference():
    g = cases.Lcg(99)
    for _ in range(40):
        data = cases.make(cases.KINDS[g.below(len(cases.KINDS))], g.below(100000), g.below(1000))
        assert O.crc32(data) == R.crc32(data)
        cut = g.below(len(data) + 1)
        a, b = data[:cut], data[cut:]
        assert O.crc32_combine(O.crc32(a), O.crc32(b), len(b)) == R.crc32_combine(R.crc32(a), R.crc32(b), len(b)) == R.crc32(data)
    data = CP.chunks(0, 3, 3).tobytes()
    for lvl in (1, 6, 9):
        z = O.deflate_stream_gzip(data, lvl)
        rc, back, used, msg, adler = R.inflate_wbits(z, 31, len(data) + 8)
        assert (rc, back, used, adler) == (1, data, len(z), R.crc32(data))
        assert R.inflate_wbits(z, 47, len(data) + 8)[1] == data  # zlib-or-gzip detection
    small = cases.make("text", 3000, 1)
    for lvl in (1, 6, 9):
        assert O.deflate_stream_gzip(small, lvl) == R.deflate_wbits(small, lvl, 31)  # one chunk: the reference's own gzip member
