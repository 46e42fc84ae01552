"""Pins the CPU restatement directly to the compiled reference (oracle/_ref/libzref.so).  Skipped when
that build is absent (a checkout without /root/reference and without a prebuilt _ref/)."""
import ctypes as C
import os

import pytest

from oracle import cases, corpus_py as CP, oracle_py as O, refzlib as R

pytestmark = pytest.mark.skipif(not R.available(), reason="oracle/_ref/libzref.so not built")


def test_reference_identity():
    assert R.version() == "1.2.3"
    assert R.compress2(cases.HELLO, 6).hex() == "789ccb48cdc9c9d751c800518a0c0026060496"


@pytest.mark.parametrize("level", range(0, 10))
def test_chunk_function_matches_reference(level):
    g = cases.Lcg(1000 + level)
    for kind in cases.KINDS:
        for _ in range(6):
            n = [g.below(300), g.below(70000) % 65537, 65536 - g.below(300), 65536][g.below(4)]
            data = cases.make(kind, n, seed=g.below(1 << 20))
            for last in (False, True):
                for p0 in ((False, True) if level else (False,)):
                    assert O.deflate_chunk(data, level, last, p0) == R.deflate_chunk_raw(data, level, last, p0), (kind, n, last, p0)


def test_stream_matches_reference_mode_b():
    data = CP.chunks(CP.KIND_SILESIA, 40, 12).tobytes()[:-12345]
    for lvl in (1, 6, 9):
        z = O.deflate_stream(data, lvl)
        assert z == R.deflate_mode_b(data, lvl)
        rc, out = R.uncompress(z, len(data))
        assert rc == 0 and out == data


def test_reference_inflate_accepts_the_256_mib_prefix_stream():
    """BASELINE.json config 4 asks that the stock inflate accepts the chunked stream.  The first 256 MiB of the 4 GiB workload as the mode-B stream
    (the restatement's bytes: the device's are compared with them chunk by chunk in tests/test_gpu_fullsize.py, and handed to the same reference call
    in tests/test_gpu_fullsize.py::test_reference_inflate_accepts_the_device_stream) through the compiled reference's uncompress()."""
    import hashlib
    import json
    n = 256 << 20
    data = CP.chunks(CP.KIND_SILESIA, 0, n // 65536)
    z = O.deflate_stream(data, 6)
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "corpus_silesia.json")))
    rc, out = R.uncompress(z, n)
    assert rc == 0 and len(out) == n and hashlib.sha256(out).digest() == hashlib.sha256(data.tobytes()).digest()
    assert int.from_bytes(z[-4:], "big") == R.adler32(out)
    # the last chunk of this prefix is one of the fixture's last_rows (BFINAL set): the stream ends with exactly those bytes
    row = [r for r in g["last_rows"] if r[0] == n // 65536 - 1][0]
    assert hashlib.sha256(z[-4 - row[4]:-4]).hexdigest()[:16] == row[5]


def test_mode_a_equals_mode_b_except_position0():
    """in one zlib stream flushed with Z_FULL_FLUSH every 64 KiB, chunk 0 is F(.., pos0=0)
    and every later chunk is F(.., pos0_matchable=1)."""
    data = CP.chunks(CP.KIND_SILESIA, 0, 6).tobytes()
    for lvl in (1, 6):
        a = R.deflate_mode_a(data, lvl)
        parts = [R.zlib_header(lvl)]
        for k in range(6):
            parts.append(O.deflate_chunk(data[k * 65536:(k + 1) * 65536], lvl, k == 5, pos0_matchable=(k > 0)))
        parts.append(O.adler32(data).to_bytes(4, "big"))
        assert b"".join(parts) == a


def test_inflate_matches_reference_on_corruption():
    g = cases.Lcg(77)
    for kind in ("text", "rand", "runs", "mix"):
        data = cases.make(kind, 30000, 3)
        for lvl in (0, 1, 6, 9):
            raw = R.deflate_chunk_raw(data, lvl, True)
            assert O.inflate_raw(raw, len(data)) == R.inflate_raw(raw, len(data))
            for _ in range(40):
                c = bytearray(raw)
                for _ in range(1 + g.below(3)):
                    c[g.below(len(c))] ^= 1 << g.below(8)
                a = R.inflate_raw(bytes(c), len(data) + 100)
                b = O.inflate_raw(bytes(c), len(data) + 100)
                assert (a[0], a[3]) == (b[0], b[3])
                if a[0] == 1:
                    assert a == b


def test_checksums_match_reference():
    import ctypes as C
    L = R.lib()
    L.crc32.argtypes = [C.c_ulong, C.c_char_p, C.c_uint]
    L.crc32.restype = C.c_ulong
    g = cases.Lcg(5)
    for n in (0, 1, 15, 16, 17, 5551, 5552, 5553, 70000):
        d = cases.make("rand", n, 3)
        assert O.adler32(d) == R.adler32(d)
        assert O.crc32(d) == L.crc32(0, d, n)
    for _ in range(2000):
        a, b, n = g.next() & 0xFFFFFFFF, g.next() & 0xFFFFFFFF, g.below(1 << 30)
        a = (a & 0xFFFF) % 65521 | ((a >> 16) % 65521) << 16
        b = (b & 0xFFFF) % 65521 | ((b >> 16) % 65521) << 16
        assert O.adler32_combine(a, b, n) == L.adler32_combine(a, b, n)


def test_prime_golden_file_is_what_the_reference_writes_now():
    """tests/golden/prime_kat.json against the compiled reference (oracle/gen_golden_prime.py made it)."""
    import hashlib
    import json
    from oracle import gen_golden_prime as G
    L = R.lib()
    L.deflatePrime.argtypes = [C.POINTER(R.ZStream), C.c_int, C.c_int]
    kat = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "prime_kat.json")))
    assert len(kat) == len(G.INPUTS) * len(G.LEVELS) * len(G.PRIMES)
    for c in kat[::5]:
        z = G.ref_primed(L, cases.make(c["kind"], c["n"], c["seed"]), c["level"], c["wbits"], tuple(c["prime"]), tuple(c["mid"]) if c["mid"] else None)
        assert len(z) == c["len"] and hashlib.sha256(z).hexdigest()[:16] == c["sha"]


def test_geometry_and_fullblock_golden_files_are_what_the_reference_writes_now():
    """tests/golden/geometry_kat.json and fullblock_kat.json against the compiled reference (a sample of the former, all of the latter)."""
    import hashlib
    import json
    from oracle import gen_golden_fullblock as F, gen_golden_geometry as G
    L = R.lib()
    gold = os.path.join(os.path.dirname(__file__), "golden")
    kat = json.load(open(os.path.join(gold, "geometry_kat.json")))
    assert len(kat["chunk"]) == 7 * 9 * len(G.LEVELS) * 4
    for c in kat["chunk"][::23]:
        d = cases.make(c["kind"], c["n"], c["seed"])
        for last in (0, 1):
            z = G.ref_chunk(L, d, c["level"], c["w"], c["m"], bool(last))
            assert (len(z), hashlib.sha256(z).hexdigest()[:16]) == (c["len"][last], c["sha"][last])
    ins = dict(F.inputs())
    for c in json.load(open(os.path.join(gold, "fullblock_kat.json"))):
        for last in (0, 1):
            z = R.deflate_chunk_raw(ins[c["name"]], c["level"], bool(last))
            assert (len(z), hashlib.sha256(z).hexdigest()[:16]) == (c["len"][last], c["sha"][last])
