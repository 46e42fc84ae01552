"""deflatePrime (qcsrc/deflate.c:404-413) through the z_stream API and through the engine's C ABI, against the compiled reference's streams
(tests/golden/prime_kat.json, written by oracle/gen_golden_prime.py): bit-exact, all levels, raw and zlib-wrapped, stored blocks included (their
padding moves with the primed bits)."""
import ctypes as C
import hashlib
import json
import os

import pytest

pytestmark = pytest.mark.gpu

from oracle import cases  # noqa: E402
import zhost as Z  # noqa: E402

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "prime_kat.json")))


def _primed(L, data, level, wbits, prime, mid, whole):
    s = Z.ZStream()
    assert L.deflateInit2_(C.byref(s), level, 8, wbits, 8, 0, b"1.2.3", C.sizeof(Z.ZStream)) == Z.Z_OK
    assert L.deflatePrime(C.byref(s), prime[0], prime[1]) == Z.Z_OK
    cap = len(data) + (len(data) >> 7) + 512
    out = C.create_string_buffer(cap)
    inb = C.create_string_buffer(data, max(len(data), 1))
    s.next_out = C.addressof(out); s.avail_out = cap
    if whole:  # one call: the library cuts the chunks itself
        s.next_in = C.addressof(inb); s.avail_in = len(data)
        assert L.deflate(C.byref(s), Z.Z_FINISH) == Z.Z_STREAM_END
    else:
        nchunks = max(1, (len(data) + 65535) // 65536)
        for k in range(nchunks):
            s.next_in = C.addressof(inb) + k * 65536; s.avail_in = min(65536, len(data) - k * 65536)
            last = k + 1 == nchunks
            assert L.deflate(C.byref(s), Z.Z_FINISH if last else Z.Z_FULL_FLUSH) == (Z.Z_STREAM_END if last else Z.Z_OK)
            if k == 0 and mid is not None and not last:
                assert L.deflatePrime(C.byref(s), mid[0], mid[1]) == Z.Z_OK
    z = out.raw[: s.total_out]
    L.deflateEnd(C.byref(s))
    return z


def test_deflate_prime_matches_the_reference():
    L = Z.lib()
    L.deflatePrime.argtypes = [C.POINTER(Z.ZStream), C.c_int, C.c_int]
    bad = []
    for c in KAT:
        d = cases.make(c["kind"], c["n"], c["seed"])
        z = _primed(L, d, c["level"], c["wbits"], c["prime"], c["mid"], False)
        # (api_*: the reference as ONE stream driven by the same calls -- what the z_stream API writes since round 4; len / sha: a fresh stream per chunk,
        #  what the engine's chunk entry points write; the two coincide for a single chunk)
        if len(z) != c["api_len"] or hashlib.sha256(z).hexdigest()[:16] != c["api_sha"] or (c["stream"] is not None and c["n"] <= 65536 and z.hex() != c["stream"]):
            bad.append((c["kind"], c["n"], c["level"], c["wbits"], c["prime"], c["mid"], len(z), c["api_len"]))
        if c["mid"] is None and c["n"] <= 65536:  # fed in one piece the library writes the same stream (one continuous stream since round 4: only when there is no flush point)
            assert _primed(L, d, c["level"], c["wbits"], c["prime"], None, True) == z
    assert not bad, bad[:10]


def test_engine_prime_parameter_and_its_limits():
    import numpy as np
    import zlib_amd
    from zlib_amd import gpu
    e = zlib_amd.Engine(0)
    try:
        for c in KAT:
            if c["wbits"] != -15 or c["mid"] is not None or c["level"] == 0:
                continue
            d = np.frombuffer(cases.make(c["kind"], c["n"], c["seed"]), dtype=np.uint8)
            z = e.deflate_host(d, c["level"], flags=gpu.F_FINAL, prime=tuple(c["prime"]))
            assert len(z) == c["len"] and hashlib.sha256(z).hexdigest()[:16] == c["sha"], c
        with pytest.raises(gpu.EngineError):
            e.deflate_host(b"abc", 6, flags=gpu.F_FINAL, prime=(17, 0))
        with pytest.raises(gpu.EngineError):
            e.deflate_host(b"abc", 6, flags=gpu.F_FINAL | gpu.F_ZLIB_WRAP, prime=(3, 1))
    finally:
        e.close()


def test_deflate_prime_refuses_what_it_cannot_place():
    L = Z.lib()
    L.deflatePrime.argtypes = [C.POINTER(Z.ZStream), C.c_int, C.c_int]
    s = Z.ZStream()
    assert L.deflateInit2_(C.byref(s), 6, 8, -15, 8, 0, b"1.2.3", C.sizeof(Z.ZStream)) == Z.Z_OK
    assert L.deflatePrime(C.byref(s), 17, 0) == Z.Z_STREAM_ERROR
    assert L.deflatePrime(C.byref(s), -1, 0) == Z.Z_STREAM_ERROR
    data = b"x" * 1000
    inb = C.create_string_buffer(data)
    out = C.create_string_buffer(4096)
    s.next_in = C.addressof(inb); s.avail_in = 1000; s.next_out = C.addressof(out); s.avail_out = 4096
    assert L.deflate(C.byref(s), Z.Z_NO_FLUSH) == Z.Z_OK  # (the bytes wait for their chunk: the reference would have bits in flight here)
    assert L.deflatePrime(C.byref(s), 3, 1) == Z.Z_STREAM_ERROR
    L.deflateEnd(C.byref(s))
