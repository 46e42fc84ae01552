#!/usr/bin/env python3
"""Golden vectors of deflatePrime (qcsrc/deflate.c:404-413), produced by the compiled reference:  python oracle/gen_golden_prime.py
->  tests/golden/prime_kat.json

Every case is the reference's chunk function per 65536 bytes (a fresh raw stream, Z_FULL_FLUSH behind the chunk, Z_FINISH on the last one:
mode B, the stream the product writes) with deflatePrime(bits, value) called before chunk 0 is compressed.  Streams of up to 200 bytes are kept whole (hex), all of them by
length and sha256[:16].  "mid" cases call deflatePrime a second time in front of chunk 1.
TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cases, refzlib as R  # noqa: E402

PRIMES = [(1, 1), (3, 5), (5, 0), (7, 0x55), (8, 0xA7), (13, 0x1234), (16, 0xBEEF)]
INPUTS = [("text", 0, 1), ("text", 10, 2), ("text", 5000, 3), ("rand", 3000, 4), ("rand", 65536, 5), ("mix", 70000, 6), ("runs", 140000, 7), ("ab", 65537, 8)]
LEVELS = [0, 1, 3, 4, 6, 9]


def ref_primed_chunk(L, data, level, last, prime):
    """The reference's chunk function (a fresh raw stream, Z_FULL_FLUSH or Z_FINISH behind the chunk) with deflatePrime called first."""
    s = R.ZStream()
    assert L.deflateInit2_(C.byref(s), level, 8, -15, 8, 0, b"1.2.3", C.sizeof(R.ZStream)) == 0
    if prime is not None:
        assert L.deflatePrime(C.byref(s), prime[0], prime[1]) == 0
    cap = len(data) + (len(data) >> 7) + 512
    out = C.create_string_buffer(cap)
    inb = C.create_string_buffer(data, max(len(data), 1))
    s.next_in = C.addressof(inb); s.avail_in = len(data); s.next_out = C.addressof(out); s.avail_out = cap
    rc = L.deflate(C.byref(s), R.Z_FINISH if last else R.Z_FULL_FLUSH)
    assert rc == (1 if last else 0) and s.avail_in == 0
    z = out.raw[: s.total_out]
    L.deflateEnd(C.byref(s))
    return z


def ref_primed(L, data, level, wbits, prime, mid=None):
    """The product's stream (mode B, SURVEY.md section 8c: the chunk function per 65536 bytes) with the primed bits in front of chunk 0 and,
    for `mid`, a second deflatePrime in front of chunk 1."""
    nchunks = max(1, (len(data) + R.CHUNK - 1) // R.CHUNK)
    parts = [R.zlib_header(level)] if wbits == 15 else []
    for k in range(nchunks):
        parts.append(ref_primed_chunk(L, data[k * R.CHUNK:(k + 1) * R.CHUNK], level, k + 1 == nchunks, prime if k == 0 else mid if k == 1 else None))
    if wbits == 15:
        parts.append(R.adler32(data).to_bytes(4, "big"))
    return b"".join(parts)


def ref_primed_one_stream(L, data, level, wbits, prime, mid=None):
    """ONE stream of the reference driven as tests/test_gpu_prime.py drives the product's z_stream API: deflatePrime, then 65536 bytes and Z_FULL_FLUSH per call
    (Z_FINISH on the last), `mid`: a second deflatePrime behind the first call.  What libzamd_z.so writes since round 4 (one continuous stream)."""
    s = R.ZStream()
    assert L.deflateInit2_(C.byref(s), level, 8, wbits, 8, 0, b"1.2.3", C.sizeof(R.ZStream)) == 0
    assert L.deflatePrime(C.byref(s), prime[0], prime[1]) == 0
    cap = len(data) + (len(data) >> 7) + 512
    out = C.create_string_buffer(cap)
    inb = C.create_string_buffer(data, max(len(data), 1))
    s.next_out = C.addressof(out); s.avail_out = cap
    nchunks = max(1, (len(data) + R.CHUNK - 1) // R.CHUNK)
    for k in range(nchunks):
        s.next_in = C.addressof(inb) + k * R.CHUNK; s.avail_in = min(R.CHUNK, len(data) - k * R.CHUNK)
        last = k + 1 == nchunks
        assert L.deflate(C.byref(s), R.Z_FINISH if last else R.Z_FULL_FLUSH) == (1 if last else 0)
        if k == 0 and mid is not None and not last:
            assert L.deflatePrime(C.byref(s), mid[0], mid[1]) == 0
    z = out.raw[: s.total_out]
    L.deflateEnd(C.byref(s))
    return z


def main():
    L = R.lib()
    L.deflatePrime.argtypes = [C.POINTER(R.ZStream), C.c_int, C.c_int]
    out = []
    for kind, n, seed in INPUTS:
        d = cases.make(kind, n, seed)
        for level in LEVELS:
            for i, prime in enumerate(PRIMES):
                wbits = -15 if (i + level) % 3 else 15
                mid = PRIMES[(i + 3) % len(PRIMES)] if (n > R.CHUNK and i % 2 == 0) else None
                z = ref_primed(L, d, level, wbits, prime, mid)
                za = ref_primed_one_stream(L, d, level, wbits, prime, mid)
                out.append(dict(kind=kind, n=n, seed=seed, level=level, wbits=wbits, prime=list(prime), mid=list(mid) if mid else None, len=len(z),
                                sha=hashlib.sha256(z).hexdigest()[:16], stream=z.hex() if len(z) <= 200 else None,
                                api_len=len(za), api_sha=hashlib.sha256(za).hexdigest()[:16]))
    with open(os.path.join(ROOT, "tests", "golden", "prime_kat.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("wrote prime_kat.json: %d cases" % len(out))


if __name__ == "__main__":
    main()
