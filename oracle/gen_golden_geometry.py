#!/usr/bin/env python3
"""Golden vectors of deflateInit2's geometry (qcsrc/deflate.c:222-297: windowBits 9..15, memLevel 1..9), produced by the compiled reference:
    python oracle/gen_golden_geometry.py   ->  tests/golden/geometry_kat.json

chunk   the reference's chunk function (a fresh raw stream, Z_FULL_FLUSH or Z_FINISH behind the chunk) under every windowBits x memLevel, levels
        0 1 2 3 4 6 9, four inputs each (rotating through the list below): length and sha256[:16] of both endings
stream  whole streams through deflateInit2 (zlib wrapper: the header's CINFO is the window size; raw; gzip), mode B (the chunk function per 65536
        bytes), 150 000 bytes, a handful of geometries
dict    the chunk function behind deflateSetDictionary with small windows (the dictionary is cut to MAX_DIST)
TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cases, refzlib as R  # noqa: E402

INPUTS = [("text", 65536, 11), ("mix", 65536, 12), ("rand", 65536, 13), ("runs", 65536, 14), ("ab", 40000, 15), ("period", 65536, 16), ("zeros", 65536, 17),
          ("text", 300, 18), ("mix", 20000, 19), ("rand", 1, 20), ("text", 0, 21), ("runs", 33000, 22)]
LEVELS = [0, 1, 2, 3, 4, 6, 9]
STREAM_GEOS = [(9, 1), (9, 9), (10, 4), (12, 8), (13, 9), (14, 2), (15, 9), (15, 1), (15, 7), (8, 8)]


def h16(b):
    return hashlib.sha256(b).hexdigest()[:16]


def ref_chunk(L, data, level, wbits, mem, last, dictionary=None):
    s = R.ZStream()
    assert L.deflateInit2_(C.byref(s), level, 8, -wbits, mem, 0, b"1.2.3", C.sizeof(R.ZStream)) == 0
    if dictionary is not None:
        assert L.deflateSetDictionary(C.byref(s), dictionary, len(dictionary)) == 0
    cap = len(data) + (len(data) >> 2) + 4096
    out = C.create_string_buffer(cap)
    inb = C.create_string_buffer(data, max(len(data), 1))
    s.next_in = C.addressof(inb); s.avail_in = len(data); s.next_out = C.addressof(out); s.avail_out = cap
    rc = L.deflate(C.byref(s), R.Z_FINISH if last else R.Z_FULL_FLUSH)
    assert rc == (1 if last else 0) and s.avail_in == 0 and s.avail_out > 0
    z = out.raw[: s.total_out]
    L.deflateEnd(C.byref(s))
    return z


def chunk_inputs(wbits, mem, level):
    k = (wbits * 9 + mem + level * 5) % len(INPUTS)
    return [INPUTS[(k + 3 * i) % len(INPUTS)] for i in range(4)]


def zlib_header(level, wbits):
    hdr = (8 + ((wbits - 8) << 4)) << 8
    hdr |= (0 if level < 2 else 1 if level < 6 else 2 if level == 6 else 3) << 6
    hdr += 31 - hdr % 31
    return bytes([hdr >> 8, hdr & 255])


def ref_stream(L, data, level, wbits, mem, wrap):
    """mode B with the wrapper the reference writes for this deflateInit2 (checked below against the reference's own one-chunk stream)."""
    w = 9 if wbits == 8 else wbits
    n = len(data)
    nchunks = max(1, (n + R.CHUNK - 1) // R.CHUNK)
    body = b"".join(ref_chunk(L, data[k * R.CHUNK:(k + 1) * R.CHUNK], level, w, mem, k + 1 == nchunks) for k in range(nchunks))
    if wrap == "raw":
        return body
    if wrap == "zlib":
        return zlib_header(level, w) + body + R.adler32(data).to_bytes(4, "big")
    xfl = 2 if level == 9 else 4 if level < 2 else 0
    return bytes([31, 139, 8, 0, 0, 0, 0, 0, xfl, 3]) + body + R.crc32(data).to_bytes(4, "little") + (n & 0xFFFFFFFF).to_bytes(4, "little")


def ref_whole(L, data, level, wbits_arg, mem):
    s = R.ZStream()
    assert L.deflateInit2_(C.byref(s), level, 8, wbits_arg, mem, 0, b"1.2.3", C.sizeof(R.ZStream)) == 0
    cap = len(data) + (len(data) >> 2) + 4096
    out = C.create_string_buffer(cap)
    inb = C.create_string_buffer(data, max(len(data), 1))
    s.next_in = C.addressof(inb); s.avail_in = len(data); s.next_out = C.addressof(out); s.avail_out = cap
    assert L.deflate(C.byref(s), R.Z_FINISH) == 1
    z = out.raw[: s.total_out]
    L.deflateEnd(C.byref(s))
    return z


def main():
    L = R.lib()
    out = {"chunk": [], "stream": [], "dict": []}
    for wbits in range(9, 16):
        for mem in range(1, 10):
            for level in LEVELS:
                for kind, n, seed in chunk_inputs(wbits, mem, level):
                    d = cases.make(kind, n, seed)
                    zs = [ref_chunk(L, d, level, wbits, mem, last) for last in (False, True)]
                    out["chunk"].append(dict(w=wbits, m=mem, level=level, kind=kind, n=n, seed=seed, len=[len(z) for z in zs], sha=[h16(z) for z in zs]))
    data = cases.make("mix", 150000, 31)
    one = cases.make("text", 9000, 32)
    for wbits, mem in STREAM_GEOS:
        for level in (0, 1, 6):
            for wrap, arg in (("zlib", wbits), ("raw", -wbits), ("gzip", wbits + 16)):
                # the wrapper bytes this generator puts around mode B are the reference's own: a one-chunk stream is the same in both modes
                assert ref_stream(L, one, level, wbits, mem, wrap) == ref_whole(L, one, level, arg, mem), (wbits, mem, level, wrap)
                z = ref_stream(L, data, level, wbits, mem, wrap)
                out["stream"].append(dict(w=wbits, m=mem, level=level, wrap=wrap, len=len(z), sha=h16(z), head=z[:10].hex()))
    dic = cases.make("text", 40000, 33)
    for wbits, mem in ((9, 8), (11, 3), (13, 9), (15, 9), (15, 5)):
        for level in (0, 1, 4, 9):
            for dl in (100, 600, 40000):
                dd = dic[:dl]
                keep = min(dl, (1 << wbits) - 262)
                d = cases.make("text", 65536 - keep, 34)  # (the first chunk behind a dictionary takes what is left of 64 KiB)
                zs = [ref_chunk(L, d, level, wbits, mem, last, dd) for last in (False, True)]
                out["dict"].append(dict(w=wbits, m=mem, level=level, dict=dl, n=len(d), len=[len(z) for z in zs], sha=[h16(z) for z in zs]))
    with open(os.path.join(ROOT, "tests", "golden", "geometry_kat.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("wrote geometry_kat.json: %d chunk cases, %d streams, %d dictionary cases" % (len(out["chunk"]), len(out["stream"]), len(out["dict"])))


if __name__ == "__main__":
    main()
