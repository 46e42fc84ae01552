#!/usr/bin/env python3
"""Golden vectors of the gzip / CRC-32 row, produced by the compiled reference (oracle/_ref):

    python oracle/gen_golden_gzip.py      ->  tests/golden/gzip_kat.json

crc32() of seeded inputs, crc32_combine() vectors, the gzip member deflate() writes for windowBits 31 on single-chunk
inputs, and mode-B gzip members of multi-chunk inputs (the reference's own 10-byte header, its raw chunk streams, its CRC)."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cases, corpus_py as CP, refzlib as R  # noqa: E402


def mode_b_gzip(data, level, chunk=65536):
    head = R.deflate_wbits(b"x", level, 31)[:10]  # the header bytes as the reference writes them at this level
    n = len(data)
    nchunks = max(1, (n + chunk - 1) // chunk)
    body = b"".join(R.deflate_chunk_raw(data[k * chunk:(k + 1) * chunk], level, k == nchunks - 1) for k in range(nchunks))
    return head + body + R.crc32(data).to_bytes(4, "little") + (n & 0xFFFFFFFF).to_bytes(4, "little")


def main():
    out = {"crc32": {}, "combine": [], "single": {}, "multi": {}}
    g = cases.Lcg(4242)
    for kind in cases.KINDS:
        for n in (0, 1, 2, 3, 7, 255, 256, 257, 4095, 65535, 65536, 65537, 200000):
            out["crc32"]["%s/%d" % (kind, n)] = R.crc32(cases.make(kind, n, 5))
    for _ in range(64):
        c1, c2 = g.below(1 << 32), g.below(1 << 32)
        ln = [0, 1, 2, 255, 65536, 65537, g.below(1 << 20), g.below(1 << 31), (1 << 32) + g.below(1 << 20)][g.below(9)]
        out["combine"].append([c1, c2, ln, R.crc32_combine(c1, c2, ln)])
    for kind in ("text", "rand", "zeros", "mix"):
        for n in (0, 1, 14, 1000, 65536):
            data = cases.make(kind, n, 9)
            for lvl in (0, 1, 6, 9):
                z = R.deflate_wbits(data, lvl, 31)
                out["single"]["%s/%d/%d" % (kind, n, lvl)] = z.hex() if len(z) <= 64 else [len(z), hashlib.sha256(z).hexdigest()[:16]]
    multi = {"corpus0x5": CP.chunks(0, 7, 5).tobytes(), "corpus1x3-ragged": CP.chunks(1, 2, 3).tobytes()[:-4321], "hello": cases.hello_1mib()[:300000]}
    for name, data in multi.items():
        for lvl in (1, 6, 9):
            z = mode_b_gzip(data, lvl)
            rc, back, used, msg, adler = R.inflate_wbits(z, 31, len(data) + 16)  # the reference reads it back, CRC and length checked
            assert rc == 1 and back == data and used == len(z) and adler == R.crc32(data), (name, lvl, rc, msg)
            out["multi"]["%s/%d" % (name, lvl)] = [len(z), hashlib.sha256(z).hexdigest()[:16], R.crc32(data)]
    with open(os.path.join(ROOT, "tests", "golden", "gzip_kat.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("wrote gzip_kat.json: %d crc, %d combine, %d single, %d multi" % (len(out["crc32"]), len(out["combine"]), len(out["single"]), len(out["multi"])))


if __name__ == "__main__":
    main()
