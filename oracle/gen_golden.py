#!/usr/bin/env python3
"""Generate tests/golden/*.json from the REAL reference (oracle/_ref/libzref.so, compiled from
/root/reference by oracle/Makefile).  Run in the development container:

    make -C oracle && python oracle/gen_golden.py

The fixtures are data only: inputs are regenerated from seeds (oracle/cases.py, zlib_amd/csrc/corpus.h),
expected outputs are stored as hex (small) or length + SHA-256 prefix (large).  TEST INFRASTRUCTURE.
"""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cases, corpus_py as CP, refzlib as R  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def h16(b: bytes) -> str:
    return hashlib.sha256(b).hexdigest()[:16]


def enc(b: bytes):
    """complete bytes when short, else [length, sha prefix]"""
    return b.hex() if len(b) <= 96 else [len(b), h16(b)]


def dump(name, obj):
    with open(os.path.join(OUT, name), "w") as f:
        json.dump(obj, f, separators=(",", ":"), sort_keys=True)
        f.write("\n")
    print("wrote", name, os.path.getsize(os.path.join(OUT, name)), "bytes")


def kat():
    d = {"reference": "ChrisHird/ZLIB zlib " + R.version(), "hello": {}, "hello_1mib": {}}
    for lvl in (-1, 0, 1, 6, 9):
        d["hello"][str(lvl)] = R.compress2(cases.HELLO, lvl).hex()
    big = cases.hello_1mib()
    d["hello_1mib"]["adler32"] = "%08x" % R.adler32(big)
    d["hello_1mib"]["sha256_input"] = hashlib.sha256(big).hexdigest()
    for lvl in (1, 6, 9):
        c = R.compress2(big, lvl)
        b = R.deflate_mode_b(big, lvl)
        a = R.deflate_mode_a(big, lvl)
        d["hello_1mib"][str(lvl)] = {
            "compress2_len": len(c), "compress2_sha256": hashlib.sha256(c).hexdigest(), "compress2_head": c[:16].hex(),
            "mode_b_len": len(b), "mode_b_sha256": hashlib.sha256(b).hexdigest(),
            "mode_a_len": len(a), "mode_a_sha256": hashlib.sha256(a).hexdigest()}
    d["compressBound"] = {str(n): R.lib().compressBound(n) for n in (0, 1, 65536, 1 << 20)}
    d["adler32_combine"] = [[a, b, n, R.lib().adler32_combine(a, b, n)] for (a, b, n) in
                            [(1, 1, 0), (0x00620062, 0x00630063, 1), (0xc08f758f, 0x12345678, 65536),
                             (0xfff0fff0, 0xfff0fff0, 65520), (0x0001fff1, 0x00010000, 5), (65520 | (3 << 16), 2 | (7 << 16), 123456789)]]
    dump("kat.json", d)


def small():
    d = {}
    for name, data in cases.small_cases():
        e = {}
        for lvl in range(0, 10):
            for last in (0, 1):
                e["L%d-last%d" % (lvl, last)] = enc(R.deflate_chunk_raw(data, lvl, bool(last)))
                if lvl > 0:
                    e["L%d-last%d-p0" % (lvl, last)] = enc(R.deflate_chunk_raw(data, lvl, bool(last), True))
        d[name] = e
    dump("chunk_small.json", d)
    d = {}
    for name, data in cases.big_cases():
        e = {}
        for lvl in range(0, 10):
            for last in (0, 1):
                o = R.deflate_chunk_raw(data, lvl, bool(last))
                e["L%d-last%d" % (lvl, last)] = [len(o), h16(o)]
            if lvl in (1, 6, 9):
                o = R.deflate_chunk_raw(data, lvl, False, True)
                e["L%d-last0-p0" % lvl] = [len(o), h16(o)]
        d[name] = e
    dump("chunk_big.json", d)


def corpus(kind, total_chunks, nsample, fname):
    t0 = time.time()
    ids = [k * (total_chunks // nsample) + (k % (total_chunks // nsample)) % 16 for k in range(nsample)]
    rows = []
    for i in ids:
        data = CP.chunk(kind, i)
        row = [i, h16(data)]
        for lvl in (1, 6, 9):
            o = R.deflate_chunk_raw(data, lvl, False)
            row += [len(o), h16(o)]
        rows.append(row)
    dump(fname, {"kind": kind, "seed": CP.default_seed(kind), "total_chunks": total_chunks,
                 "columns": ["chunk", "sha_in", "len_L1", "sha_L1", "len_L6", "sha_L6", "len_L9", "sha_L9"],
                 "note": "reference F(chunk, level, pos0_matchable=0, is_last=0); sha = first 16 hex of SHA-256",
                 "rows": rows})
    print("  %.1fs" % (time.time() - t0))


def inflate_errors():
    """Corrupted raw-deflate streams with the reference's verdict (return code, message, bytes produced)."""
    g = cases.Lcg(4242)
    rows = []
    for kind in ("text", "rand", "runs"):
        for n in (50, 700):
            data = cases.make(kind, n, 5)
            for lvl in (0, 1, 6, 9):
                raw = R.deflate_chunk_raw(data, lvl, True)
                for _ in range(12):
                    c = bytearray(raw)
                    for _ in range(1 + g.below(2)):
                        c[g.below(min(len(c), 200))] ^= 1 << g.below(8)
                    rc, out, used, msg = R.inflate_raw(bytes(c), n + 64)
                    rows.append([bytes(c).hex(), n + 64, rc, msg, h16(out) if rc == 1 else None, len(out) if rc == 1 else None])
                for cut in (1, 3):
                    t = raw[:-cut]
                    rc, out, used, msg = R.inflate_raw(t, n + 64)
                    rows.append([t.hex(), n + 64, rc, msg, None, None])
    dump("inflate_cases.json", {"columns": ["stream_hex", "out_cap", "rc", "msg", "sha_out", "len_out"], "rows": rows})


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "quick":  # everything except the corpus samples
        kat(); small(); inflate_errors(); sys.exit(0)
    kat()
    small()
    inflate_errors()
    corpus(CP.KIND_SILESIA, 65536, 4096, "corpus_silesia.json")
    corpus(CP.KIND_LOGTEXT, 1048576, 512, "corpus_logtext.json")
