"""Writes tests/golden/continuous_kat.json from the COMPILED REFERENCE (oracle/_ref/libzref.so): ONE continuous zlib stream -- what compress2() /
deflate(Z_NO_FLUSH ... Z_FINISH) of the reference emits (qcsrc/compress.c:22-58, qcsrc/deflate.c:552-856) -- for un-flushed inputs of 65537 bytes, 1 MiB and
16 MiB of both synthetic corpora (zlib_amd/csrc/corpus.h) at levels 1, 4, 6, 9 (and 0, 2, 3, 5, 7, 8 at the two small sizes), and the same inputs with one
Z_SYNC_FLUSH in the middle (the window is kept across it, deflate.c:808-819).  Stored: length and SHA-256 of every stream, the streams' first 32 bytes.

TEST INFRASTRUCTURE: run where /root/reference is mounted (`python oracle/gen_golden_continuous.py`); the tests only read the JSON."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import corpus_py as CP, refzlib as R  # noqa: E402

SIZES = (65537, 1 << 20, 16 << 20)
SEEDS = {0: 0x5EED5117, 1: 0x10C7E47}


def corpus(kind, nbytes):
    """(the rows of sizes up to 16 MiB: the chunks from index SEEDS[kind] on of the corpus with its default seed -- the second argument of CP.chunks is
    the first chunk; tests/test_gpu_continuous.py builds its inputs with the same call)"""
    return CP.chunks(kind, SEEDS[kind], (nbytes + 65535) // 65536).tobytes()[:nbytes]


def bench_rows():
    """the bench's check (bench.py extra.continuous): the first 256 MiB of the headline workload -- chunks 0 .. 4095 of the silesia-mix corpus -- as one stream"""
    big = CP.chunks(0, 0, (256 << 20) // 65536).tobytes()
    rows = []
    for level in (6, 1):
        z = R.deflate_calls(big, level, (), wbits=15)
        rows.append({"corpus": 0, "first_chunk": 0, "n": len(big), "level": level, "sync_at": None, "len": len(z), "sha256": hashlib.sha256(z).hexdigest(), "head": z[:32].hex()})
        print(0, len(big), level, None, len(z), flush=True)
    return rows


def main():
    rows = []
    for kind in (0, 1):
        for n in SIZES:
            data = corpus(kind, n)
            levels = (1, 4, 6, 9) if n > (1 << 20) else (0, 1, 2, 3, 4, 5, 6, 7, 8, 9)
            for level in levels:
                for sync_at in (None, n // 2 + 1234):
                    calls = () if sync_at is None else ((sync_at, R.Z_SYNC_FLUSH),)
                    z = R.deflate_calls(data, level, calls, wbits=15)
                    rows.append({"corpus": kind, "n": n, "level": level, "sync_at": sync_at, "len": len(z), "sha256": hashlib.sha256(z).hexdigest(), "head": z[:32].hex()})
                    print(kind, n, level, sync_at, len(z), flush=True)
    rows += bench_rows()
    out = {"reference": R.version(), "seeds": {str(k): v for k, v in SEEDS.items()}, "rows": rows}
    with open(os.path.join(ROOT, "tests", "golden", "continuous_kat.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", len(rows), "rows")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "bench":  # only the two 256 MiB rows again, the others as they are in the file
        path = os.path.join(ROOT, "tests", "golden", "continuous_kat.json")
        g = json.load(open(path))
        g["rows"] = [r for r in g["rows"] if r["n"] != (256 << 20)] + bench_rows()
        with open(path, "w") as f:
            json.dump(g, f, indent=0)
    else:
        main()
