#!/usr/bin/env python3
"""Golden hashes of the strategy row, produced by the compiled reference:  python oracle/gen_golden_strategies.py
->  tests/golden/strategy_kat.json  (sha256[:16] of the reference's raw chunk stream for not-last/last x pos0 off/on)."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import cases, refzlib as R  # noqa: E402
from test_strategies_cpu import kat_cases  # noqa: E402

out = {}
for strategy, level, kind, n, seed in kat_cases():
    d = cases.make(kind, n, seed)
    out["%d/%d/%s/%d/%d" % (strategy, level, kind, n, seed)] = [
        hashlib.sha256(R.deflate_chunk_raw(d, level, last, p0, strategy=strategy)).hexdigest()[:16] for last in (False, True) for p0 in (False, True)]
with open(os.path.join(ROOT, "tests", "golden", "strategy_kat.json"), "w") as f:
    json.dump(out, f, indent=0, sort_keys=True)
print("wrote strategy_kat.json:", len(out), "cases x 4")
