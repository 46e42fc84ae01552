#!/usr/bin/env python3
"""Golden hashes of the preset-dictionary row, by the compiled reference:  python oracle/gen_golden_dict.py -> tests/golden/dict_kat.json"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import refzlib as R  # noqa: E402
from test_dictionary_cpu import dict_cases  # noqa: E402

out = {}
for key, lvl, strat, d, x in dict_cases():
    out[key] = [hashlib.sha256(R.deflate_chunk_dict_raw(d, x, lvl, last, strat)).hexdigest()[:16] for last in (False, True)]
with open(os.path.join(ROOT, "tests", "golden", "dict_kat.json"), "w") as f:
    json.dump(out, f, indent=0, sort_keys=True)
print("wrote dict_kat.json:", len(out), "cases x 2")
