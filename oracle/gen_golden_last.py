#!/usr/bin/env python3
"""Add `last_rows` to tests/golden/corpus_*.json: the REFERENCE's output for the chunks a run ends on, with is_last = 1 (BFINAL in the last block, no
flush marker) -- what the stream's very last chunk looks like, which `rows` (is_last = 0) cannot check.  Same columns as `rows`.  The chunks: the last
of the 4 GiB silesia-mix workload and of its 256 MiB / 1 GiB prefixes; the last of N x 8 GiB of log-text for N = 1, 2, 4, 8 (config 5: the last rank's
last chunk).  From the compiled reference (oracle/_ref/libzref.so):  make -C oracle && python oracle/gen_golden_last.py.  TEST INFRASTRUCTURE.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import corpus_py as CP, refzlib as R  # noqa: E402
from oracle.gen_golden import h16, OUT  # noqa: E402


def add(fname, kind, chunks):
    path = os.path.join(OUT, fname)
    g = json.load(open(path))
    assert g["kind"] == kind
    rows = []
    for i in chunks:
        data = CP.chunk(kind, i)
        row = [i, h16(data)]
        for lvl in (1, 6, 9):
            o = R.deflate_chunk_raw(data, lvl, True)
            row += [len(o), h16(o)]
        rows.append(row)
    g["last_rows"] = rows
    g["note_last"] = "last_rows: the same columns with is_last=1 (the chunk a stream ends on)"
    with open(path, "w") as f:
        json.dump(g, f, separators=(",", ":"), sort_keys=True)
        f.write("\n")
    print(fname, rows)


if __name__ == "__main__":
    add("corpus_silesia.json", CP.KIND_SILESIA, [4095, 16383, 65535])
    add("corpus_logtext.json", CP.KIND_LOGTEXT, [131072 * n - 1 for n in (1, 2, 4, 8)])
