#!/usr/bin/env python3
"""Golden vectors for chunks whose token count is a multiple of lit_bufsize - 1 = 16383, produced by the compiled reference:
    python oracle/gen_golden_fullblock.py   ->  tests/golden/fullblock_kat.json
deflate_fast flushes the full block and the end of the chunk adds an empty one; deflate_slow tallies the last byte's literal behind its loop without
looking at "buffer full" (qcsrc/deflate.c:1660-1665), so the full block is itself the last one.  Inputs: cases.nomatch (all literals) of 16383 * k
bytes and one byte either side; cases.nomatch twice (the second half is one long run of matches) for token counts that are reached with a match.
TEST INFRASTRUCTURE ONLY."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cases, refzlib as R  # noqa: E402

SIZES = [16382, 16383, 16384, 32766, 32767]
LEVELS = [1, 2, 3, 4, 5, 6, 8, 9]


def inputs():
    for n in SIZES:
        yield "nomatch/%d" % n, cases.nomatch(n)
    base = cases.nomatch(16380)
    for tail in (b"", b"xyz", b"q"):  # 16380 literals, then matches, then a few literals: the count crosses 16383 in different states
        yield "nomatch16380+copy+%d" % len(tail), base + base[:9000] + tail + cases.nomatch(40)[30:]


def main():
    out = []
    for name, d in inputs():
        for level in LEVELS:
            zs = [R.deflate_chunk_raw(d, level, last) for last in (False, True)]
            out.append(dict(name=name, n=len(d), level=level, len=[len(z) for z in zs], sha=[hashlib.sha256(z).hexdigest()[:16] for z in zs]))
    with open(os.path.join(ROOT, "tests", "golden", "fullblock_kat.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("wrote fullblock_kat.json: %d cases" % len(out))


if __name__ == "__main__":
    main()
