import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import zlib_amd
from zlib_amd import gpu
from oracle import refzlib as R, corpus_py as CP
import cont_feed as CF
eng = zlib_amd.Engine(0)
rnd = random.Random(11)
d = CF.corpus(CP.KIND_SILESIA, 11, 3 << 20)
tests = []
for it in range(6):
    calls = []; pos = 0
    while True:
        pos += rnd.choice([1, 5, 100, 4096, 32768 - 262 + rnd.randrange(0, 300), 65536, 200000])
        if pos >= len(d): break
        calls.append((pos, rnd.choice([0, 0, 0, 1, 2, 3])))
    tests.append((tuple(calls), rnd.choice([70000, 200000, 1 << 62])))
level = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for calls, more_at in tests:
    if more_at > (1 << 40): continue
    lo, hi = 0, len(calls)
    def bad(k):
        cc = calls[:k]
        n = cc[-1][0] + 1000 if cc else 1000
        dd = d[:n]
        return CF.stream(eng, dd, level, cc, more_at=more_at) != R.deflate_calls(dd, level, cc)
    if not bad(hi):
        print("pattern ok", len(calls), more_at); continue
    while hi - lo > 1:
        mid = (lo + hi) // 2
        if bad(mid): hi = mid
        else: lo = mid
    print("level", level, "more_at", more_at, "first failing prefix: %d calls:" % hi, calls[max(0, hi - 6):hi], flush=True)
