import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zlib_amd
from zlib_amd import gpu
from oracle import refzlib as R, corpus_py as CP
eng = zlib_amd.Engine(0)
F = gpu.F_FINAL | gpu.F_CONTINUOUS
for n in (70000, 100000, 140000, 200000, 400000, 1 << 20):
    d = CP.chunks(CP.KIND_SILESIA, 22, (n + 65535) // 65536).tobytes()[:n]
    want = R.deflate_calls(d, 6)
    for bt in ("1", "2", "1000"):
        os.environ["ZGPU_CONT_BATCH_TILES"] = bt
        got = eng.deflate_host(d, 6, flags=F)
        diffs = [i for i in range(min(len(got), len(want))) if got[i] != want[i]]
        print(n, "batch", bt, len(want), len(got), "ndiff", len(diffs), diffs[:8], [(hex(got[i]), hex(want[i])) for i in diffs[:4]], flush=True)
