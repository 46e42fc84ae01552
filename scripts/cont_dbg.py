import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zlib_amd
from zlib_amd import gpu
from oracle import refzlib as R, corpus_py as CP
eng = zlib_amd.Engine(0)
F = gpu.F_FINAL | gpu.F_CONTINUOUS
level = int(sys.argv[1]) if len(sys.argv) > 1 else 3
d0 = CP.chunks(CP.KIND_SILESIA, 3, 8).tobytes()
for n in (100000, 130000, 131073, 162560, 200000, 250000, 300000):
    d = d0[:n]
    want = R.deflate_calls(d, level)
    os.environ["ZGPU_FAST_TRACE"] = "1"
    got = eng.deflate_host(d, level, flags=F)
    k = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), -1)
    print(n, "level", level, len(want), len(got), "first diff", k, flush=True)
