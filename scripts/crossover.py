"""Levels 1-3: the wave-per-chunk kernel alone against the lane-per-chunk loop with hand-on, by the size of the call (device buffers, Silesia-mix)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, zlib_amd
from zlib_amd import gpu
e = zlib_amd.Engine(0)
nmax = 65536
src = torch.empty(nmax * 65536, dtype=torch.uint8, device="cuda")
e.corpus_fill_device(int(os.environ.get("KIND", "0")), 0x5EED5117, 0, nmax, src.data_ptr())
cap = e.L.zgpu_deflate_bound(nmax * 65536, 65536)
dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
def run(n, lvl, impl, env):
    if env is None: os.environ.pop("ZGPU_HAND_ON", None)
    else: os.environ["ZGPU_HAND_ON"] = env
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        e.deflate_device(src.data_ptr(), n * 65536, lvl, dst.data_ptr(), cap, flags=gpu.F_FINAL, lz_impl=impl)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
for lvl in [int(x) for x in os.environ.get("LEVELS", "1,2,3").split(",")]:
    for n in [int(x) for x in os.environ.get("SIZES", "4096,8192,16384,24576,32768,49152,65536").split(",")]:
        a = run(n, lvl, gpu.LZ_FASTWIN, None); b = run(n, lvl, gpu.LZ_AUTO, "2"); c = run(n, lvl, gpu.LZ_SERIAL, None)
        print("level %d  %5d chunks: wave kernel %7.1f ms   loop + hand-on %7.1f ms   loop alone %7.1f ms" % (lvl, n, a, b, c), flush=True)
