"""Does the lane-per-chunk loop (random accesses over 64 GiB of tables) care what else the process holds?  Level 1, 4 GiB, with and without other allocations made first."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, zlib_amd
from zlib_amd import gpu
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
hold = []
if mode == "dummy":
    hold.append(torch.empty(100 << 30, dtype=torch.uint8, device="cuda"))
e = zlib_amd.Engine(0)
n = 65536
src = torch.empty(n * 65536, dtype=torch.uint8, device="cuda")
e.corpus_fill_device(0, 0x5EED5117, 0, n, src.data_ptr())
cap = e.L.zgpu_deflate_bound(n * 65536, 65536)
dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
def run(lvl):
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        e.deflate_device(src.data_ptr(), n * 65536, lvl, dst.data_ptr(), cap, flags=gpu.F_FINAL)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
if mode == "l6first":
    print("level 6 first: %.1f ms" % run(6))
print("%s: level 1 %.1f ms" % (mode, run(1)))
print("free/total GiB:", [x >> 30 for x in torch.cuda.mem_get_info()])
