"""Debug helper (GPU box): counters of walk_kernel from a -DZGPU_WALK_STATS build (ZAMD_GPU_LIB=build/variants/wstats.so)."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zlib_amd
from zlib_amd import gpu

lvl = int(sys.argv[1]) if len(sys.argv) > 1 else 6
kind = int(sys.argv[2]) if len(sys.argv) > 2 else 0
e = zlib_amd.Engine(0)
n = 4096
src = torch.empty(n * 65536, dtype=torch.uint8, device="cuda")
e.corpus_fill_device(kind, 0x5EED5117 if kind == 0 else 0x10C7E47, 0, n, src.data_ptr())
cap = e.L.zgpu_deflate_bound(n * 65536, 65536)
dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
f = e.L.zgpu_debug_walk_stats
f.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
out = (ctypes.c_ulonglong * 8)()
f(out, 1)
e.deflate_device(src.data_ptr(), n * 65536, lvl, dst.data_ptr(), cap, flags=gpu.F_FINAL)
torch.cuda.synchronize()
f(out, 0)
bodies, act, passes, served, searches, folds, parked, limbo = [int(out[i]) for i in range(8)]
print("level %d kind %d, per chunk:" % (lvl, kind))
print("  searches %.0f (%.3f per byte), %.1f%% started from memory one pass late" % (searches / n, searches / n / 65536, 100.0 * limbo / max(searches, 1)))
print("  bodies %.0f   candidate steps %.0f (%.2f per byte), lane utilisation %.1f%%" % (bodies / n, act / n, act / n / 65536, 100.0 * act / max(bodies * 256, 1)))
print("  passes %.0f (%.1f lanes served each)   folds %.0f (%.1f parked each; %.3f parked per byte)" % (passes / n, served / max(passes, 1), folds / n, parked / max(folds, 1), parked / n / 65536))
