"""Debug helper (GPU box): counters of the lockstep match kernel from a -DZGPU_M3_STATS build (ZAMD_GPU_LIB=build/variants/stats.so)."""
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
f = e.L.zgpu_debug_m3_stats
f.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
out = (ctypes.c_ulonglong * 8)()
f(out, 1)
e.deflate_device(src.data_ptr(), n * 65536, lvl, dst.data_ptr(), cap, flags=gpu.F_FINAL)
torch.cuda.synchronize()
f(out, 0)
steps, act, ent, folds, it0, pos, itw = [int(out[i]) for i in (0, 1, 2, 3, 4, 5, 6)]
blocks = pos / 64
print("level %d kind %d: positions %d" % (lvl, kind, pos))
print("  wave-steps/block %.1f   active lanes/step %.1f (%.0f%%)   candidates/position %.1f" % (steps / blocks, act / steps, 100 * act / steps / 64, act / pos))
print("  ring entries/position %.2f (%.1f%% of candidates)   folds/block %.1f   entries/fold %.1f" % (ent / pos, 100 * ent / act, folds / blocks, ent / max(folds, 1)))
print("  compare iterations: wave %.1f per fold, lane-0 %.2f per fold" % (itw / max(folds, 1), it0 / max(folds, 1)))
