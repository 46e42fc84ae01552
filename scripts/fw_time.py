"""Debug helper (GPU box): where the wave of fastwin_kernel (levels 1-3) spends its cycles, from a -DZGPU_FW_TIME build
(scripts/build_variant.sh fwtime -DZGPU_FW_TIME; ZAMD_GPU_LIB=build/variants/fwtime.so python scripts/fw_time.py [level kind nchunks])."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zlib_amd
from zlib_amd import gpu

lvl = int(sys.argv[1]) if len(sys.argv) > 1 else 1
kind = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
e = zlib_amd.Engine(0)
src = torch.empty(n * 65536, dtype=torch.uint8, device="cuda")
e.corpus_fill_device(kind, 0x5EED5117 if kind == 0 else 0x10C7E47, 0, n, src.data_ptr())
cap = e.L.zgpu_deflate_bound(n * 65536, 65536)
dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
f = e.L.zgpu_debug_fw_time
f.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
out = (ctypes.c_ulonglong * 16)()
e.deflate_device(src.data_ptr(), n * 65536, lvl, dst.data_ptr(), cap, flags=gpu.F_FINAL)
f(out, 1)
e.deflate_device(src.data_ptr(), n * 65536, lvl, dst.data_ptr(), cap, flags=gpu.F_FINAL)
torch.cuda.synchronize()
f(out, 0)
names = ["window top: staging, loads, ring", "round top", "evaluation", "walk", "clears + stale check", "tokens", "whole-bucket search"]
tot = sum(int(out[i]) for i in range(7))
win = int(out[8]) / n
print("level %d kind %d, %d chunks: %.0f windows a chunk, %.2f rounds and %.2f evaluations a window, %.1f measured matches and %.1f whole-bucket searches a chunk"
      % (lvl, kind, n, win, int(out[9]) / max(1, int(out[8])), int(out[10]) / max(1, int(out[8])), int(out[11]) / n, int(out[12]) / n))
for i, nm in enumerate(names):
    print("%-34s %9.0f cycles per chunk  %6.0f per window  %5.1f%%" % (nm, int(out[i]) / n, int(out[i]) / n / win, 100.0 * int(out[i]) / tot))
print("total %.0f cycles per chunk, %.0f per window" % (tot / n, tot / n / win))
