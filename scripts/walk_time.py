"""Debug helper (GPU box): where a wave of walk_kernel spends its cycles, from a -DZGPU_WALK_TIME build (ZAMD_GPU_LIB=build/variants/wtime.so)."""
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
n = 16384
src = torch.empty(n * 65536, dtype=torch.uint8, device="cuda")
e.corpus_fill_device(kind, 0x5EED5117 if kind == 0 else 0x10C7E47, 0, n, src.data_ptr())
cap = e.L.zgpu_deflate_bound(n * 65536, 65536)
dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
f = e.L.zgpu_debug_walk_time
f.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
out = (ctypes.c_ulonglong * 8)()
e.deflate_device(src.data_ptr(), n * 65536, lvl, dst.data_ptr(), cap, flags=gpu.F_FINAL)
f(out, 1)
e.deflate_device(src.data_ptr(), n * 65536, lvl, dst.data_ptr(), cap, flags=gpu.F_FINAL)
torch.cuda.synchronize()
f(out, 0)
names = ["pass: starts", "bodies (rest)", "folds in bodies", "bodies: byte reads", "pass: drain", "pass: parse", "pass: blocks", "bodies: take-over wait"]
tot = sum(int(out[i]) for i in range(8))
for i, nm in enumerate(names):
    print("%-16s %9.0f cycles per wave and chunk  %5.1f%%" % (nm, int(out[i]) / n / 8, 100.0 * int(out[i]) / tot))
print("total %.0f cycles per wave and chunk" % (tot / n / 8))
