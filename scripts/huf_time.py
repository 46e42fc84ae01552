"""Debug helper (GPU box): per-phase clock of the Huffman kernel from a -DZGPU_HUF_TIME build (ZAMD_GPU_LIB=build/variants/huftime.so)."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zlib_amd
from zlib_amd import gpu

e = zlib_amd.Engine(0)
n = 16384
src = torch.empty(n * 65536, dtype=torch.uint8, device="cuda")
e.corpus_fill_device(0, 0x5EED5117, 0, n, src.data_ptr())
cap = e.L.zgpu_deflate_bound(n * 65536, 65536)
dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
f = e.L.zgpu_debug_huf_time
f.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
out = (ctypes.c_ulonglong * 8)()
e.deflate_device(src.data_ptr(), n * 65536, 6, dst.data_ptr(), cap, flags=gpu.F_FINAL)
f(out, 1)
e.deflate_device(src.data_ptr(), n * 65536, 6, dst.data_ptr(), cap, flags=gpu.F_FINAL)
torch.cuda.synchronize()
f(out, 0)
names = ["histogram", "L and D trees (two lanes)", "bit-length tree, block type (one lane)", "block header (one lane)", "token bits (all lanes)"]
tot = sum(int(out[i]) for i in range(5))
for i, nm in enumerate(names):
    print("%-40s %8.1f us/chunk  %5.1f%%" % (nm, int(out[i]) / n / 100.0, 100.0 * int(out[i]) / tot))  # wall_clock64: 100 MHz
print("total %.1f us/chunk" % (tot / n / 100.0))
