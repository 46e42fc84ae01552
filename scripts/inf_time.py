"""Debug helper (GPU box): per-phase clock of the inflate kernel from a -DZGPU_INF_TIME build (ZAMD_GPU_LIB=build/variants/inftime.so)."""
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
offs = torch.empty(n + 1, dtype=torch.int64, device="cuda")
back = torch.empty(n * 65536, dtype=torch.uint8, device="cuda")
f = e.L.zgpu_debug_inf_time
f.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
out = (ctypes.c_ulonglong * 16)()
r = e.deflate_device(src.data_ptr(), n * 65536, int(os.environ.get("LEVEL", "6")), dst.data_ptr(), cap, flags=gpu.F_FINAL | gpu.F_ZLIB_WRAP, d_offsets=offs.data_ptr())
e.inflate_device(dst.data_ptr(), r.out_bytes, offs.data_ptr(), n, back.data_ptr(), n * 65536)
f(out, 1)
e.inflate_device(dst.data_ptr(), r.out_bytes, offs.data_ptr(), n, back.data_ptr(), n * 65536)
torch.cuda.synchronize()
f(out, 0)
assert torch.equal(src, back)
names = {0: "0 setup", 1: "R header+tables", 3: "W flush", 4: "W end", 8: "R lane decode", 9: "R walk+emit", 10: "W scan+literals", 11: "W matches", 12: "R one-symbol", 13: "R waits", 14: "W waits"}
tot = sum(int(out[i]) for i in names)
for i, nm in names.items():
    print("%-16s %8.1f us/chunk  %5.1f%%" % (nm, int(out[i]) / n / 100.0, 100.0 * int(out[i]) / tot))  # wall_clock64: 100 MHz
nl, nm_ = int(out[5]), int(out[6])
print("total %.1f us/chunk; rounds %.0f matches %.0f one-symbol steps %.0f per chunk; symbols phase %.1f ns/round" % (tot / n / 100.0, nl / n, nm_ / n, int(out[7]) / n, int(out[2]) * 10.0 / max(nl, 1)))
