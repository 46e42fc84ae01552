"""GPU box, under rocprofv3 --kernel-trace --memory-copy-trace: three zgpu_deflate_host calls over 1 GiB (the timeline shows what overlaps)."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, zlib_amd
from zlib_amd import gpu
e = zlib_amd.Engine(0)
n = 16384
src = torch.empty(n * 65536, dtype=torch.uint8, device="cuda")
e.corpus_fill_device(0, 0x5EED5117, 0, n, src.data_ptr())
host = src.cpu().numpy()
cap = e.L.zgpu_deflate_bound(host.size, 65536)
zbuf = np.zeros(cap, dtype=np.uint8)
p = gpu._Params(6, 65536, gpu.F_FINAL | gpu.F_ZLIB_WRAP, gpu.LZ_AUTO, 0, 0)
dres = gpu.DeflateResult()
for _ in range(3):
    t0 = time.perf_counter()
    rc = e.L.zgpu_deflate_host(e.h, host.ctypes.data, host.size, C.byref(p), zbuf.ctypes.data, cap, None, C.byref(dres))
    t1 = time.perf_counter()
    print("zgpu_deflate_host: %.1f ms" % ((t1 - t0) * 1e3), flush=True)
