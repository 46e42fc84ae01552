"""GPU box: zgpu_deflate_host (host buffers in and out) at level 6, `GIB` GiB of the Silesia-mix; run once per setting of
ZGPU_FIRST_BATCH / ZGPU_HOST_BATCH, given as first:batch in chunks (the engine reads them per call).  Prints the best of three calls and the resident rate beside it."""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import zlib_amd
from zlib_amd import gpu

gib = int(os.environ.get("GIB", "1"))
e = zlib_amd.Engine(0)
n = 16384 * gib
src = torch.empty(n * 65536, dtype=torch.uint8, device="cuda")
e.corpus_fill_device(0, 0x5EED5117, 0, n, src.data_ptr())
host = src.cpu().numpy()
cap = e.L.zgpu_deflate_bound(host.size, 65536)
zbuf = np.zeros(cap, dtype=np.uint8)
p = gpu._Params(6, 65536, gpu.F_FINAL | gpu.F_ZLIB_WRAP, gpu.LZ_AUTO, 0, 0)
dres = gpu.DeflateResult()
dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e.deflate_device(src.data_ptr(), host.size, 6, dst.data_ptr(), cap)
    torch.cuda.synchronize(); res_ms = (time.perf_counter() - t0) * 1e3
ref = None
for setting in sys.argv[1:] or ["2048:16384"]:
    first, batch = setting.split(":")
    os.environ["ZGPU_FIRST_BATCH"] = first
    os.environ["ZGPU_HOST_BATCH"] = batch
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        rc = e.L.zgpu_deflate_host(e.h, host.ctypes.data, host.size, C.byref(p), zbuf.ctypes.data, cap, None, C.byref(dres))
        d = time.perf_counter() - t0
        assert rc == 0, rc
        best = d if best is None or d < best else best
    z = zbuf[:dres.out_bytes].tobytes()
    if ref is None:
        ref = z
    assert z == ref, "stream differs between settings"
    print("first %s batch %s: %.1f ms = %.2f GiB/s (resident: %.1f ms = %.2f GiB/s; ratio %.2f)" % (
        first, batch, best * 1e3, gib / best, res_ms, gib / (res_ms / 1e3), res_ms / 1e3 / best), flush=True)
