"""GPU box: the rate of the host-buffer entry points (PCIe inclusive: H2D of the input, D2H of the stream), 1 GiB of the Silesia-mix."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import zlib_amd
from zlib_amd import gpu

e = zlib_amd.Engine(0)
n = 16384
src = torch.empty(n * 65536, dtype=torch.uint8, device="cuda")
e.corpus_fill_device(0, 0x5EED5117, 0, n, src.data_ptr())
host = src.cpu().numpy()
for _ in range(2):
    t0 = time.perf_counter(); z = e.deflate_host(host, 6); t1 = time.perf_counter()
print("zgpu_deflate_host  level 6: %.2f GiB/s (%.1f ms for 1 GiB, stream %d bytes)" % (1.0 / (t1 - t0), (t1 - t0) * 1e3, len(z)))
zz, offs = e.deflate_host(host, 6, flags=gpu.F_FINAL, want_offsets=True)
for _ in range(2):
    t0 = time.perf_counter(); out = e.inflate_host(zz, offs, out_len=host.size); t1 = time.perf_counter()
assert out == host.tobytes()
print("zgpu_inflate_host: %.2f GiB/s of output (%.1f ms)" % (1.0 / (t1 - t0), (t1 - t0) * 1e3))

# the C entry points alone (ctypes, buffers allocated and touched beforehand: no page faults, no Python copies inside the timing)
import ctypes as C
zarr = np.frombuffer(zz, dtype=np.uint8)
offs64 = np.ascontiguousarray(offs, dtype=np.uint64)
outbuf = np.zeros(host.size, dtype=np.uint8)
res = gpu.InflateResult()
for _ in range(3):
    t0 = time.perf_counter()
    rc = e.L.zgpu_inflate_host(e.h, zarr.ctypes.data, zarr.size, offs64.ctypes.data, len(offs64) - 1, 65536, outbuf.ctypes.data, outbuf.size, C.byref(res))
    t1 = time.perf_counter()
assert rc == 0 and outbuf.tobytes() == host.tobytes()
print("zgpu_inflate_host (C call only): %.2f GiB/s of output (%.1f ms)" % (1.0 / (t1 - t0), (t1 - t0) * 1e3))
cap = e.L.zgpu_deflate_bound(host.size, 65536)
zbuf = np.zeros(cap, dtype=np.uint8)
p = gpu._Params(6, 65536, gpu.F_FINAL | gpu.F_ZLIB_WRAP, gpu.LZ_AUTO, 0, 0)
dres = gpu.DeflateResult()
for _ in range(3):
    t0 = time.perf_counter()
    rc = e.L.zgpu_deflate_host(e.h, host.ctypes.data, host.size, C.byref(p), zbuf.ctypes.data, cap, None, C.byref(dres))
    t1 = time.perf_counter()
assert rc == 0
print("zgpu_deflate_host (C call only): %.2f GiB/s (%.1f ms)" % (1.0 / (t1 - t0), (t1 - t0) * 1e3))
# raw copies for comparison
dbuf = torch.empty(host.size, dtype=torch.uint8, device="cuda")
hp = torch.from_numpy(outbuf)
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); hp.copy_(dbuf); torch.cuda.synchronize(); t1 = time.perf_counter()
print("pageable D2H of 1 GiB: %.1f ms;" % ((t1 - t0) * 1e3), end=" ")
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); dbuf.copy_(hp); torch.cuda.synchronize(); t1 = time.perf_counter()
print("pageable H2D of 1 GiB: %.1f ms;" % ((t1 - t0) * 1e3), end=" ")
pin = torch.empty(host.size, dtype=torch.uint8).pin_memory()
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); pin.copy_(dbuf); torch.cuda.synchronize(); t1 = time.perf_counter()
print("pinned D2H: %.1f ms" % ((t1 - t0) * 1e3))
