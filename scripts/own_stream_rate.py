"""GPU box: uncompress() of this library's own stream without a side table (host buffers; the chunk boundaries are found by the marker scan) against
zgpu_inflate_host with the table."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, zlib_amd
from zlib_amd import gpu
e = zlib_amd.Engine(0)
n = 16384
src = torch.empty(n * 65536, dtype=torch.uint8, device="cuda")
e.corpus_fill_device(0, 0x5EED5117, 0, n, src.data_ptr())
host = src.cpu().numpy()
z, offs = e.deflate_host(host, 6, flags=gpu.F_FINAL, want_offsets=True)
dst = np.zeros(host.size, dtype=np.uint8)
for _ in range(3):
    t0 = time.perf_counter(); out = e.inflate_stream_host(z, host.size, out=dst); t1 = time.perf_counter()
assert out.tobytes() == host.tobytes()
print("no table (marker scan, compact decode, stitch): %.1f ms = %.2f GiB/s of output" % ((t1 - t0) * 1e3, 1 / (t1 - t0)))
zarr = np.frombuffer(z, dtype=np.uint8); offs64 = np.ascontiguousarray(offs, dtype=np.uint64); res = gpu.InflateResult()
for _ in range(3):
    t0 = time.perf_counter(); rc = e.L.zgpu_inflate_host(e.h, zarr.ctypes.data, zarr.size, offs64.ctypes.data, len(offs64) - 1, 65536, dst.ctypes.data, dst.size, C.byref(res)); t1 = time.perf_counter()
print("with the table: %.1f ms = %.2f GiB/s" % ((t1 - t0) * 1e3, 1 / (t1 - t0)))
