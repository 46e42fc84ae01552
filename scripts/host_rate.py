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
