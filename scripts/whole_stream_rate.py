"""GPU box: rate of the end-to-end decode of a stream that was not produced in chunks (ZGPU_WHOLE_STREAM, one workgroup)."""
import os
import sys
import time
import zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import zlib_amd
from zlib_amd import gpu

e = zlib_amd.Engine(0)
n = 512                                                     # 32 MiB of the Silesia-mix
src = torch.empty(n * 65536, dtype=torch.uint8, device="cuda")
e.corpus_fill_device(0, 0x5EED5117, 0, n, src.data_ptr())
data = src.cpu().numpy().tobytes()
co = zlib.compressobj(6, zlib.DEFLATED, -15)
raw = co.compress(data) + co.flush()
offs = np.array([0, len(raw)], dtype=np.uint64)
for _ in range(2):
    t0 = time.perf_counter(); out = e.inflate_host(raw, offs, chunk_size=gpu.WHOLE_STREAM, out_len=len(data)); t1 = time.perf_counter()
assert out == data
t2 = time.perf_counter(); ref = zlib.decompress(raw, -15); t3 = time.perf_counter()
print("whole-stream inflate of %d MiB (system zlib level 6 stream, ratio %.2f): %.1f ms = %.1f MiB/s of output; system zlib on one host core: %.1f MiB/s"
      % (len(data) >> 20, len(data) / len(raw), (t1 - t0) * 1e3, (len(data) >> 20) / (t1 - t0), (len(data) >> 20) / (t3 - t2)))
