"""GPU box: rate of the decode of a stream that was not produced in chunks (system zlib, level 6), through zgpu_inflate_stream_host2 (host
buffers in and out, PCIe included) -- decoded in pieces (spec_* in zgpu_inflate.hip) -- against the system zlib on one host core.
  python scripts/foreign_stream_rate.py [MiB of input, default 1024]"""
import os
import sys
import time
import zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zlib_amd

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
e = zlib_amd.Engine(0)
n = mib * 16
src = torch.empty(n * 65536, dtype=torch.uint8, device="cuda")
e.corpus_fill_device(0, 0x5EED5117, 0, n, src.data_ptr())
data = src.cpu().numpy().tobytes()
t0 = time.perf_counter()
co = zlib.compressobj(6, zlib.DEFLATED, -15)
raw = b"".join(co.compress(data[i:i + (64 << 20)]) for i in range(0, len(data), 64 << 20)) + co.flush()
print("system zlib compressed %d MiB to %d MiB in %.1f s" % (mib, len(raw) >> 20, time.perf_counter() - t0), flush=True)
import numpy as np
e.profile(True)
best = 1e9
dst = np.zeros(len(data), dtype=np.uint8)  # the caller's buffer, touched (a fresh mapping costs a page fault per 4 KiB inside the copy)
for _ in range(3):
    t0 = time.perf_counter(); out = e.inflate_stream_host(raw, len(data), out=dst); t1 = time.perf_counter()
    best = min(best, t1 - t0)
assert out.tobytes() == data
print("in pieces / one workgroup so far:", e.spec_counts())
prof = e.profile_read()
print("device spans:", {k: v for k, v in prof.items() if v[1]})
t2 = time.perf_counter(); ref = zlib.decompress(raw, -15); t3 = time.perf_counter()
print("foreign stream of %d MiB (ratio %.2f): %.1f ms = %.2f GiB/s of output end to end (host buffers); system zlib on one host core: %.0f MiB/s"
      % (mib, len(data) / len(raw), best * 1e3, mib / 1024 / best, mib / (t3 - t2)))
