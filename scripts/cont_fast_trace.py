"""GPU box: one continuous stream at a fast level with the rounds' trace (ZGPU_FAST_TRACE) and the stage clock.  usage: cont_fast_trace.py [gib] [level]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ZGPU_FAST_TRACE"] = "1"
import torch
import zlib_amd
from zlib_amd import gpu
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
level = int(sys.argv[2]) if len(sys.argv) > 2 else 3
e = zlib_amd.Engine(0)
nch = int(gib * 16384)
src = torch.empty(nch * 65536, dtype=torch.uint8, device="cuda")
e.corpus_fill_device(0, 0x5EED5117, 0, nch, src.data_ptr())
cap = e.L.zgpu_deflate_cont_bound(src.numel()) + 64
dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
F = gpu.F_FINAL | gpu.F_CONTINUOUS
e.deflate_device(src.data_ptr(), src.numel(), level, dst.data_ptr(), cap, flags=F)
print("---- second call", file=sys.stderr, flush=True)
e.profile(True)
torch.cuda.synchronize(); t0 = time.perf_counter()
r = e.deflate_device(src.data_ptr(), src.numel(), level, dst.data_ptr(), cap, flags=F)
torch.cuda.synchronize(); d = time.perf_counter() - t0
pr = e.profile_read()
print("level %d %.2f GiB: %.1f ms = %.2f GiB/s, out %d, stages/ms %s" % (level, gib, d * 1e3, gib / d, r.out_bytes, {k: round(v[0], 1) for k, v in pr.items() if v[1]}), flush=True)
