"""GPU box: time of the inflate kernel path on the headline's level-6 stream (no check of the bytes: for timing experiments with builds whose output is void)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zlib_amd
from zlib_amd import gpu
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
good = os.environ.get("ZAMD_GPU_LIB_GOOD")
e = zlib_amd.Engine(0)
src = torch.empty(n * 65536, dtype=torch.uint8, device="cuda")
e.corpus_fill_device(0, 0x5EED5117, 0, n, src.data_ptr())
cap = e.L.zgpu_deflate_bound(src.numel(), 65536)
dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
offs = torch.empty(n + 1, dtype=torch.int64, device="cuda")
back = torch.empty(src.numel(), dtype=torch.uint8, device="cuda")
r = e.deflate_device(src.data_ptr(), src.numel(), 6, dst.data_ptr(), cap, flags=gpu.F_FINAL, d_offsets=offs.data_ptr())
best = None
for _ in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    try:
        e.inflate_device(dst.data_ptr(), r.out_bytes, offs.data_ptr(), n, back.data_ptr(), src.numel())
    except Exception as ex:
        pass
    torch.cuda.synchronize(); d = time.perf_counter() - t0
    best = d if best is None or d < best else best
print("%d chunks: %.2f ms, %.2f GiB/s of output; bytes equal: %s" % (n, best * 1e3, src.numel() / best / 2**30, bool(torch.equal(src, back))))
