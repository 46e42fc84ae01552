"""GPU box: rate against input size (device buffers), levels 1 / 4 / 6 and inflate of the level-6 stream -- the kernels are tuned for
launches that fill the chip several times over; this shows what smaller calls get."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zlib_amd
from zlib_amd import gpu

e = zlib_amd.Engine(0)
nmax = 16384
src = torch.empty(nmax * 65536, dtype=torch.uint8, device="cuda")
e.corpus_fill_device(0, 0x5EED5117, 0, nmax, src.data_ptr())
cap = max(e.L.zgpu_deflate_bound(src.numel(), 65536), e.L.zgpu_deflate_cont_bound(src.numel()) + 64)
dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
back = torch.empty(src.numel(), dtype=torch.uint8, device="cuda")
offs = torch.empty(nmax + 1, dtype=torch.int64, device="cuda")


def best_of(f, k=5):
    b = None
    for _ in range(k):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); d = time.perf_counter() - t0
        b = d if b is None or d < b else b
    return b


print("%8s %22s %22s %22s %22s | one continuous stream: %19s %22s %22s" % ("MiB", "level 1", "level 4", "level 6", "inflate (of level 6)", "level 1", "level 4", "level 6"))
for mib in (1, 4, 16, 64, 256, 1024):
    n = mib << 20
    row = []
    for lv in (1, 4, 6):
        d = best_of(lambda: e.deflate_device(src.data_ptr(), n, lv, dst.data_ptr(), cap, flags=gpu.F_FINAL, d_offsets=offs.data_ptr()))
        row.append("%7.2f ms %6.2f GiB/s" % (d * 1e3, n / d / 2**30))
    r = e.deflate_device(src.data_ptr(), n, 6, dst.data_ptr(), cap, flags=gpu.F_FINAL, d_offsets=offs.data_ptr())
    d = best_of(lambda: e.inflate_device(dst.data_ptr(), r.out_bytes, offs.data_ptr(), n >> 16, back.data_ptr(), n))
    assert torch.equal(back[:n], src[:n])
    row.append("%7.2f ms %6.2f GiB/s" % (d * 1e3, n / d / 2**30))
    for lv in (1, 4, 6):
        d = best_of(lambda: e.deflate_device(src.data_ptr(), n, lv, dst.data_ptr(), cap, flags=gpu.F_FINAL | gpu.F_CONTINUOUS), 3)
        row.append("%7.2f ms %6.2f GiB/s" % (d * 1e3, n / d / 2**30))
    print("%8d %22s %22s %22s %22s | %22s %22s %22s" % (mib, *row), flush=True)
