"""GPU box: degenerate 1 GiB inputs through the device entry points (deflate level 6 and 9, inflate back, compare)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zlib_amd
from zlib_amd import gpu

e = zlib_amd.Engine(0)
n = 16384 * 65536
cases = {
    "zeros": torch.zeros(n, dtype=torch.uint8, device="cuda"),
    "random": torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda"),
    "period-7": (torch.arange(n, device="cuda") % 7).to(torch.uint8),
    "two-symbols": torch.randint(0, 2, (n,), dtype=torch.uint8, device="cuda") * 65,
}
cap = e.L.zgpu_deflate_bound(n, 65536)
dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
offs = torch.empty(16384 + 1, dtype=torch.int64, device="cuda")
back = torch.empty(n, dtype=torch.uint8, device="cuda")
for name, src in cases.items():
    for lvl in (6, 9, 1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = e.deflate_device(src.data_ptr(), n, lvl, dst.data_ptr(), cap, flags=gpu.F_FINAL, d_offsets=offs.data_ptr())
        torch.cuda.synchronize(); t1 = time.perf_counter()
        ir = e.inflate_device(dst.data_ptr(), r.out_bytes, offs.data_ptr(), 16384, back.data_ptr(), n)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        ok = bool(torch.equal(src, back)) and ir.out_bytes == n
        print("%-12s L%d: deflate %7.1f ms (%.2f GiB/s, ratio %.1f)  inflate %7.1f ms  round trip %s" % (
            name, lvl, (t1 - t0) * 1e3, 1.0 / (t1 - t0), n / r.out_bytes, (t2 - t1) * 1e3, "ok" if ok else "MISMATCH"), flush=True)
        assert ok
