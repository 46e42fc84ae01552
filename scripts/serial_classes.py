"""Which chunks hold the lane-per-chunk loop back (level = argv[1], default 1): the Silesia-mix without one class of segments at a time (4 GiB generated, rows of 1 MiB)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, zlib_amd
from zlib_amd import gpu
e = zlib_amd.Engine(0)
n = 65536
src = torch.empty(n * 65536, dtype=torch.uint8, device="cuda")
e.corpus_fill_device(0, 0x5EED5117, 0, n, src.data_ptr())
rows = src.view(-1, 1 << 20)
slots = [0, 1, 0, 2, 0, 3, 0, 1, 2, 0, 4, 0, 3, 0, 1, 5, 0, 2, 0, 6]
cls = torch.tensor([slots[i % 20] for i in range(rows.shape[0])], device="cuda")
names = ["text", "markup", "logs", "code", "numeric", "lowbin", "random"]
cap = e.L.zgpu_deflate_bound(n * 65536, 65536)
dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
LVL = int(sys.argv[1]) if len(sys.argv) > 1 else 1
def run(t, impl, lvl=None):
    lvl = LVL if lvl is None else lvl
    nb = t.numel()
    best = 1e9
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = e.deflate_device(t.data_ptr(), nb, lvl, dst.data_ptr(), cap, flags=gpu.F_FINAL, lz_impl=impl)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3, r.ntokens / (nb / 65536)
ms, tk = run(src, gpu.LZ_SERIAL)
print("all classes: %.1f ms, %.0f tokens a chunk" % (ms, tk))
for k in range(7):
    sub = rows[cls != k].contiguous().view(-1)
    ms, tk = run(sub, gpu.LZ_SERIAL)
    print("without %-8s (%5d chunks): %.1f ms, %.0f tokens a chunk" % (names[k], sub.numel() // 65536, ms, tk))
    del sub
for k in range(7):
    only = rows[cls == k].contiguous().view(-1)
    reps = max(1, (n * 65536 // 2) // only.numel())
    only = only.repeat(reps)
    ms, tk = run(only, gpu.LZ_SERIAL)
    ms2, _ = run(only, gpu.LZ_FASTWIN)
    print("only %-8s x%d (%5d chunks): loop %.1f ms, wave kernel %.1f ms, %.0f tokens a chunk" % (names[k], reps, only.numel() // 65536, ms, ms2, tk))
    del only
