"""Levels 4-9: the walk kernel's time per class of the Silesia-mix (each class alone, repeated to about 2 GiB)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, zlib_amd
from zlib_amd import gpu
LVL = int(sys.argv[1]) if len(sys.argv) > 1 else 9
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
def run(t):
    nb = t.numel(); best = 1e9
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = e.deflate_device(t.data_ptr(), nb, LVL, dst.data_ptr(), cap, flags=gpu.F_FINAL)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3, nb / r.out_bytes
ms, ratio = run(src)
print("level %d all classes: %.1f ms (ratio %.2f)" % (LVL, ms, ratio))
share = [9, 3, 3, 2, 1, 1, 1]
for k in range(7):
    only = rows[cls == k].contiguous().view(-1)
    reps = max(1, (n * 65536 // 2) // only.numel())
    only = only.repeat(reps)
    ms, ratio = run(only)
    per = ms / (only.numel() / 65536) * 1e3
    print("only %-8s x%-2d (%5d chunks): %7.1f ms  %6.2f us a chunk  ratio %.2f  -> %5.1f ms of a 4 GiB mix" % (names[k], reps, only.numel() // 65536, ms, per, ratio, per * 65536 * share[k] / 20 / 1e3))
    del only
