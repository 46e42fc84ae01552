"""GPU box, experiment: does the chip do more when two engines (two streams, two workspaces) compress side by side?  The kernels of one
call run one after the other; the latency-bound ones (sort, block construction) leave the vector units idle, the search kernel fills the
register file.  Two calls in flight let the hardware overlap one's latency-bound kernels with the other's search -- if resources allow."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, zlib_amd
from zlib_amd import gpu

total = 65536
src = torch.empty(total * 65536, dtype=torch.uint8, device="cuda")
e0 = zlib_amd.Engine(0)
e0.corpus_fill_device(0, 0x5EED5117, 0, total, src.data_ptr())
torch.cuda.synchronize()


def run(engines, parts, reps=3):
    n = total // parts
    caps = [e0.L.zgpu_deflate_bound(n * 65536, 65536)] * parts
    dsts = [torch.empty(caps[i], dtype=torch.uint8, device="cuda") for i in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    def work(i):
        e = engines[i % len(engines)]
        e.deflate_device(src.data_ptr() + i * n * 65536, n * 65536, 6, dsts[i].data_ptr(), caps[i], flags=gpu.F_FINAL, stream=streams[i].cuda_stream)
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if len(engines) == 1:
            for i in range(parts): work(i)
        else:
            th = [threading.Thread(target=work, args=(i,)) for i in range(parts)]
            for t in th: t.start()
            for t in th: t.join()
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best


t1 = run([e0], 1)
print("one engine, 4 GiB in one call: %.1f ms = %.2f GiB/s" % (t1 * 1e3, 4 / t1))
t2 = run([e0], 2)
print("one engine, two calls of 2 GiB one after the other: %.1f ms = %.2f GiB/s" % (t2 * 1e3, 4 / t2))
e1 = zlib_amd.Engine(0)
t3 = run([e0, e1], 2)
print("two engines, 2 GiB each, side by side: %.1f ms = %.2f GiB/s" % (t3 * 1e3, 4 / t3))
e2 = zlib_amd.Engine(0); e3 = zlib_amd.Engine(0)
t4 = run([e0, e1, e2, e3], 4)
print("four engines, 1 GiB each, side by side: %.1f ms = %.2f GiB/s" % (t4 * 1e3, 4 / t4))
