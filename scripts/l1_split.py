"""GPU box, experiment: level 1 in chunks with the two kernels side by side -- the lane-per-chunk loop on one part of the input, the wave-per-chunk kernel on the rest, two engines on two streams."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, zlib_amd
from zlib_amd import gpu

level = int(sys.argv[1]) if len(sys.argv) > 1 else 1
total = 65536
src = torch.empty(total * 65536, dtype=torch.uint8, device="cuda")
e0 = zlib_amd.Engine(0); e1 = zlib_amd.Engine(0)
e0.corpus_fill_device(0, 0x5EED5117, 0, total, src.data_ptr())
torch.cuda.synchronize()
cap = e0.L.zgpu_deflate_bound(total * 65536, 65536)
d0 = torch.empty(cap, dtype=torch.uint8, device="cuda"); d1 = torch.empty(cap, dtype=torch.uint8, device="cuda")
s0 = torch.cuda.Stream(); s1 = torch.cuda.Stream()


def timed(f, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3


print("level %d: all chunks, auto: %.1f ms" % (level, timed(lambda: e0.deflate_device(src.data_ptr(), total * 65536, level, d0.data_ptr(), cap, flags=gpu.F_FINAL, stream=s0.cuda_stream))))
for pct in (20, 30, 40, 50):
    na = total * pct // 100; nb = total - na
    def both():
        ta = threading.Thread(target=lambda: e1.deflate_device(src.data_ptr(), na * 65536, level, d1.data_ptr(), cap, flags=0, lz_impl=gpu.LZ_FASTWIN, stream=s1.cuda_stream))
        tb = threading.Thread(target=lambda: e0.deflate_device(src.data_ptr() + na * 65536, nb * 65536, level, d0.data_ptr(), cap, flags=gpu.F_FINAL, stream=s0.cuda_stream))
        ta.start(); tb.start(); ta.join(); tb.join()
    ta_alone = timed(lambda: e1.deflate_device(src.data_ptr(), na * 65536, level, d1.data_ptr(), cap, flags=0, lz_impl=gpu.LZ_FASTWIN, stream=s1.cuda_stream))
    tb_alone = timed(lambda: e0.deflate_device(src.data_ptr() + na * 65536, nb * 65536, level, d0.data_ptr(), cap, flags=gpu.F_FINAL, stream=s0.cuda_stream))
    print("  %2d %% to the wave kernel: alone %.1f ms, the loop on the rest alone %.1f ms, side by side %.1f ms" % (pct, ta_alone, tb_alone, timed(both)), flush=True)
