import time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, zlib_amd
from zlib_amd import gpu
e = zlib_amd.Engine(0)
n = 16384 * 4
src = torch.empty(n * 65536, dtype=torch.uint8, device="cuda")
e.corpus_fill_device(0, 0x5EED5117, 0, n, src.data_ptr())
h = src.cpu().numpy()
for lvl in (1, 2, 3, 6):
    for impl in (gpu.LZ_AUTO,):
        for rep in range(2):
            e.profile(True)
            t = time.perf_counter(); z = e.deflate_host(h, lvl, lz_impl=impl); dt = time.perf_counter() - t
            prof = e.profile_read(); e.profile(False)
        print("host buffers 4 GiB level %d impl %d: %.1f ms  %.2f GiB/s  stages %s" % (lvl, impl, dt * 1e3, n * 65536 / dt / 2**30, {k: (round(v[0], 1), v[1]) for k, v in prof.items() if v[1]}))
