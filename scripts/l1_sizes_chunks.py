"""GPU box: level 1-3 in chunks against input size (device buffers).  usage: l1_sizes_chunks.py [level]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zlib_amd
from zlib_amd import gpu
level = int(sys.argv[1]) if len(sys.argv) > 1 else 1
e = zlib_amd.Engine(0)
nmax = 65536
src = torch.empty(nmax * 65536, dtype=torch.uint8, device="cuda")
e.corpus_fill_device(0, 0x5EED5117, 0, nmax, src.data_ptr())
cap = e.L.zgpu_deflate_bound(src.numel(), 65536)
dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
row = []
for mib in [int(x) for x in os.environ.get("SIZES", "1,4,16,64,256,1024,4096").split(",")]:
    n = mib << 20
    best = None
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        e.deflate_device(src.data_ptr(), n, level, dst.data_ptr(), cap, flags=gpu.F_FINAL, lz_impl={"auto": gpu.LZ_AUTO, "fastwin": gpu.LZ_FASTWIN, "serial": gpu.LZ_SERIAL}[os.environ.get("LZ", "auto")])
        torch.cuda.synchronize(); d = time.perf_counter() - t0
        best = d if best is None or d < best else best
    row.append("%d MiB %.1f ms" % (mib, best * 1e3))
print("level %d chunks, ZGPU_FW_RING=%s LZ=%s: %s" % (level, os.environ.get("ZGPU_FW_RING"), os.environ.get("LZ", "auto"), ", ".join(row)), flush=True)
