import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, zlib_amd
from zlib_amd import gpu
eng = zlib_amd.Engine(0)
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
level = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n = int(gib * 2**30) // 65536 * 65536
src = torch.empty(n, dtype=torch.uint8, device="cuda")
kind = int(os.environ.get("KIND", "0"))
eng.corpus_fill_device(kind, 0x5EED5117 if kind == 0 else 0x10C7E47, 0, n // 65536, src.data_ptr())
cap = eng.L.zgpu_deflate_cont_bound(n) + 64
dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
os.environ["ZGPU_FAST_TRACE"] = "1"
t = time.time()
eng.deflate_device(src.data_ptr(), n, level, dst.data_ptr(), cap, flags=gpu.F_FINAL | gpu.F_ZLIB_WRAP | gpu.F_CONTINUOUS)
print("total %.1f ms" % ((time.time() - t) * 1e3))
