"""GPU box: ONE continuous stream of 3.5 GiB (the most a 32-bit avail_in takes) against the compiled reference's compress2(), levels given (default 6, 1).  The reference runs
on host threads side by side (one stream is one core's work: minutes)."""
import sys, os, time, hashlib, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zlib_amd
from zlib_amd import gpu
from oracle import refzlib as R
levels = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "6,1").split(",")]
nchunks = 57344  # 3.5 GiB
n = nchunks * 65536
eng = zlib_amd.Engine(0)
src = torch.empty(n, dtype=torch.uint8, device="cuda")
eng.corpus_fill_device(0, 0x5EED5117, 0, nchunks, src.data_ptr())
host = src.cpu().numpy()
data = host.tobytes()
del host
want = {}
def ref(level):
    t0 = time.time(); z = R.compress2(data, level); want[level] = (len(z), hashlib.sha256(z).hexdigest(), time.time() - t0)
th = [threading.Thread(target=ref, args=(lv,)) for lv in levels]
for t in th: t.start()
cap = eng.L.zgpu_deflate_cont_bound(n) + 64
dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
got = {}
for lv in levels:
    t0 = time.time()
    res = eng.deflate_device(src.data_ptr(), n, lv, dst.data_ptr(), cap, flags=gpu.F_FINAL | gpu.F_ZLIB_WRAP | gpu.F_CONTINUOUS)
    z = dst[: res.out_bytes].cpu().numpy().tobytes()
    got[lv] = (len(z), hashlib.sha256(z).hexdigest(), time.time() - t0)
    print("device level %d: %d bytes in %.2f s" % (lv, got[lv][0], got[lv][2]), flush=True)
for t in th: t.join()
bad = 0
for lv in levels:
    ok = got[lv][:2] == want[lv][:2]
    bad += not ok
    print("level %d, %.2f GiB: device %d bytes, reference %d bytes (%.0f s on one core): %s" % (lv, n / 2**30, got[lv][0], want[lv][0], want[lv][2], "identical" if ok else "DIFFERENT"), flush=True)
sys.exit(1 if bad else 0)
