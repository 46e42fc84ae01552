"""GPU box: rate and stage times of the chunked stream (mode B) on device-resident input.  usage: chunks_rate.py [gib] [levels]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zlib_amd
from zlib_amd import gpu
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
levels = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "6").split(",")]
eng = zlib_amd.Engine(0)
nchunks = int(gib * 2**30) // 65536
n = nchunks * 65536
src = torch.empty(n, dtype=torch.uint8, device="cuda")
eng.corpus_fill_device(0, 0x5EED5117, 0, nchunks, src.data_ptr())
cap = eng.L.zgpu_deflate_bound(n, 65536)
dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for level in levels:
    res = eng.deflate_device(src.data_ptr(), n, level, dst.data_ptr(), cap, flags=gpu.F_FINAL | gpu.F_ZLIB_WRAP, stream=st)
    eng.profile(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(2):
        res = eng.deflate_device(src.data_ptr(), n, level, dst.data_ptr(), cap, flags=gpu.F_FINAL | gpu.F_ZLIB_WRAP, stream=st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
    pr = eng.profile_read(); eng.profile(False)
    print("level %d chunks %.2f GiB: %7.1f ms = %6.2f GiB/s, out %d, stages/ms %s [lib %s fuse %s]" % (level, n / 2**30, dt * 1e3, n / dt / 2**30, res.out_bytes,
          {k: round(v[0] / 2, 1) for k, v in pr.items() if v[1]}, os.path.basename(os.environ.get("ZAMD_GPU_LIB", "default")), os.environ.get("ZGPU_WALK_FUSE")), flush=True)
