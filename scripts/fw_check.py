"""Debug helper (GPU box): levels 1-3 through fastwin_kernel against the lane-per-chunk loop, chunk by chunk, several runs (timing-dependent faults show as
chunks that differ between runs)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import zlib_amd
from zlib_amd import gpu
from oracle import corpus_py as CP

lvl = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 3
first = int(sys.argv[4]) if len(sys.argv) > 4 else 0
e = zlib_amd.Engine(0)
data = CP.chunks(CP.KIND_SILESIA, first, n)
zs, os_ = e.deflate_host(data, lvl, flags=0, lz_impl=gpu.LZ_SERIAL, want_offsets=True)
for r in range(runs):
    zf, of = e.deflate_host(data, lvl, flags=0, lz_impl=gpu.LZ_FASTWIN, want_offsets=True)
    bad = [i for i in range(n) if zs[int(os_[i]):int(os_[i + 1])] != zf[int(of[i]):int(of[i + 1])]]
    print("level %d run %d: %d of %d chunks differ %s" % (lvl, r, len(bad), n, bad[:12]), flush=True)
