import sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import cases, oracle_py as O
import zlib_amd
e = zlib_amd.Engine(0)
for n in (1, 100, 5000, 65536):
    data = cases.make("text", n, 1)
    seg = O.deflate_chunk(data, 6, True)
    offs = np.array([0, len(seg)], dtype=np.uint64)
    print("n", n, flush=True)
    out = e.inflate_host(seg, offs, out_len=max(len(data), 1))
    print("  ok", out == data, flush=True)
