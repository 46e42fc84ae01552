import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import cases, oracle_py as O
import zlib_amd
e = zlib_amd.Engine(0)
data = cases.make("text", 300, 1)
seg = O.deflate_chunk(data, 6, True)
try:
    print(e.inflate_host(seg, np.array([0, len(seg)], dtype=np.uint64), out_len=300) == data)
except Exception as ex:
    print("ERR", ex)
