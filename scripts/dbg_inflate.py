import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import cases, oracle_py as O
import zlib_amd
e = zlib_amd.Engine(0)
def run(name, data, level):
    seg, info, toks = O.deflate_chunk(data, level, True, want_tokens=True)
    offs = np.array([0, len(seg)], dtype=np.uint64)
    try:
        out = e.inflate_host(seg, offs, out_len=max(len(data), 1))
        ok = out == data
        first = next((i for i in range(min(len(out), len(data))) if out[i] != data[i]), None)
        print(name, level, "btype", list(info.btype)[:info.nblocks], "OK" if ok else ("MISMATCH len %d/%d first %s" % (len(out), len(data), first)))
    except Exception as ex:
        print(name, level, "btype", list(info.btype)[:info.nblocks], "ERR", ex)
for lvl in (1, 6):
    run("a*100", b"a" * 100, lvl)
    run("abc", b"abcabcabcabcxyzxyz", lvl)
    run("hello", cases.HELLO, lvl)
    run("text300", cases.make("text", 300, 1), lvl)
    run("text5000", cases.make("text", 5000, 1), lvl)
    run("text65536", cases.make("text", 65536, 1), lvl)
    run("rand3000", cases.make("rand", 3000, 1), lvl)
    run("runs20000", cases.make("runs", 20000, 1), lvl)
    run("ab20000", cases.make("ab", 20000, 1), lvl)
