"""GPU box: the continuous stream of a large input at the fast levels against the compiled reference (oracle/_ref/libzref.so) on both corpora.  usage: cont_big_check.py [mib] [levels]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zlib_amd
from zlib_amd import gpu
from oracle import refzlib as R, corpus_py as CP
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 96
levels = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,3").split(",")]
eng = zlib_amd.Engine(0)
bad = 0
for kind in (0, 1):
    d = CP.chunks(kind, 777, mib * 16).tobytes()[:-4321]
    for level in levels:
        t0 = time.time(); want = R.compress2(d, level); t1 = time.time()
        got = eng.deflate_host(d, level, flags=gpu.F_FINAL | gpu.F_ZLIB_WRAP | gpu.F_CONTINUOUS); t2 = time.time()
        ok = got == want
        bad += not ok
        print("corpus %d, %d MiB, level %d: %d bytes, reference %d (%.1f s), device %.2f s: %s" % (kind, mib, level, len(got), len(want), t1 - t0, t2 - t1, "identical" if ok else "DIFFERENT"), flush=True)
sys.exit(1 if bad else 0)
