"""GPU box: compress2() of the host library (libzamd_z.so, one continuous stream, host buffers in and out) against input size.  usage: host_cont_rate.py [level]"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import zhost as Z
from oracle import corpus_py as CP
level = int(sys.argv[1]) if len(sys.argv) > 1 else 6
L = Z.lib()
big = CP.chunks(0, 0, 4096)  # 256 MiB, numpy
for mib in (1, 16, 64, 256):
    n = mib << 20
    src = big[:n]
    cap = L.compressBound(n)
    dst = (C.c_ubyte * cap)()
    best = None
    for _ in range(3):
        dl = C.c_ulong(cap)
        t0 = time.perf_counter()
        rc = L.compress2(dst, C.byref(dl), src.ctypes.data_as(C.c_void_p), n, level)
        d = time.perf_counter() - t0
        assert rc == 0, rc
        best = d if best is None or d < best else best
    print("compress2 level %d, %4d MiB: %7.2f ms = %5.2f GiB/s, %d bytes [feed %s]" % (level, mib, best * 1e3, n / best / 2**30, dl.value, os.environ.get("ZAMD_FEED_BYTES")), flush=True)
