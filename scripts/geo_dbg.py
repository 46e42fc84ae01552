import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zlib_amd
from zlib_amd import gpu
from oracle import cases
e = zlib_amd.Engine(0)
for (w, m, level) in ((9, 8, 6), (9, 8, 9), (9, 8, 4), (9, 8, 1), (9, 7, 6), (10, 8, 6)):
    e.set_geometry(w, m)
    d = cases.make("rand", 65536, 13)
    z = e.deflate_segments_host([d], level, flags=gpu.F_FINAL)[0]
    open("gpurun_out/geo_%d_%d_%d.z" % (w, m, level), "wb").write(z)
    print(w, m, level, len(z))
