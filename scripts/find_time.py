"""Debug helper (GPU box): per-phase clock of the block finder from a -DZGPU_FIND_TIME build (ZAMD_GPU_LIB=build/variants/findtime.so)."""
import ctypes, os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, zlib_amd
e = zlib_amd.Engine(0)
n = 512 * 16
src = torch.empty(n * 65536, dtype=torch.uint8, device="cuda")
e.corpus_fill_device(0, 0x5EED5117, 0, n, src.data_ptr())
data = src.cpu().numpy().tobytes()
co = zlib.compressobj(6, zlib.DEFLATED, -15)
raw = b"".join(co.compress(data[i:i + (64 << 20)]) for i in range(0, len(data), 64 << 20)) + co.flush()
dst = np.zeros(len(data), dtype=np.uint8)
f = e.L.zgpu_debug_find_time
f.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
out = (ctypes.c_ulonglong * 8)()
e.inflate_stream_host(raw, len(data), out=dst); f(out, 1)
e.inflate_stream_host(raw, len(data), out=dst); f(out, 0)
names = ["stored sieve", "staging", "sieve 1 (type bits, counts)", "sieve 2 (code-length code complete)", "header parses"]
nf = int(out[6]); tot = sum(int(out[i]) for i in range(5))
for i, nm in enumerate(names):
    print("%-40s %8.1f us/finder %5.1f%%" % (nm, int(out[i]) / nf / 100.0, 100.0 * int(out[i]) / tot))
print("%d finders, %.1f header parses each, %.1f us each" % (nf, int(out[5]) / nf, int(out[4]) / max(int(out[5]), 1) / 100.0))
