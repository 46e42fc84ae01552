import sys, os, json, hashlib, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import zhost as Z
from oracle import cases, refzlib as R
import test_gpu_prime as TP
L = Z.lib()
L.deflatePrime.argtypes = [C.POINTER(Z.ZStream), C.c_int, C.c_int]
for c in TP.KAT:
    if c["n"] != 140000 or c["level"] != 0: continue
    d = cases.make(c["kind"], c["n"], c["seed"])
    z = TP._primed(L, d, c["level"], c["wbits"], c["prime"], c["mid"], False)
    ok = len(z) == c["len"] and hashlib.sha256(z).hexdigest()[:16] == c["sha"]
    # the reference, driven the same way
    RL = R.lib(); RL.deflatePrime.argtypes = [C.POINTER(R.ZStream), C.c_int, C.c_int]
    s = R.ZStream(); RL.deflateInit2_(C.byref(s), 0, 8, c["wbits"], 8, 0, b"1.2.3", C.sizeof(R.ZStream)); RL.deflatePrime(C.byref(s), *c["prime"])
    out = C.create_string_buffer(len(d) + 4096); inb = C.create_string_buffer(d, len(d)); s.next_out = C.addressof(out); s.avail_out = len(d) + 4096
    n = (len(d) + 65535) // 65536
    for k in range(n):
        s.next_in = C.addressof(inb) + k * 65536; s.avail_in = min(65536, len(d) - k * 65536)
        RL.deflate(C.byref(s), 4 if k + 1 == n else 3)
        if k == 0 and c["mid"] is not None: RL.deflatePrime(C.byref(s), *c["mid"])
    want = out.raw[: s.total_out]
    k = next((i for i in range(min(len(z), len(want))) if z[i] != want[i]), -1)
    print(c["wbits"], c["prime"], c["mid"], "ok" if ok else "BAD", len(z), len(want), "first diff", k, z[max(0,k-4):k+12].hex() if k >= 0 else "", want[max(0,k-4):k+12].hex() if k >= 0 else "")
