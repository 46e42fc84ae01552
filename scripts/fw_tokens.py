"""Debug helper (GPU box): the tokens fastwin_kernel writes for one chunk (a -DZGPU_FW_DUMP build) against deflate_fast in Python: the first token that differs."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import zlib_amd
from zlib_amd import gpu
from oracle import corpus_py as CP

CFG = {1: (4, 8, 4), 2: (5, 16, 8), 3: (6, 32, 32)}
MAXD = 32506


def ref_tokens(b, level):
    maxins, nice, chain = CFG[level]
    n = len(b)
    buckets = {}
    ins = bytearray(n)
    toks = []
    p = 0
    while p < n:
        ln, ms = 2, 0
        look = n - p
        if p + 3 <= n:
            h = (((b[p] & 31) << 10) ^ (b[p + 1] << 5) ^ b[p + 2]) & 0x7fff
            lst = buckets.setdefault(h, [])
            first, best, ch = True, 2, chain
            ni = min(nice, look)
            limit = p - MAXD if p > MAXD else 0
            cap = min(look, 258)
            for q in reversed(lst):
                if not ins[q]:
                    continue
                if first:
                    if q <= 0 or p - q > MAXD:
                        break
                    first = False
                elif q <= limit:
                    break
                l = 0
                while l < cap and b[q + l] == b[p + l]:
                    l += 1
                if l > best:
                    best, ms = l, q
                    if l >= ni:
                        break
                ch -= 1
                if ch == 0:
                    break
            ln = 2 if first else min(best, look)
        if ln >= 3:
            toks.append((p, ln, p - ms))
            short = ln <= maxins and look - ln >= 3
            for k in range(ln):
                if p + k + 3 <= n:
                    h = (((b[p + k] & 31) << 10) ^ (b[p + k + 1] << 5) ^ b[p + k + 2]) & 0x7fff
                    buckets.setdefault(h, []).append(p + k)
                    if k == 0 or short:
                        ins[p + k] = 1
            p += ln
        else:
            toks.append((p, 1, 0))
            if p + 3 <= n:
                lst.append(p)
                ins[p] = 1
            p += 1
    return toks


lvl = int(sys.argv[1]) if len(sys.argv) > 1 else 1
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 1849
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 1
e = zlib_amd.Engine(0)
data = CP.chunks(CP.KIND_SILESIA, chunk, 1)
b = data.tobytes()
want = ref_tokens(b, lvl)
f = e.L.zgpu_debug_fw_tokens
f.argtypes = [ctypes.c_void_p]
f.restype = ctypes.c_uint32
out = np.zeros(65536, dtype=np.uint32)
fl = e.L.zgpu_debug_fw_lanes
fl.argtypes = [ctypes.c_void_p]
dbg = np.zeros(64 * 8 + 64, dtype=np.uint32)
for r in range(runs):
    e.deflate_host(data, lvl, flags=0, lz_impl=gpu.LZ_FASTWIN)
    nt = f(out.ctypes.data)
    got, p = [], 0
    for t in out[:nt]:
        t = int(t)
        if t >> 8:
            got.append((p, (t & 255) + 3, t >> 8)); p += (t & 255) + 3
        else:
            got.append((p, 1, 0)); p += 1
    k = 0
    while k < min(len(got), len(want)) and got[k] == want[k]:
        k += 1
    if k == len(want) == len(got):
        print("run %d: %d tokens, all equal" % (r, nt))
    else:
        fl(dbg.ctypes.data)
        print("rounds seen %d: %s" % (dbg[512], [tuple(int(x) for x in dbg[513 + 4 * i: 517 + 4 * i]) for i in range(4)]))
        for ln in range(64):
            v = [int(x) for x in dbg[ln * 8: ln * 8 + 8]]
            print("  lane %2d: idx %5d rank %5d res %#x mstart %5d seen %#010x range %#010x own %#010x (data %#010x) stg0 %5d nxt %d"
                  % (ln, v[0] & 0xffff, v[0] >> 16, v[1], v[2], v[3], v[4], v[5], int.from_bytes(b[1024 + ln: 1028 + ln], "little"), v[6], v[7]))
        print("run %d: %d / %d tokens; first difference at token %d: got %s want %s (window %d lane %d); next got %s want %s"
              % (r, len(got), len(want), k, got[k:k + 1], want[k:k + 1], want[k][0] // 64, want[k][0] % 64, got[k + 1:k + 3], want[k + 1:k + 3]))
