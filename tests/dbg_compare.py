"""Debug helper (GPU box): compare engine output with the oracle and report the first difference."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import cases, corpus_py as CP, oracle_py as O
import zlib_amd
from zlib_amd import gpu

e = zlib_amd.Engine(0)


def cmp(data, lvl, impl=gpu.LZ_AUTO, cs=65536, label=""):
    z, offs = e.deflate_host(data, lvl, lz_impl=impl, chunk_size=cs, want_offsets=True)
    w = O.deflate_stream(data, lvl, cs)
    if z == w:
        print("OK  ", label, lvl, len(z))
        return True
    n = min(len(z), len(w))
    first = next((i for i in range(n) if z[i] != w[i]), n)
    ck = int(np.searchsorted(offs, first, side="right")) - 1
    print("DIFF", label, "lvl", lvl, "len", len(z), len(w), "first diff at", first, "chunk", ck, "chunk off", int(offs[ck]) if ck >= 0 else None,
          "adler", hex(e.last.adler32), hex(O.adler32(data)), "ntok", e.last.ntokens)
    print("   got ", z[first - 4:first + 12].hex(), " want", w[first - 4:first + 12].hex())
    return False


if __name__ == "__main__":
    IMPL = int(os.environ.get("DBG_IMPL", "0"))
    big = cases.hello_1mib()
    for lvl in (6, 9):
        cmp(big, lvl, impl=IMPL, label="hello1m")
    for lvl in (4, 6):
        cmp(big[:65536 * 2], lvl, impl=IMPL, label="hello128k")
        cmp(big[:65536 + 10], lvl, impl=IMPL, label="hello64k+10")
        cmp(CP.chunks(0, 0, 4).tobytes(), lvl, impl=IMPL if lvl >= 4 else 0, label="corpus4")
