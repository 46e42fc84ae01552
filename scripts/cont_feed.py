"""Development check of the FEED interface of the continuous stream (zgpu_deflate_cont_host) on the GPU box: a Python stand-in for what deflate() of the
host library does with it -- slices, Z_SYNC_FLUSH / Z_FULL_FLUSH / Z_PARTIAL_FLUSH, Z_FINISH -- against the compiled reference driven by the same calls."""
import sys, os, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zlib_amd
from zlib_amd import gpu
from oracle import refzlib as R, corpus_py as CP

HIST = 32512


class BitTail:
    """the stream's unfinished byte"""
    def __init__(self): self.n = 0; self.v = 0
    def put(self, out, value, nbits):
        self.v |= value << self.n; self.n += nbits
        while self.n >= 8:
            out.append(self.v & 255); self.v >>= 8; self.n -= 8
    def align(self, out):
        if self.n: out.append(self.v & 255)
        self.n = 0; self.v = 0


def stream(eng, data, level, calls, strategy=0, more_at=1 << 62):
    """calls: [(upto, flush)], flush 0 none / 1 partial / 2 sync / 3 full; the Z_FINISH call is implied.  more_at: a Z_NO_FLUSH call feeds the engine
    (ZGPU_CONT_MORE) once that many unparsed bytes have gathered."""
    cs, carry = eng.cont_new()
    out = bytearray()
    excl = []          # stream positions that are in no chain
    floor = 0          # history in front of this position is gone (Z_FULL_FLUSH)
    fed = 0            # bytes the caller has handed over
    checked = 0        # bytes that have been through a feed's checksum
    for upto, flush in list(calls) + [(len(data), 4)]:
        fed = upto
        unparsed = fed - cs.entry
        if flush == 0 and unparsed < more_at:
            continue
        lo = max(floor, cs.entry - HIST) if cs.entry > HIST else floor
        if cs.entry - cs.block_start <= 65536 + 512: lo = min(lo, cs.block_start)
        cs.abs0 = lo
        mode = gpu.CONT_MORE if flush == 0 else gpu.CONT_FINISH if flush == 4 else gpu.CONT_FLUSH
        if mode == gpu.CONT_MORE and fed - cs.entry <= 512 + 1:
            continue
        z = eng.deflate_cont_host(data[lo:fed], checked - lo, level, mode, cs, carry, strategy=strategy, excl=[p for p in excl if p >= lo])
        checked = fed
        out += z
        if flush in (1, 2, 3):
            t = BitTail(); t.n = cs.bit_count; t.v = cs.bit_value
            if flush == 1:  # _tr_align, trees.c:892-915
                t.put(out, 2, 3); t.put(out, 0, 7)
                if 1 + cs.last_eob + 10 - t.n < 9:
                    t.put(out, 2, 3); t.put(out, 0, 7)
                cs.last_eob = 7
            else:           # the empty stored block, trees.c:867-879
                t.put(out, 0, 3); t.align(out); out += b"\x00\x00\xff\xff"; cs.last_eob = 8
            cs.bit_count = t.n; cs.bit_value = t.v
            if flush == 3:
                floor = fed; excl = []; carry[1][:] = 0
            else:
                excl += [p for p in (fed - 2, fed - 1) if p >= floor and p >= 0 and p not in excl]
        excl = [p for p in excl if p + 40000 > cs.entry]
    return bytes(out)


def corpus(kind, seed, nbytes):
    return CP.chunks(kind, seed, (nbytes + 65535) // 65536).tobytes()[:nbytes]


LEVELS = [int(x) for x in (sys.argv[1] if __name__ == "__main__" and len(sys.argv) > 1 else "6,9,4").split(",")]


def main():
    eng = zlib_amd.Engine(0)
    rnd = random.Random(11)
    d = corpus(CP.KIND_SILESIA, 11, 3 << 20)
    bad = 0
    tests = [((), 1 << 62), (((len(d) // 2, 2),), 1 << 62), (((len(d) // 3, 0), (len(d) // 2, 3), (len(d) * 3 // 4, 1)), 1 << 62),
             (tuple((i, 0) for i in range(1000, len(d), 33333)), 100000), (tuple((i, 0) for i in range(65536, len(d), 65536)), 300000),
             (tuple((i, 2) for i in range(50000, len(d), 250001)), 1 << 62), (tuple((i, 1) for i in range(70000, len(d), 300007)), 1 << 62)]
    for it in range(6):
        calls = []; pos = 0
        while True:
            pos += rnd.choice([1, 5, 100, 4096, 32768 - 262 + rnd.randrange(0, 300), 65536, 200000])
            if pos >= len(d): break
            calls.append((pos, rnd.choice([0, 0, 0, 1, 2, 3])))
        tests.append((tuple(calls), rnd.choice([70000, 200000, 1 << 62])))
    for calls, more_at in tests:
        for level in LEVELS:
            want = R.deflate_calls(d, level, calls)
            t = time.time()
            try:
                got = stream(eng, d, level, calls, more_at=more_at)
            except Exception as ex:
                print("ERR", level, len(calls), more_at, ex, flush=True); bad += 1; continue
            ok = got == want
            bad += not ok
            k = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), min(len(got), len(want)))
            print("%s level %d, %d calls (first %s), feed at %d: ref %d, got %d%s  %.0f ms" % ("ok  " if ok else "DIFF", level, len(calls), calls[:2], more_at, len(want), len(got),
                                                                                         "" if ok else ", first difference at %d" % k, (time.time() - t) * 1e3), flush=True)
    print("bad", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
