"""Development check of the host library's continuous deflate() / compress2() on the GPU box against the compiled reference driven by the same calls."""
import sys, os, time, random, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import zhost as Z
from oracle import refzlib as R, corpus_py as CP


def corpus(kind, seed, nbytes):
    return CP.chunks(kind, seed, (nbytes + 65535) // 65536).tobytes()[:nbytes]


def main():
    rnd = random.Random(17)
    bad = 0
    hello = (b"hello, hello! " * 80000)[:1 << 20]
    for level in (6, 9, 4, 0, 1, 2, 3):
        rc, got = Z.compress2(hello, level)
        want = R.compress2(hello, level)
        ok = rc == 0 and got == want
        bad += not ok
        print("compress2(hello 1 MiB, %d): rc %d, %d bytes, reference %d: %s" % (level, rc, len(got), len(want), "identical" if ok else "DIFFERENT"), flush=True)
    d = corpus(CP.KIND_SILESIA, 11, 3 << 20)
    L = Z.lib()
    print("compressBound(1 MiB) = %d (reference %d)" % (L.compressBound(1 << 20), R.lib().compressBound(1 << 20)))
    bad += L.compressBound(1 << 20) != R.lib().compressBound(1 << 20)
    plans = [[(len(d), 4)], [(len(d) // 2, 2), (len(d) - len(d) // 2, 4)], [(1000000, 0), (500000, 3), (700000, 1), (len(d) - 2200000, 4)],
             [(100000, 2)] * 20 + [(len(d) - 2000000, 4)]]
    for it in range(5):
        plan = []; pos = 0
        while pos < len(d):
            n = min(len(d) - pos, rnd.choice([1, 5, 100, 4096, 32768 - 262 + rnd.randrange(0, 300), 65536, 200000, 900000]))
            pos += n
            plan.append((n, 4 if pos == len(d) else rnd.choice([0, 0, 0, 1, 2, 3])))
        plans.append(plan)
    for plan in plans:
        for level, wbits in ((6, 15), (9, -15), (4, 31), (0, 15), (1, 15), (2, -15), (3, 31)):
            for in_step, out_step in ((None, None), (30011, 4099)):
                calls = []; pos = 0
                for n, f in plan:
                    if in_step:  # the slices inside a piece are calls of their own (level 0 cuts its blocks by them)
                        for q in range(pos + in_step, pos + n, in_step):
                            calls.append((q, 0))
                    pos += n
                    if f != 4: calls.append((pos, f))
                want = R.deflate_calls(d, level, calls, wbits=wbits)
                t = time.time()
                got, codes, info = Z.deflate_stream(d, level, plan, in_step=in_step, out_step=out_step, window_bits=wbits)
                ok = got == want
                bad += not ok
                k = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), min(len(got), len(want)))
                print("%s deflate level %d wbits %d, %d pieces (steps %s/%s): ref %d got %d%s  %.0f ms" % ("ok  " if ok else "DIFF", level, wbits, len(plan), in_step, out_step, len(want), len(got),
                                                                                                      "" if ok else " first difference at %d" % k, (time.time() - t) * 1e3), flush=True)
    print("bad", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
