"""Development sweep of the continuous stream on the GPU box: edge sizes around the window slides on data with candidates exactly MAX_DIST back,
strategies, several batches per feed, a large input -- all against the compiled reference driven the same way."""
import sys, os, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zlib_amd
from zlib_amd import gpu
from oracle import refzlib as R, corpus_py as CP


def corpus(kind, seed, nbytes):
    return CP.chunks(kind, seed, (nbytes + 65535) // 65536).tobytes()[:nbytes]


def main():
    eng = zlib_amd.Engine(0)
    r = random.Random(3)
    base = bytes(r.getrandbits(8) for _ in range(32506))
    words = [bytes(r.choice(b"abcdefghij ") for _ in range(r.randrange(2, 9))) for _ in range(300)]
    tb = b"".join(r.choice(words) for _ in range(9000))[:32506]
    srcs = {"per": base * 8, "pert": tb * 8, "rnd": bytes(r.getrandbits(8) for _ in range(140000))}
    sizes = set()
    for ps in (65274, 98042, 130810):
        for d in (0, 1, 2, 3, 100, 259, 260, 261, 262, 263, 300):
            sizes.add(ps + d)
    for s in (65536, 65537, 65024, 65025, 65023, 97536, 97537, 32512, 32513, 1, 2, 3, 4, 262, 263):
        sizes.add(s)
    LV = [int(x) for x in os.environ.get("LEVELS", "4,6,9").split(",")]
    FASTONLY = max(LV) <= 3
    bad = runs = 0
    F = gpu.F_FINAL | gpu.F_CONTINUOUS
    t0 = time.time()
    for name, src in srcs.items():
        for n in sorted(sizes):
            if n > len(src):
                continue
            for level in LV:
                runs += 1
                want = R.deflate_calls(src[:n], level)
                got = eng.deflate_host(src[:n], level, flags=F)
                if got != want:
                    bad += 1
                    print("DIFF edge", name, n, level, len(want), len(got), flush=True)
    print("edge sizes: runs", runs, "bad", bad, "%.1fs" % (time.time() - t0), flush=True)
    d = corpus(CP.KIND_SILESIA, 21, 1 << 20)
    for strat in ((1, 4) if FASTONLY else (1, 2, 3, 4)):
        for level in LV:
            want = R.deflate_calls(d, level, strategy=strat)
            got = eng.deflate_host(d, level, flags=F, strategy=strat)
            if got != want:
                bad += 1
                print("DIFF strategy", strat, level, len(want), len(got), flush=True)
    print("strategies done, bad", bad, flush=True)
    for name, d in (("a*2M", b"a" * (2 << 20)), ("ab*1M", b"ab" * (1 << 19)), ("abc*1M+", (b"abc" * 400000)[:1000001]), ("zeros65k", bytes(65536 * 3))):
        for level, strat in ([(l, 0) for l in LV] if FASTONLY else ((4, 0), (6, 0), (9, 0), (6, 3), (9, 3))):
            t = time.time()
            want = R.deflate_calls(d, level, strategy=strat)
            got = eng.deflate_host(d, level, flags=F, strategy=strat)
            if got != want:
                bad += 1
                print("DIFF runs", name, level, strat, len(want), len(got), flush=True)
            else:
                print("ok runs", name, level, strat, len(got), "%.1f ms" % ((time.time() - t) * 1e3), flush=True)
    d = corpus(CP.KIND_SILESIA, 22, 5 << 20) + bytes(r.getrandbits(8) for _ in range(300000)) + corpus(1, 5, 2 << 20)
    for bt in ("1", "3", "7", "64"):
        os.environ["ZGPU_CONT_BATCH_TILES"] = bt
        for level in LV[:2]:
            want = R.compress2(d, level)
            got = eng.deflate_host(d, level, flags=F | gpu.F_ZLIB_WRAP)
            if got != want:
                bad += 1
                print("DIFF batches of", bt, "level", level, len(want), len(got), flush=True)
    del os.environ["ZGPU_CONT_BATCH_TILES"]
    print("batches done, bad", bad, flush=True)
    big = corpus(CP.KIND_SILESIA, 0x5EED, 256 << 20)
    for level in LV[:1]:
        t = time.time(); want = R.compress2(big, level); tr = time.time() - t
        t = time.time(); got = eng.deflate_host(big, level, flags=F | gpu.F_ZLIB_WRAP); tg = time.time() - t
        print("256 MiB level %d: reference %.1f s, device %.3f s (host buffers), %d bytes, %s" % (level, tr, tg, len(got), "identical" if got == want else "DIFFERENT"), flush=True)
        bad += got != want
    print("bad", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
