"""Development check of the continuous stream on the GPU box: zgpu_deflate_host(ZGPU_F_CONTINUOUS) against the compiled reference's compress2()."""
import sys, os, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zlib_amd
from zlib_amd import gpu
from oracle import refzlib as R, corpus_py as CP


def corpus(kind, seed, nbytes):
    return CP.chunks(kind, seed, (nbytes + 65535) // 65536).tobytes()[:nbytes]


def main():
    eng = zlib_amd.Engine(0)
    rnd = random.Random(5)
    cases = [("hello1M", (b"hello, hello! " * 80000)[:1 << 20]), ("sil300k", corpus(CP.KIND_SILESIA, 3, 300000)), ("log200k", corpus(1, 4, 200001)),
             ("sil65537", corpus(CP.KIND_SILESIA, 9, 65537)), ("rand200k", bytes(rnd.getrandbits(8) for _ in range(200000))), ("tiny", b"abc"), ("empty", b""),
             ("sil3M", corpus(CP.KIND_SILESIA, 11, 3 << 20)), ("sil100", corpus(CP.KIND_SILESIA, 2, 100)), ("sil65024", corpus(CP.KIND_SILESIA, 2, 65024)),
             ("sil65025", corpus(CP.KIND_SILESIA, 2, 65025)), ("sil97536", corpus(CP.KIND_SILESIA, 2, 97536))]
    levels = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "6,4,9,5,7,8".split(","))]
    bad = 0
    for name, d in cases:
        for level in levels:
            want = R.compress2(d, level)
            t = time.time()
            try:
                got = eng.deflate_host(d, level, flags=gpu.F_FINAL | gpu.F_ZLIB_WRAP | gpu.F_CONTINUOUS)
            except Exception as ex:
                print("ERR", name, level, ex, flush=True); bad += 1; continue
            dt = time.time() - t
            if got != want:
                bad += 1
                k = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), min(len(got), len(want)))
                print("DIFF %s level %d: ref %d bytes, got %d, first difference at %d  (%.1f ms)" % (name, level, len(want), len(got), k, dt * 1e3), flush=True)
            else:
                print("ok   %s level %d: %d -> %d bytes (%.1f ms)" % (name, level, len(d), len(got), dt * 1e3), flush=True)
    print("bad", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
