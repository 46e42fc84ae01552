"""Rate of the continuous stream (ZGPU_F_CONTINUOUS) next to the chunked stream (mode B) on device-resident input, with the engine's stage times.
usage: cont_rate.py [gib] [levels] [check]   -- check: compare a 1 GiB prefix run with the compiled reference, inflate the big stream with the system zlib"""
import sys, os, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zlib_amd
from zlib_amd import gpu


def main():
    gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    levels = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "6,9,4").split(",")]
    check = len(sys.argv) > 3 and sys.argv[3] == "check"
    kind = int(os.environ.get("KIND", "0"))
    eng = zlib_amd.Engine(0)
    dev = torch.device("cuda", 0)
    nchunks = int(gib * 2**30) // 65536
    n = nchunks * 65536
    src = torch.empty(n, dtype=torch.uint8, device=dev)
    eng.corpus_fill_device(kind, 0x5EED5117 if kind == 0 else 0x10C7E47, 0, nchunks, src.data_ptr())
    cap = max(eng.L.zgpu_deflate_bound(n, 65536), eng.L.zgpu_deflate_cont_bound(n)) + 64
    dst = torch.empty(cap, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for level in levels:
        for name, flags in (("continuous", gpu.F_FINAL | gpu.F_ZLIB_WRAP | gpu.F_CONTINUOUS), ("chunks", gpu.F_FINAL | gpu.F_ZLIB_WRAP)):
            res = eng.deflate_device(src.data_ptr(), n, level, dst.data_ptr(), cap, flags=flags, stream=st)
            eng.profile(True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            steps = 2
            for _ in range(steps):
                res = eng.deflate_device(src.data_ptr(), n, level, dst.data_ptr(), cap, flags=flags, stream=st)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
            pr = eng.profile_read(); eng.profile(False)
            print("level %d %-10s %.2f GiB: %7.1f ms = %6.2f GiB/s, ratio %.4f, stages/ms %s" % (
                level, name, n / 2**30, dt * 1e3, n / dt / 2**30, n / res.out_bytes, {k: round(v[0] / steps, 1) for k, v in pr.items() if v[1]}), flush=True)
            if check and name == "continuous":
                z = dst[: res.out_bytes].cpu().numpy().tobytes()
                t0 = time.time()
                d = zlib.decompressobj()
                got = 0; ad = 1; ok = True; pos = 0
                host = src.cpu().numpy()
                for off in range(0, len(z), 1 << 26):
                    piece = d.decompress(z[off: off + (1 << 26)])
                    if piece != host[pos: pos + len(piece)].tobytes(): ok = False
                    pos += len(piece)
                piece = d.flush(); ok = ok and piece == host[pos: pos + len(piece)].tobytes(); pos += len(piece)
                print("   system zlib inflates the stream to %d bytes, %s (%.1f s)" % (pos, "equal to the input" if ok and pos == n and d.eof else "DIFFERENT", time.time() - t0), flush=True)
    if check:
        from oracle import refzlib as R
        m = min(n, 1 << 30)
        host = src[:m].cpu().numpy().tobytes()
        for level in levels[:1]:
            t0 = time.time(); want = R.deflate_calls(host, level, wbits=15); tr = time.time() - t0
            res = eng.deflate_device(src.data_ptr(), m, level, dst.data_ptr(), cap, flags=gpu.F_FINAL | gpu.F_ZLIB_WRAP | gpu.F_CONTINUOUS, stream=st)
            got = dst[: res.out_bytes].cpu().numpy().tobytes()
            print("level %d, %d MiB against the compiled reference (%.1f s on one core): %s, %d bytes" % (level, m >> 20, tr, "identical" if got == want else "DIFFERENT", len(got)), flush=True)


if __name__ == "__main__":
    main()
