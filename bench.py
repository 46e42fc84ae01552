#!/usr/bin/env python3
"""bench.py -- GiB/s of raw input compressed (deflate level 6) on N MI355X, one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): 4 GiB of the seeded synthetic Silesia-mix per GPU (65 536 chunks of 64 KiB,
zlib_amd/csrc/corpus.h), generated directly in HBM before the timed region.  A step = one pass of the hot path over
that batch: LZ77 + Huffman + stitch into one RFC 1950 stream per GPU; for N > 1 the per-GPU streams are gathered
to rank 0 over RCCL inside the step (the path's only exchange).  Weak scaling: rank r compresses chunks
[r*65536, (r+1)*65536) of the corpus.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel stage, timed with HIP events on the launch
stream (zgpu_profile_*); `cpu_baseline` is the compiled reference (or the repo's CPU restatement when
oracle/_ref is absent) timed on the host cores over a bounded sample of the same workload, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--level", type=int, default=6)
    ap.add_argument("--gib", type=float, default=None, help="GiB of input per GPU (default: 4 at N = 1, 8 at N > 1)")
    ap.add_argument("--workload", default=None, choices=["silesia-mix", "log-text"],
                    help="default: silesia-mix at N = 1 (BASELINE.json configs[1]), log-text at N > 1 (configs[4]: 8 GiB of it per rank)")
    ap.add_argument("--lz", default="auto", choices=["auto", "serial", "parallel", "sorted", "walk", "fast", "fastwin"])
    ap.add_argument("--op", default="deflate", choices=["deflate", "inflate"])
    ap.add_argument("--cpu-sample-mib", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--continuous", action="store_true", help="the timed step makes ONE continuous stream (what compress2() of the reference emits) instead of independent chunks: for profiles of that path")
    ap.add_argument("--no-extras", action="store_true", help="headline only: skip inflate / level 1 / level 9 / host-buffer runs")
    return ap.parse_args()


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cores():
    """(threads worth starting, logical CPUs of the host): the affinity mask, cut down to the cgroup's CPU quota when there is one
    (a GPU box hands a one-GPU job about 16 cores of a much larger host; more threads than that only take turns)"""
    try:
        host = len(os.sched_getaffinity(0))
    except AttributeError:
        host = os.cpu_count() or 1
    n = host
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(round(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return min(n, 64), host


def cpu_baseline(sample, level, threads, op="deflate"):
    """Time the reference (oracle/_ref) -- or the CPU restatement -- per 64 KiB chunk on `threads` host threads.
    op "inflate": the chunks are compressed first (untimed) and the timed part is the reference's inflate of those streams."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import refzlib, oracle_py
    use_ref = refzlib.available()
    nchunks = len(sample) // 65536
    if use_ref:
        refzlib.lib()
        enc = lambda b: refzlib.deflate_chunk_raw(b, level, True)
        dec = lambda z: refzlib.inflate_raw(z, 65536)[1]
    else:
        oracle_py.lib()
        enc = lambda b: oracle_py.deflate_chunk(b, level, True)
        dec = lambda z: oracle_py.inflate_raw(z, 65536)[1]
    parts = [range(t, nchunks, threads) for t in range(threads)]
    if op == "inflate":
        with ThreadPoolExecutor(threads) as ex:
            streams = sum(ex.map(lambda r: [(k, enc(sample[k * 65536:(k + 1) * 65536])) for k in r], parts), [])
        streams = dict(streams)

        def work(r):
            n = 0
            for k in r:
                n += len(dec(streams[k]))
            return n
    else:
        def work(r):
            n = 0
            for k in r:
                n += len(enc(sample[k * 65536:(k + 1) * 65536]))
            return n
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        out_bytes = sum(ex.map(work, parts))
    dt = time.perf_counter() - t0
    what = "inflate of the level-%d streams of the" % level if op == "inflate" else "level %d over the" % level
    return {"value": round(len(sample) / dt / 2**30, 4), "unit": "GiB/s", "cores": threads,
            "kind": "reference" if use_ref else "port",
            "sample": "%s first %d MiB of the workload, 64 KiB chunks, %.1f s wall%s" % (
                what, len(sample) >> 20, dt, "" if op == "inflate" else ", ratio %.3f" % (len(sample) / out_bytes))}


def check_sampled_chunks(torch, dst, offs, nchunks, rank_chunk0, level, workload, nsample=64, final=True, fail=sys.exit):
    """The claim "bit-exact" checked in the run that makes it: `nsample` of the chunks the committed fixture samples (tests/golden/corpus_*.json: the
    REFERENCE's output length and SHA-256 per chunk at levels 1 / 6 / 9, generated by oracle/gen_golden.py from the compiled reference) are cut out of the
    stream just produced and compared.  Returns the number checked, or exits.  (None: no fixture for this level or workload.)"""
    import hashlib
    name = {"silesia-mix": "corpus_silesia.json", "log-text": "corpus_logtext.json"}[workload]
    col = {1: 2, 6: 4, 9: 6}.get(level)
    path = os.path.join(ROOT, "tests", "golden", name)
    if col is None or not os.path.exists(path):
        return None
    g = json.load(open(path))
    rows = [r for r in g["rows"] if rank_chunk0 <= r[0] < rank_chunk0 + nchunks - 1]
    rows = rows[:: max(1, len(rows) // nsample)][:nsample]
    if final:  # this range ends the stream: its last chunk carries BFINAL, the variant the fixture's `last_rows` hold
        rows += [r for r in g.get("last_rows", []) if r[0] == rank_chunk0 + nchunks - 1]
    if not rows:
        return None
    idx = torch.tensor([r[0] - rank_chunk0 for r in rows], dtype=torch.int64, device=offs.device)
    lo = offs[idx].cpu().tolist()
    hi = offs[idx + 1].cpu().tolist()
    for r, a, b in zip(rows, lo, hi):
        seg = dst[a:b].cpu().numpy().tobytes()
        if [len(seg), hashlib.sha256(seg).hexdigest()[:16]] != r[col:col + 2]:
            return fail("chunk %d at level %d differs from the reference's output (%d bytes, fixture %d)" % (r[0], level, len(seg), r[col]))
    return len(rows)


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    import zlib_amd
    from zlib_amd import gpu, shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("launch with torch.distributed.run for --gpus > 1")
    # N = 1: the configuration BASELINE.json's metric is quoted on (4 GiB silesia-mix); N > 1: its multi-GPU configuration (SURVEY.md 8d config 5:
    # log-text, 8 GiB per rank, rank r compresses chunks [131072 r, 131072 (r + 1)) of the corpus tests/golden/corpus_logtext.json samples)
    if a.workload is None:
        a.workload = "silesia-mix" if world == 1 else "log-text"
    if a.gib is None:
        a.gib = 4.0 if world == 1 else 8.0
    # ZAMD_BENCH_SHARE_GPU=1: a REHEARSAL of the N > 1 flow on a box with one GPU (every rank on GPU 0, gloo for the process group, and
    # ZAMD_RCCL_LIB naming the test double of tests/tools/fake_rccl.cpp).  Its JSON line says so; it is no measurement.
    share = os.environ.get("ZAMD_BENCH_SHARE_GPU") == "1"
    if share:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cdev = torch.device("cpu") if share else dev  # where the few control tensors of the process group live
    if world > 1:
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    eng = zlib_amd.Engine(local)
    comm, comm_note = None, None
    if world > 1:
        def exchange_id(b):  # rank 0's RCCL id to everybody, over the group that exists for the barrier
            t = torch.tensor(list(b), dtype=torch.uint8, device=cdev)
            dist.broadcast(t, 0)
            return bytes(t.cpu().tolist())
        comm_note = "zgpu_deflate_gather (RCCL inside the C library)"
        try:
            comm = gpu.Comm(local, world, rank, exchange_id)
        except Exception as ex:  # the ranks agree below on what carries the gather
            comm, comm_note = None, "torch.distributed gather (the C library's communicator could not be made: %s)" % ex
        agreed = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=cdev)
        dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
        if int(agreed.item()) == 0 and comm is not None:
            comm.close()
            comm, comm_note = None, "torch.distributed gather (another rank could not make the C library's communicator)"
    kind = 0 if a.workload == "silesia-mix" else 1
    seed = 0x5EED5117 if kind == 0 else 0x10C7E47
    nchunks = int(a.gib * 2**30) // 65536
    nbytes = nchunks * 65536
    src = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    eng.corpus_fill_device(kind, seed, rank * nchunks, nchunks, src.data_ptr())
    cap = eng.L.zgpu_deflate_bound(nbytes, 65536)
    if a.continuous:
        if world > 1 or a.op != "deflate":
            sys.exit("--continuous: one GPU, deflate")
        cap = max(cap, eng.L.zgpu_deflate_cont_bound(nbytes) + 64)
    dst = torch.empty(cap, dtype=torch.uint8, device=dev)
    offs = torch.empty(nchunks + 1, dtype=torch.int64, device=dev)
    lz = {"auto": gpu.LZ_AUTO, "serial": gpu.LZ_SERIAL, "parallel": gpu.LZ_PARALLEL, "sorted": gpu.LZ_SORTED, "walk": gpu.LZ_WALK, "fast": gpu.LZ_FAST, "fastwin": gpu.LZ_FASTWIN}[a.lz]
    stream = torch.cuda.current_stream().cuda_stream
    gather_buf = None
    state = {}

    def deflate_step():
        if world == 1:
            state["res"] = eng.deflate_device(src.data_ptr(), nbytes, a.level, dst.data_ptr(), cap, flags=gpu.F_FINAL | gpu.F_ZLIB_WRAP | (gpu.F_CONTINUOUS if a.continuous else 0),
                                              lz_impl=lz, d_offsets=None if a.continuous else offs.data_ptr(), stream=stream)
            return
        # N > 1: every rank emits the raw body of its chunk range (BFINAL only on the last rank's last chunk); rank 0 gathers
        # the bodies over RCCL and frames them into one RFC 1950 stream (zlib_amd/shard.py)
        res = eng.deflate_device(src.data_ptr(), nbytes, a.level, dst.data_ptr(), cap, flags=gpu.F_FINAL if rank == world - 1 else 0,
                                 lz_impl=lz, d_offsets=offs.data_ptr(), stream=stream)
        state["res"] = res
        nonlocal gather_buf
        if comm is None:  # zlib_amd/shard.py: the same exchange over the process group
            body = dst[: res.out_bytes]
            _, total = shard.gather_stream(body, res.adler32, nbytes, a.level, out=gather_buf if rank == 0 else None)
            if rank == 0 and (gather_buf is None or gather_buf.numel() < total):
                gather_buf = torch.empty(total + (total >> 4), dtype=torch.uint8, device=dev)  # (the next step reuses it)
            state["gathered"] = total
            return
        # the C library's RCCL gather (include/zamd_gpu.h zgpu_deflate_gather): sizes first, so that rank 0's buffer is exactly as large as the stream
        table, total = comm.sizes(res.out_bytes, res.adler32, nbytes, stream=stream)
        if rank == 0 and (gather_buf is None or gather_buf.numel() < total):
            gather_buf = torch.empty(total + (total >> 4), dtype=torch.uint8, device=dev)
        comm.gather(dst.data_ptr(), table, a.level, gather_buf.data_ptr() if rank == 0 else None, gather_buf.numel() if rank == 0 else 0, stream=stream)
        state["gathered"] = total

    def inflate_step():
        r = eng.inflate_device(state["z_ptr"], state["z_len"], offs.data_ptr(), nchunks, src2.data_ptr(), nbytes, stream=stream)
        state["ires"] = r

    step = deflate_step
    if a.op == "inflate":
        eng.inflate_set_checks(gpu.CHECK_ADLER32)  # a zlib stream: the check its trailer holds
        deflate_step()
        res = state["res"]
        # raw body without the 2-byte header: offsets are relative to the stream start, which includes the header
        state["z_ptr"], state["z_len"] = dst.data_ptr(), res.out_bytes
        src2 = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        step = inflate_step

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, steps, warmup):
        """warmup untimed calls, then exactly `steps` calls between barrier + synchronize; returns (seconds, per-stage device time)"""
        for _ in range(warmup):
            fn()
        eng.profile(True)
        sync_all()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        sync_all()
        dt = time.perf_counter() - t0
        prof = eng.profile_read()
        eng.profile(False)
        return dt, prof

    dt, prof = timed(step, a.steps, a.warmup)
    per_rank = None
    if world > 1:
        # every rank checks ITS sampled chunks against the reference's hashes (the last rank's last chunk carries BFINAL: the fixture's last_rows) and
        # times its own steps (device events of its stages); rank 0 reports all of them and the job fails when any rank's bytes differ
        own_ms = sum(v[0] for v in prof.values() if v[1]) / a.steps
        bad = []
        nchk = check_sampled_chunks(torch, dst, offs, nchunks, rank * nchunks, a.level, a.workload, final=rank == world - 1, fail=bad.append) if a.op == "deflate" else None
        mine = torch.tensor([dt, own_ms, -1.0 if bad else float(nchk or 0), float(state["res"].out_bytes)], dtype=torch.float64, device=cdev)
        table = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(table, mine)
        table = [t.cpu().tolist() for t in table]
        if bad:
            print("rank %d: %s" % (rank, bad[0]), file=sys.stderr, flush=True)
        if any(t[2] < 0 for t in table):
            sys.exit("rank(s) %s produced bytes that differ from the reference's" % [r for r, t in enumerate(table) if t[2] < 0])
        dt = max(t[0] for t in table)
        per_rank = {"wall_ms_per_step": [round(t[0] / a.steps * 1e3, 2) for t in table], "device_ms_per_step": [round(t[1], 2) for t in table],
                    "chunks_checked_against_reference_hashes": [int(t[2]) for t in table], "body_bytes": [int(t[3]) for t in table]}

    if a.op == "inflate":
        ok = bool(torch.equal(src, src2))
        if not ok:
            sys.exit("inflate output differs from the original input")

    def roofline_of(prof, steps, in_bytes, out_bytes, op, level):
        """Roofline object for the dominant stage of a run: algorithmic bytes (input read + stream written) per launch over the
        stage's HIP-event time per launch (events recorded by the engine on its launch stream)."""
        stages = {k: v for k, v in prof.items() if v[1] > 0}
        if not stages:
            return None
        dom = max(stages, key=lambda k: stages[k][0])
        ms, launches = stages[dom]
        per_launch_bytes = (in_bytes + out_bytes) * steps / launches
        achieved = per_launch_bytes / (ms / launches * 1e-3) / 1e9
        # HBM traffic of the dominant kernel cannot be counted from inside this process (rocprofv3 collects FETCH_SIZE and WRITE_SIZE in
        # separate passes, scripts/prof_cache.sh); the committed counters are quoted when they are for this very workload and kernel.
        traffic, traffic_src, limiter = None, None, None
        try:
            tj = {}
            for name in ("r02_traffic.json", "r03_traffic.json", "r04_traffic.json"):  # (the later round's counters replace the earlier ones for the same kernel and workload)
                fn = os.path.join(ROOT, "profiles", name)
                if os.path.exists(fn):
                    tj.update(json.load(open(fn)))
            key = "%s-L%d-%s-%.0fgib-%s" % (op, level, a.workload, in_bytes / 2**30, dom)
            if key in tj and (op != "deflate" or a.lz == "auto"):
                traffic, traffic_src, limiter = tj[key]["traffic_bytes_per_launch"], tj[key]["source"], tj[key].get("limiter")
        except (OSError, ValueError, KeyError):
            pass
        return {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": int(per_launch_bytes), "limiter": limiter,
                "avg_launch_ms": round(ms / launches, 4), "launches": launches,
                "stage_ms_per_step": {k: round(v[0] / steps, 3) for k, v in stages.items()}}

    if rank == 0:
        res = state["res"]
        ratio = nbytes / res.out_bytes
        total_in = nbytes * world * a.steps
        value = total_in / dt / 2**30
        line = {
            "metric": "GiB/s raw input %s (deflate level %d)" % ("decompressed" if a.op == "inflate" else "compressed", a.level),
            "value": round(value, 4), "unit": "GiB/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s %.2f GiB per GPU, %s, level %d, bit-exact vs zlib 1.2.3" % (
                a.workload, nbytes / 2**30, "ONE continuous stream" if a.continuous else "64 KiB independent chunks", a.level), "op": a.op, "lz": a.lz, "chunks_per_gpu": nchunks,
                "compression_ratio": round(ratio, 4), "stream_bytes": int(res.out_bytes),
                "gathered_bytes": int(state.get("gathered", res.out_bytes)), "gather": comm_note,
                **({"rehearsal": "ZAMD_BENCH_SHARE_GPU: all ranks on one GPU, no measurement"} if share else {}),
                "chunks_checked_against_reference_hashes": (sum(per_rank["chunks_checked_against_reference_hashes"]) if per_rank else
                                                            check_sampled_chunks(torch, dst, offs, nchunks, 0, a.level, a.workload) if a.op == "deflate" and not a.continuous else None),
                **({"per_rank": per_rank} if per_rank else {})},
            "roofline": roofline_of(prof, a.steps, nbytes, res.out_bytes, "deflate-continuous" if a.continuous else a.op, a.level),
        }
    # ---- the other configurations of BASELINE.json on the same input, outside the timed region of the headline (N = 1 only):
    #      inflate of the headline's stream (config 4), level 1 and level 9 (config 3), and the host-buffer entry point (PCIe included)
    if world == 1 and a.op == "deflate" and not a.no_extras and not a.continuous:
        extra = {}
        z_len = state["res"].out_bytes
        offs2 = torch.empty(nchunks + 1, dtype=torch.int64, device=dev)

        def leg_cpu_baseline(level, op):
            # the reference on the host cores for this leg, on a sample sized to a few seconds (level 9 is 3.5 times slower per byte than level 6)
            if a.no_cpu_baseline:
                return None
            mib = {1: 512, 9: 96}.get(level, 256) if op == "deflate" else 512
            sample = src[: min(nbytes, mib << 20)].cpu().numpy().tobytes()
            threads, host_cpus = usable_cores()
            cb = cpu_baseline(sample, level, threads, op)
            cb["host_logical_cpus"] = host_cpus
            cb["cpu_model"] = cpu_model()
            return cb

        def one_level(level, steps):
            st = {}

            def f():
                st["res"] = eng.deflate_device(src.data_ptr(), nbytes, level, dst2.data_ptr(), cap, flags=gpu.F_FINAL | gpu.F_ZLIB_WRAP,
                                               lz_impl=gpu.LZ_AUTO, d_offsets=offs2.data_ptr(), stream=stream)
            d, pr = timed(f, steps, 1)
            return {"metric": "GiB/s raw input compressed (deflate level %d)" % level, "value": round(nbytes * steps / d / 2**30, 4),
                    "unit": "GiB/s", "steps": steps, "warmup": 1, "ms_per_step": round(d / steps * 1e3, 3),
                    "compression_ratio": round(nbytes / st["res"].out_bytes, 4),
                    "chunks_checked_against_reference_hashes": check_sampled_chunks(torch, dst2, offs2, nchunks, 0, level, a.workload),
                    "roofline": roofline_of(pr, steps, nbytes, st["res"].out_bytes, "deflate", level),
                    "cpu_baseline": leg_cpu_baseline(level, "deflate")}

        # config 4: the stream the headline produced, back to the original bytes (checked)
        src2 = torch.empty(nbytes, dtype=torch.uint8, device=dev)

        def inf():
            state["ires"] = eng.inflate_device(dst.data_ptr(), z_len, offs.data_ptr(), nchunks, src2.data_ptr(), nbytes, stream=stream)
        eng.inflate_set_checks(gpu.CHECK_ADLER32)  # a zlib stream: the check its trailer holds (as inflate() of the host library asks)
        d, pr = timed(inf, a.steps, 1)
        eng.inflate_set_checks(gpu.CHECK_ADLER32 | gpu.CHECK_CRC32)
        if state["ires"].adler32 != state["res"].adler32:
            sys.exit("inflate: Adler-32 of the output differs from the stream's")
        if not bool(torch.equal(src, src2)):
            sys.exit("inflate output differs from the original input")
        extra["inflate"] = {"metric": "GiB/s raw output decompressed (inflate of the level-%d stream)" % a.level,
                            "value": round(nbytes * a.steps / d / 2**30, 4), "unit": "GiB/s", "steps": a.steps, "warmup": 1,
                            "ms_per_step": round(d / a.steps * 1e3, 3), "bytes_equal": True,
                            "roofline": roofline_of(pr, a.steps, nbytes, z_len, "inflate", a.level),
                            "cpu_baseline": leg_cpu_baseline(a.level, "inflate")}
        del src2
        dst2 = torch.empty(cap, dtype=torch.uint8, device=dev)
        extra["level1"] = one_level(1, a.steps)
        extra["level9"] = one_level(9, max(1, a.steps // 2))  # (the deepest chains: fewer steps, stated in "steps")
        del offs2
        # SURVEY.md 8f N1 (round 4): the same input as ONE continuous stream -- what plain compress2() of the reference emits: the window slides through the
        # whole input, matches cross every 64 KiB boundary, blocks are cut every 16383 tokens from the stream's start.  Next to the headline's independent
        # chunks: rate, ratio, and the first 256 MiB as a stream of their own against the compiled reference's length and SHA-256 (tests/golden/continuous_kat.json)
        def continuous(level, steps):
            import hashlib
            ccap = eng.L.zgpu_deflate_cont_bound(nbytes) + 64
            cdst = dst2 if ccap <= cap else torch.empty(ccap, dtype=torch.uint8, device=dev)
            st = {}

            def f():
                st["res"] = eng.deflate_device(src.data_ptr(), nbytes, level, cdst.data_ptr(), max(cap, ccap), flags=gpu.F_FINAL | gpu.F_ZLIB_WRAP | gpu.F_CONTINUOUS, stream=stream)
            d, pr = timed(f, steps, 2)  # (two untimed calls: the engine sizes its batches by the memory that is free, and grows them once more in the second call when this process holds a lot)
            out = {"metric": "GiB/s raw input compressed (deflate level %d), ONE continuous stream" % level, "value": round(nbytes * steps / d / 2**30, 4), "unit": "GiB/s",
                   "steps": steps, "warmup": 2, "ms_per_step": round(d / steps * 1e3, 3), "compression_ratio": round(nbytes / st["res"].out_bytes, 4),
                   "stream_bytes": int(st["res"].out_bytes), "roofline": roofline_of(pr, steps, nbytes, st["res"].out_bytes, "deflate-continuous", level)}
            try:
                rows = [r for r in json.load(open(os.path.join(ROOT, "tests", "golden", "continuous_kat.json")))["rows"]
                        if r["corpus"] == kind and r["level"] == level and r["sync_at"] is None and r["n"] == (256 << 20) and r["n"] <= nbytes]
            except (OSError, ValueError, KeyError):
                rows = []
            if rows and a.workload == "silesia-mix":
                r = rows[0]
                res = eng.deflate_device(src.data_ptr(), r["n"], level, cdst.data_ptr(), max(cap, ccap), flags=gpu.F_FINAL | gpu.F_ZLIB_WRAP | gpu.F_CONTINUOUS, stream=stream)
                z = cdst[: res.out_bytes].cpu().numpy().tobytes()
                if len(z) != r["len"] or hashlib.sha256(z).hexdigest() != r["sha256"]:
                    sys.exit("continuous stream of the first %d MiB at level %d differs from the reference's compress2() (%d bytes, fixture %d)" % (r["n"] >> 20, level, len(z), r["len"]))
                out["first_256_mib_equal_to_reference_compress2"] = True
            if not a.no_cpu_baseline:
                from oracle import refzlib
                if refzlib.available():  # one stream is one core's work: the reference's own loop cannot be cut into threads
                    sample = src[: min(nbytes, 48 << 20)].cpu().numpy().tobytes()
                    t0 = time.perf_counter(); zr = refzlib.compress2(sample, level); dt = time.perf_counter() - t0
                    out["cpu_baseline"] = {"value": round(len(sample) / dt / 2**30, 4), "unit": "GiB/s", "cores": 1, "kind": "reference",
                                           "sample": "compress2() of the first %d MiB, level %d, %.1f s, ratio %.3f" % (len(sample) >> 20, level, dt, len(sample) / len(zr)), "cpu_model": cpu_model()}
            return out
        extra["continuous"] = continuous(a.level, max(1, a.steps // 2))
        extra["continuous_level1"] = continuous(1, 1)
        del dst2
        # the zlib-API path hands over host buffers: H2D of the input, the same kernels, D2H of the stream (SURVEY.md 8d "end-to-end")
        import ctypes as C
        import numpy as np
        host_n = nbytes  # (the headline's workload: the two rates are of the same bytes)
        host_in = src[:host_n].cpu().numpy()
        host_cap = eng.L.zgpu_deflate_bound(host_n, 65536)
        host_out = np.zeros(host_cap, dtype=np.uint8)  # (touched: the timed call does not pay for page faults of a fresh allocation)
        hp = gpu._Params(a.level, 65536, gpu.F_FINAL | gpu.F_ZLIB_WRAP, gpu.LZ_AUTO, 0, 0)
        hres = gpu.DeflateResult()
        best = None
        for _ in range(3):  # the second call finds the engine's staging buffers allocated
            t0 = time.perf_counter()
            rc = eng.L.zgpu_deflate_host(eng.h, host_in.ctypes.data, host_n, C.byref(hp), host_out.ctypes.data, host_cap, None, C.byref(hres))
            d = time.perf_counter() - t0
            if rc != 0:
                sys.exit("zgpu_deflate_host failed: %d" % rc)
            best = d if best is None or d < best else best
        extra["end_to_end_host_buffers"] = {"metric": "GiB/s raw input compressed, host buffers in and out (H2D, kernels, D2H; pageable memory)",
                                            "value": round(host_n / best / 2**30, 4), "unit": "GiB/s", "input_bytes": host_n,
                                            "stream_bytes": int(hres.out_bytes), "note": "zgpu_deflate_host, best of three calls",
                                            "of_resident_rate": round(host_n / best / 2**30 / line["value"], 3)}
        # a stream that was NOT produced in chunks (the system zlib's level-6 output for the first 256 MiB), decoded in pieces at
        # block starts found by search; host buffers in and out (that is how such a stream arrives: uncompress(), inflate(), a zip member)
        import zlib as syszlib
        f_n = min(host_n, 256 << 20)
        f_in = host_in[:f_n].tobytes()
        co = syszlib.compressobj(6, syszlib.DEFLATED, -15)
        f_raw = co.compress(f_in) + co.flush()
        f_dst = np.zeros(f_n, dtype=np.uint8)
        before = eng.spec_counts()
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            f_out = eng.inflate_stream_host(f_raw, f_n, out=f_dst)
            d = time.perf_counter() - t0
            best = d if best is None or d < best else best
        after = eng.spec_counts()
        if f_out.tobytes() != f_in:
            sys.exit("foreign-stream inflate differs from the original input")
        extra["foreign_stream_inflate"] = {"metric": "GiB/s raw output decompressed, a %s stream not produced in chunks, host buffers in and out" % (
                                               "system zlib %s level-6" % syszlib.ZLIB_RUNTIME_VERSION),
                                           "value": round(f_n / best / 2**30, 4), "unit": "GiB/s", "output_bytes": f_n, "stream_bytes": len(f_raw),
                                           "ms": round(best * 1e3, 2), "bytes_equal": True, "decoded_in_pieces": after[0] - before[0],
                                           "one_workgroup_fallbacks": after[1] - before[1], "note": "zgpu_inflate_stream_host2, best of three calls"}
        del f_in, f_raw, f_dst
        line["extra"] = extra
    if rank == 0:
        if world == 1 and not a.no_cpu_baseline:
            sample_bytes = min(nbytes, a.cpu_sample_mib << 20)
            sample = src[:sample_bytes].cpu().numpy().tobytes()
            threads, host_cpus = usable_cores()
            cb = cpu_baseline(sample, a.level, threads, a.op)
            cb["host_logical_cpus"] = host_cpus
            one = cpu_baseline(sample[: min(len(sample), 48 << 20)], a.level, 1, a.op)
            cb["single_thread"] = {"value": one["value"], "unit": "GiB/s", "sample": one["sample"]}
            cb["cpu_model"] = cpu_model()
            line["cpu_baseline"] = cb
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
