"""Condense the round's rocprofv3 counter summaries (profiles/r04_*_summary.txt, written by scripts/prof_round.sh + prof_summarize.py) into
profiles/r04_traffic.json: HBM-side bytes per launch of the dominant kernel of every bench leg, which bench.py quotes as roofline.traffic / .limiter."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def kernel_block(path, kernel):
    txt = open(path).read()
    calls = int(re.search(re.escape(kernel) + r".*?calls\s+(\d+)", txt).group(1))
    pmc = txt.split("== pmc")[1]
    m = re.search(r"^[^\n]*" + re.escape(kernel) + r"[^\n]*\n((?:    [^\n]*\n)+)", pmc, re.M)
    vals = {}
    for ln in m.group(1).splitlines():
        f = ln.split()
        if len(f) >= 2 and re.fullmatch(r"[A-Z_a-z0-9]+", f[0]) and re.fullmatch(r"\d+", f[1]):
            vals[f[0]] = int(f[1])
    return calls, vals


def entry(path, kernel, limiter):
    calls, v = kernel_block(os.path.join(P, path), kernel)
    fetch = v["FETCH_SIZE"] * 1024 * 2 // calls  # KiB -> bytes; x2: the gfx950 correction of MI355X_MICROARCH.md (128-B requests tallied at 64 B)
    write = v["WRITE_SIZE"] * 1024 // calls
    d = {"launches_profiled": calls, "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "traffic_bytes_per_launch": fetch + write,
         "source": "profiles/%s: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (scripts/prof_round.sh), KiB summed over the %d launches; FETCH_SIZE doubled per the gfx950 "
                   "correction of MI355X_MICROARCH.md (calibrated there for wide streaming reads; scattered narrow reads are not calibrated)" % (path, calls)}
    if "SQ_WAVE_CYCLES" in v and v["SQ_WAVE_CYCLES"]:
        wc = v["SQ_WAVE_CYCLES"]
        d["counters"] = {"valu_busy_pct_of_wave_cycles": round(100.0 * v.get("SQ_ACTIVE_INST_VALU", 0) / wc, 1), "wait_any_pct_of_wave_cycles": round(100.0 * v.get("SQ_WAIT_ANY", 0) / wc, 1),
                         "lds_bank_conflict_pct_of_lds_active": round(100.0 * v.get("SQ_LDS_BANK_CONFLICT", 0) / max(1, v.get("SQ_LDS_IDX_ACTIVE", 0)), 1)}
    d["limiter"] = limiter
    return d


out = {
    "deflate-L6-silesia-mix-4gib-match": entry("r04_final_4gib_L6_summary.txt", "walk_kernel<1>",
        "the dependent chain through a body of the walkers, not a pipe: VALU 64 %, LDS 40 % (half of it bank conflicts), scalar 36 % of a CU's cycles busy, a wave half its cycles in s_waitcnt; "
        "bodies 49 % / folds 14 % / passes 34 % / byte reads 7 % / waiting for groups 4 % of a wave's cycles (profiles/r04_walk_phases.txt); removing 5 % of the instructions gained 2.4 % "
        "(DESIGN.md section 4 'What round 4 tried on the walkers'); not HBM bandwidth, but 50x the algorithmic bytes cross the fabric"),
    "deflate-L9-silesia-mix-4gib-match": entry("r04_L9_4gib_summary.txt", "walk_kernel<1>",
        "the same kernel with 4096-deep chains: bodies 61 % of a wave's cycles, 44 candidate steps per byte (profiles/r04_walk_phases.txt)"),
    "deflate-L1-silesia-mix-4gib-lz_serial": entry("r04_L1_4gib_summary.txt", "lz_serial_kernel<false>",
        "latency of the slowest chunks' chains of dependent scattered reads (head, candidate bytes, prev): one lane per chunk, a launch lasts as long as the chunk with the most "
        "tokens; the chunks that do not compress go to the wave-per-chunk kernel after 4 KiB (unchanged since round 3)"),
    "deflate-continuous-L6-silesia-mix-4gib-match": entry("r04_continuous_4gib_L6_summary.txt", "walk_kernel<2>",
        "the walkers of a tile (32 KiB of new positions behind 32 KiB of history): a chunk's fixed costs -- staging 64 KiB, a thousand walker starts, ramp and tail of the walker loop, "
        "the exit function -- for half a chunk's work; the tile sort (every byte twice) and the tokens from the true entries (parse2_kernel<true, true>) are kernels of their own"),
    "inflate-L6-silesia-mix-4gib-inflate": entry("r04_inflate_4gib_L6_summary.txt", "inflate_kernel_t<false, 8192u>",
        "latency of the per-token chains of one reader and one writer wave per segment, ten segments per CU (15 KiB of LDS each, 96 registers a lane); matches that reach farther back "
        "than the 8 KiB ring (13 %) read the destination; not HBM"),
}
json.dump(out, open(os.path.join(P, "r04_traffic.json"), "w"), indent=1)
for k, v in out.items():
    print("%-46s traffic %7.2f GB per launch (fetch %7.2f, write %6.2f)  %s" % (k, v["traffic_bytes_per_launch"] / 1e9, v["fetch_bytes_per_launch"] / 1e9, v["write_bytes_per_launch"] / 1e9, v.get("counters")))
