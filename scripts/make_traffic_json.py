"""Condense the round's rocprofv3 counter summaries (profiles/r03_*_summary.txt, written by scripts/prof_round.sh + prof_summarize.py) into
profiles/r03_traffic.json: HBM-side bytes per launch of the dominant kernel of every bench leg, which bench.py quotes as roofline.traffic / .limiter."""
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
    "deflate-L6-silesia-mix-4gib-match": entry("r03_final_4gib_L6_summary.txt", "walk_kernel<true>",
        "vector-ALU issue and the latency of scattered S lines: walk_kernel<true> (parse-driven search + the rest of the parse); bodies 58 % / folds 14 % / passes 29 % of a wave's cycles "
        "(profiles/r03_walk_phases_levels_4_6_9.txt); not HBM bandwidth, but 50x the algorithmic bytes cross the fabric"),
    "deflate-L9-silesia-mix-4gib-match": entry("r03_L9_4gib_summary.txt", "walk_kernel<true>",
        "the same kernel with 4096-deep chains: bodies 67 % of a wave's cycles, 44 candidate steps per byte (profiles/r03_walk_phases_levels_4_6_9.txt)"),
    "deflate-L1-silesia-mix-4gib-lz_serial": entry("r03_L1_4gib_lz_serial_summary.txt", "lz_serial_kernel",
        "latency of the slowest chunks' chains of dependent scattered reads (head, candidate bytes, prev): one lane per chunk, a launch lasts as long as the chunk with the most "
        "tokens; the 5 % of chunks that do not compress go to the wave-per-chunk kernel after 4 KiB (377 -> 232 ms, profiles/r03_serial_loop_by_class.txt, r03_hand_on_levels_1_3.txt); "
        "that kernel, which serves launches below 1.25 GiB alone, is issue-bound instead (one wave per SIMD)"),
    "deflate-L1-silesia-mix-1gib-match": entry("r03_L1_fastwin_1gib_summary.txt", "fastwin_kernel<4, 8>",
        "instruction issue of a lone wave per SIMD (one instruction every 5-6 cycles), three chunks per CU (47 KiB of LDS each): 12 000 cycles per 64-position window, "
        "evaluation 31 % / scalar walk 49 % (profiles/r03_fastwin_phases.txt)"),
    "inflate-L6-silesia-mix-4gib-inflate": entry("r03_inflate_4gib_L6_summary.txt", "inflate_kernel_t<false>",
        "latency of the per-token chains of one reader and one writer wave per segment, four segments per CU (40 KiB of LDS each); traffic 1.0x the algorithmic bytes; not HBM.  "
        "With a 16 / 8 KiB ring (output void, timing only) the kernel takes 67 / 60 ms instead of 84: residency is worth at most 1.3-1.45x"),
}
json.dump(out, open(os.path.join(P, "r03_traffic.json"), "w"), indent=1)
for k, v in out.items():
    print("%-42s traffic %7.2f GB per launch (fetch %7.2f, write %6.2f)  %s" % (k, v["traffic_bytes_per_launch"] / 1e9, v["fetch_bytes_per_launch"] / 1e9, v["write_bytes_per_launch"] / 1e9, v.get("counters")))
