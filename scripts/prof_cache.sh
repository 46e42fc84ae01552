#!/bin/bash
# GPU box: HBM-side traffic and L2 hit rate per kernel for one bench configuration (separate rocprofv3 passes: the TCC block has
# 4 counter slots, FETCH_SIZE takes 3).  Usage: scripts/prof_cache.sh TAG [bench args]
TAG=$1; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_rd -- python3 bench.py "$@" --no-cpu-baseline --no-extras > $OUT/rd.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_wr -- python3 bench.py "$@" --no-cpu-baseline --no-extras > $OUT/wr.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- python3 bench.py "$@" --no-cpu-baseline --no-extras > $OUT/l2.log 2>&1 || exit 1
python3 - $OUT <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
for d in ("pmc_rd", "pmc_wr", "pmc_l2"):
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (out, d), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:48]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
with open(out + "/cache_summary.txt", "w") as fh:
    for k, v in sorted(acc.items()):
        n = max(len(calls[(k, c)]) for c in v)
        line = "%-50s dispatches %d  " % (k, n) + "  ".join("%s/dispatch %.4g" % (c, x / max(len(calls[(k, c)]), 1)) for c, x in sorted(v.items()))
        if "TCC_HIT_sum" in v: line += "  L2 hit rate %.1f%%" % (100 * v["TCC_HIT_sum"] / max(v["TCC_HIT_sum"] + v["TCC_MISS_sum"], 1))
        print(line); fh.write(line + "\n")
PY
