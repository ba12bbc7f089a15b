#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root:  bash scripts/profile_gpu.sh r01
# Produces gpurun_out/prof_<tag>/...: kernel-trace stats of the default bench workload and separate
# PMC passes (never combined with other trace domains).  Summaries are copied to profiles/ by hand.
set -u
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="python $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench -- $BENCH > "$OUT/bench_trace.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  N=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --kernel-trace -d "$OUT/pmc_$N" -o bench -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/pmc_$N.log" 2>&1
done
cd "$REPO"
find "$OUT" -name "*.csv" | head -50
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)):
    print("==", f)
    print(open(f).read()[:3000])
for d in sorted(glob.glob(out + "/pmc_*")):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            k = (row.get("Kernel_Name", "")[:40], row.get("Counter_Name", ""))
            agg[k][0] += float(row.get("Counter_Value", 0) or 0)
            agg[k][1] += 1
        print("==", f)
        for (kn, cn), (v, n) in sorted(agg.items()):
            print("%-42s %-28s total=%.6g  rows=%d  per_row=%.6g" % (kn, cn, v, n, v / max(n, 1)))
PY
