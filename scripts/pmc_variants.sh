#!/bin/bash
# GPU box: FETCH_SIZE of sr_var_kernel for the main-loop variants 1 and 2 in the same session (fabric-side traffic A/B).
set -u
REPO=$(pwd)
export TMPDIR=/tmp
for V in 1 2 1 2; do
  OUT=$REPO/gpurun_out/pmc_var$V
  rm -rf "$OUT"
  ( cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT" -o b -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --var-variant $V > /dev/null 2>&1 )
  python - "$OUT" $V <<'PY'
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
q = ("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
     "group by kernel_name, counter_name")
for kn, cn, n, avg in con.execute(q):
    if "sr_var_kernel" in kn:
        print("variant", sys.argv[2], kn[:24], cn, "launches", n, "KiB per launch %.5g" % avg, flush=True)
PY
done
