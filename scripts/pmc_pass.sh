#!/bin/bash
# ON THE GPU BOX: bash scripts/pmc_pass.sh <tag> "<COUNTER ...>" [bench args...]
# one rocprofv3 --pmc pass (kernel-trace only) over bench.py, prints per-launch counter means of our kernels
set -u
TAG=$1; CTRS=$2; shift 2
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc $CTRS --kernel-trace -d "$OUT" -o bench -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > "$OUT.log" 2>&1
cd "$REPO"
python - "$OUT" <<'PY'
import glob, sqlite3, sys
for f in glob.glob(sys.argv[1] + "/*.db"):
    db = sqlite3.connect(f)
    q = ("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
         "where kernel_name like '%sr_var%' or kernel_name like '%sr_kstar%' group by kernel_name, counter_name")
    for kn, cn, n, avg in db.execute(q):
        print("%-28s %-28s n=%d per_launch=%.6g" % (kn.split("(")[0][-28:], cn, n, avg))
PY
