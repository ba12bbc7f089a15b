#!/bin/bash
# GPU box: FETCH_SIZE and speed of sr_var_kernel (variant given) for several scheduling-group sizes.
set -u
REPO=$(pwd)
export TMPDIR=/tmp
V=${1:-2}
for G in 8 16 32 64 128; do
  OUT=$REPO/gpurun_out/pmc_grp$G
  rm -rf "$OUT"
  ( cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT" -o b -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --var-variant $V --var-group $G > /dev/null 2>&1 )
  python - "$OUT" $G <<'PY'
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
for kn, cn, n, avg in con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
    if "sr_var_kernel" in kn:
        print("group", sys.argv[2], cn, "KiB per launch %.5g" % avg, flush=True)
PY
  python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --var-variant $V --var-group $G 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('group', $G, 'TF', round(d['roofline']['achieved'],2))"
done
