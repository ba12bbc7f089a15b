#!/bin/bash
# ON THE GPU BOX: bash scripts/pmc_latency.sh N T   -- PMC passes (kernel-trace only, one counter group per pass) over
# scripts/latency_trace.py run N T; prints per-launch means for the kernels of the variance contraction.
set -u
N=$1; T=$2
REPO=$(pwd); export TMPDIR=/tmp
i=0
for CTRS in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"; do
  i=$((i+1)); OUT=$REPO/gpurun_out/pmc_lat_${N}_${T}_$i; rm -rf "$OUT"; mkdir -p "$OUT"
  ( cd /tmp && timeout 120 rocprofv3 --pmc $CTRS --kernel-trace -d "$OUT" -o lat -- python $REPO/scripts/latency_trace.py run $N $T > "$OUT.log" 2>&1 )
  python - "$OUT" <<'PY'
import glob, sqlite3, sys
for f in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    q = ("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
         "where kernel_name like '%sr_var%' or kernel_name like '%sr_stream%' group by kernel_name, counter_name")
    try:
        for kn, cn, n, avg in db.execute(q):
            print("%-28s %-32s n=%d per_launch=%.6g" % (kn.split("(")[0][-28:], cn, n, avg))
    except Exception as e:
        print("no counters:", e)
PY
done
