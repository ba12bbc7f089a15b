#!/bin/bash
# PMC passes of an arbitrary command, one rocprofv3 run per counter set (never combined with other trace domains), and a
# per-kernel table of the counters.  ON THE GPU BOX, from the repo root:
#   bash scripts/pmc_any.sh gpurun_out/pmc_t64 "python scripts/latency_trace.py run 5000 64" "SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_MFMA ..."
set -u
OUT=$(pwd)/$1; CMD=$2; shift 2
REPO=$(pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
for C in "$@"; do
  i=$((i + 1))
  ( cd /tmp && cd "$REPO" && timeout 300 rocprofv3 --pmc $C --kernel-trace -d "$OUT/pass$i" -o pmc -- $CMD > "$OUT/pass$i.log" 2>&1 )
done
python - "$OUT" <<'PY'
import glob, os, sqlite3, sys
for d in sorted(glob.glob(sys.argv[1] + "/pass*")):
    if not os.path.isdir(d):
        continue
    for db in glob.glob(d + "/**/*.db", recursive=True):
        q = ("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name "
             "order by kernel_name, counter_name")
        for kn, cn, n, avg in sqlite3.connect(db).execute(q):
            if kn.startswith(("sr_", "void sr_")):
                print("%-44s %-30s launches %5d  per launch %16.6g" % (kn[:44], cn, n, avg))
PY
