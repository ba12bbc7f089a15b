#!/bin/bash
# round 6: pipelined model update against the plain chain (GPU box, from the repo root)
set -u
OUT=gpurun_out/r06
mkdir -p $OUT
REPO=$(pwd)
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pipelined_model_update or potrf or gemm_tn" > $OUT/pipe_test.txt 2>&1
tail -5 $OUT/pipe_test.txt
SR_PANELS=0 SR_PIPE=${PIPES:-2,1,0} timeout 600 python scripts/factor_bench.py ${SIZES:-1000 2000 5000 10000} > $OUT/factor_bench_pipe.txt 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r06/factor_bench_pipe.txt'):
    if l.startswith('{'):
        d = json.loads(l); print(d['N'], 'pipe', d['pipelined'], d['refit_ms'], 'ms', d['TFLOPs'], 'TF', d['max|mu+s2n*alpha-y|'])
    else:
        print(l.rstrip())
PY
export TMPDIR=/tmp
rm -rf $REPO/gpurun_out/prof_r06/tl5000
( cd /tmp && SR_PIPE=${TLPIPE:-2} timeout 300 rocprofv3 --kernel-trace -d "$REPO/gpurun_out/prof_r06/tl5000" -o tl -- python "$REPO/scripts/timeline.py" run 5000 > "$REPO/$OUT/.tl.log" 2>&1 )
python scripts/timeline.py show "$REPO/gpurun_out/prof_r06/tl5000" 600 > $OUT/timeline5000_pipe.txt 2>&1
head -3 $OUT/timeline5000_pipe.txt
