#!/bin/bash
# GPU box: resident diagonal-block kernel of the GEMM-bound model update (regime 2) against launched diagonal blocks
# (SR_FACT_NO_SERVER=1), and the share of trailing updates that take the unmasked stream (SR_FACT_FREE_RATIO); lab build.
# bash scripts/r06_diag_server_ab.sh > gpurun_out/diag_server_ab.txt
SIZES=${1:-17000,30000,50000}
for pass in 1 2; do
  SR_FACT_NO_SERVER=1 timeout 600 python scripts/refit_ab.py $SIZES 2>&1 | tail -n 1
  timeout 600 python scripts/refit_ab.py $SIZES 2>&1 | tail -n 1
  for ratio in 0.5 0.0001; do
    SR_FACT_FREE_RATIO=$ratio timeout 600 python scripts/refit_ab.py $SIZES 2>&1 | tail -n 1
  done
done
for thr in 16384 1000000000; do
  SR_JOBS_ST_THR=$thr timeout 600 python scripts/refit_ab.py 30000,50000 2>&1 | tail -n 1
done
