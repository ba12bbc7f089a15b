#!/bin/bash
# GPU box: model updates of 3 .. 10 blocks of 128 rows -- the whole update on the CALLER's stream up to SR_FACT_ALL_ON_CALLER blocks (no fork
# to the priority stream, no side streams, no events), or only the chain (SR_FACT_CHAIN_ON_CALLER; trailing updates and the inversion's
# stage keep their side streams); lab build.   bash scripts/r06_small_update_ab.sh > gpurun_out/small_update_ab.txt
SIZES=${1:-300,450,600,700,800,900,1000,1200}
for pass in 1 2; do
  for cfg in "0 0" "6 0" "0 4" "0 6" "0 8" "0 10"; do
    set -- $cfg
    SR_FACT_CHAIN_ON_CALLER=$1 SR_FACT_ALL_ON_CALLER=$2 timeout 600 python scripts/refit_ab.py $SIZES 2>&1 | tail -n 1
  done
done
