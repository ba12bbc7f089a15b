#!/bin/bash
# GPU box: model update with its chain on the CALLER's stream up to SR_FACT_CHAIN_ON_CALLER blocks of 128 rows (no fork to the priority
# stream and no join back; the trailing updates and the inversion's stage keep their side streams), lab build.
# bash scripts/r06_chain_on_caller_ab.sh > gpurun_out/chain_on_caller_ab.txt
SIZES=${1:-300,500,800,1000,1500,2000,3000,5000}
for pass in 1 2; do
  for nb in 0 8 16 24 40; do
    SR_FACT_CHAIN_ON_CALLER=$nb timeout 600 python scripts/refit_ab.py $SIZES 2>&1 | tail -n 1
  done
done
