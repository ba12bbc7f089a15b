#!/bin/bash
# GPU box: inversion tree of the chain-bound model update -- SR_INV_SPINE = levels of the left spine that split late (1 = halving
# everywhere, rounds 2 - 5; 3 = [0, nb) at 7 nb / 8, then 3 nb / 4, then nb / 2); SR_INV_EVERY = e: a stage of the inversion at every
# panel boundary from nb / e on (0: at the spine's split points only).  Lab build.
# bash scripts/r06_inv_spine_ab.sh > gpurun_out/inv_spine_ab.txt
SIZES=${1:-3300,4000,5000,6500,8000,10000}
for pass in 1 2; do
  for cfg in "1 0" "3 0" "2 0" "1 4" "2 4" "3 4" "2 2"; do
    set -- $cfg
    SR_INV_SPINE=$1 SR_INV_EVERY=$2 timeout 600 python scripts/refit_ab.py $SIZES 2>&1 | tail -n 1
  done
done
