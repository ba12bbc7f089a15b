#!/bin/bash
# GPU box: model updates of 3 .. 24 blocks of 128 rows -- the whole update on the CALLER's stream up to SR_FACT_ALL_ON_CALLER blocks (no fork
# to the priority stream, no side streams, no events; 0 = the forked form of rounds 2 - 5; the product's default is 15); lab build.
# bash scripts/r06_small_update_ab.sh > gpurun_out/small_update_ab.txt
for pass in 1 2; do
  for n in 0 4 6 8 10; do
    SR_FACT_ALL_ON_CALLER=$n timeout 600 python scripts/refit_ab.py 300,450,600,700,800,900,1000,1200 2>&1 | tail -n 1
  done
  for n in 0 12 14 16 20 24; do
    SR_FACT_ALL_ON_CALLER=$n timeout 600 python scripts/refit_ab.py 1200,1500,1800,2000,2500,3000 2>&1 | tail -n 1
  done
done
