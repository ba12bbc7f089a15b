#!/bin/bash
# GPU box: super-tiles in the job-table GEMM of the triangular inversion (SR_JOBS_ST_THR: tiles of a job's grid from which
# its workgroups are dealt to the XCDs in 8 x 8 super-tiles; 1000000000 = the plain grid of rounds 2 - 5), lab build.
# bash scripts/r06_jobs_st_ab.sh > gpurun_out/jobs_st_ab.txt
for pass in 1 2; do
  for thr in 1000000000 4096 1024 256; do
    SR_JOBS_ST_THR=$thr timeout 600 python scripts/refit_ab.py ${1:-10000,20000,50000} 2>&1 | tail -n 1
  done
done
