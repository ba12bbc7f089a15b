#!/bin/bash
# round 6: partial products of the streamed MFMA kernels stored as 8-byte pieces (SR_ST_EPI=0, round 5), as one 32-byte run per lane (1), non-temporal (2, default)
OUT=gpurun_out/r06; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fuzz or small_batch_routes or streamed" 2>&1 | tail -2
for v in 0 1 2 0 1 2; do SR_ST_EPI=$v timeout 300 python scripts/bal_ab.py 1500,3000,4000,5000,8000 8,16,32,64 | sed "s/^SR_BAL_WGS=- SR_BAL_THR=- /epi=$v /"; done | tee $OUT/epi_ab.txt
