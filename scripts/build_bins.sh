#!/bin/bash
# Builds the stand-alone microbenchmarks into scripts/_bin/ (git-ignored; they travel to the GPU box with the snapshot).
# No GPU needed:  bash scripts/build_bins.sh
set -e
cd "$(dirname "$0")/.."
mkdir -p scripts/_bin
F="--offload-arch=gfx950 -O3 -std=c++17"
hipcc $F scripts/mfma_pipe_tile.hip -o scripts/_bin/pipe
hipcc $F scripts/mfma_issue.hip -o scripts/_bin/mfma_issue
hipcc $F -I safe_exploration_amd/csrc scripts/pivot_chain.hip -o scripts/_bin/pivot
hipcc $F scripts/xqueue_handoff.hip -o scripts/_bin/xq
make -C safe_exploration_amd/csrc -j8 lab
ls -la scripts/_bin
