"""Import FIRST in a measurement script that sweeps a switch of the LAB build (SR_BAL_*, SR_ST1_PROBE, SR_T64_*, SR_FACT_*,
SR_APPEND_NO_*, variance-kernel variants other than 4): points the package at scripts/_bin/libsafereach_lab.so
(`make -C safe_exploration_amd/csrc lab`, also built by scripts/build_bins.sh and __graft_entry__.build()).  The product
library reads none of those switches."""
import os

LAB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_bin", "libsafereach_lab.so")
if not os.path.exists(LAB):
    raise SystemExit("lab build missing: make -C safe_exploration_amd/csrc lab")
os.environ.setdefault("SAFEREACH_LIB", LAB)
