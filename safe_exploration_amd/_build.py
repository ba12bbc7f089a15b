"""Build libsafereach.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
# SAFEREACH_LIB (measurement scripts only): another build of the same C-ABI, e.g. scripts/_bin/libsafereach_lab.so (`make lab`:
# the A/B kernel variants and the environment switches of the sweeps compiled in)
LIB_PATH = os.environ.get("SAFEREACH_LIB") or os.path.join(PKG_DIR, "libsafereach.so")


def build(force=False, verbose=False):
    """Run ``make`` in csrc/ (incremental).  Raises RuntimeError with the compiler output on failure."""
    cmd = ["make", "-C", CSRC, "-j4"]
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=False, capture_output=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose:
        print(res.stdout)
    if res.returncode != 0 or not os.path.exists(LIB_PATH):
        raise RuntimeError("building libsafereach.so failed:\n" + res.stdout + "\n" + res.stderr)
    return LIB_PATH


if __name__ == "__main__":
    print(build(verbose=True))
