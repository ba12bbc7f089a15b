#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel of the library (hipcc -Rpass-analysis=kernel-resource-usage, gfx950).
Runs without a GPU:  python scripts/kernel_resources.py > profiles/rNN_kernel_resources.txt"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "safe_exploration_amd", "csrc")


def demangle(names):
    for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
        try:
            out = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True)
            if out.returncode == 0:
                return out.stdout.splitlines()
        except OSError:
            pass
    return names


def compile_rows(src):
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", src, "-o", "/dev/null",
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    rows, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|"
                      r"LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"file": os.path.basename(src), "name": v}
            rows.append(cur)
        elif cur is not None:
            cur["Spill" if k == "VGPRs Spill" else k.split(" ")[0]] = v
    return rows


def collect(jobs=None):
    """Rows (dicts) of every kernel of the library, sources compiled `jobs` at a time; names demangled, arguments cut."""
    from concurrent.futures import ThreadPoolExecutor
    srcs = [s for s in sorted(glob.glob(os.path.join(CSRC, "*.hip"))) if os.path.basename(s) != "sr_comm.hip"]
    with ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as ex:
        rows = [r for part in ex.map(compile_rows, srcs) for r in part]
    for r, n in zip(rows, demangle([r["name"] for r in rows])):
        r["kernel"] = re.sub(r"\(.*", "", n)
    return rows


def main():
    rows = collect()
    print("# hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage over safe_exploration_amd/csrc/*.hip")
    print("# %-16s %5s %5s %6s %8s %4s %7s  %s" % ("file", "VGPR", "AGPR", "spills", "scratchB", "occ", "LDS B", "kernel"))
    spill = 0
    for r in rows:
        n = r["kernel"]
        sc = int(r.get("ScratchSize", 0))
        spill += sc > 0
        print("%-18s %5s %5s %6s %8d %4s %7s  %s" % (r["file"], r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("Spill", "?"), sc,
                                                  r.get("Occupancy", "?"), r.get("LDS", "?"), n))
    print("# %d kernels, %d with scratch" % (len(rows), spill))


if __name__ == "__main__":
    sys.exit(main())
