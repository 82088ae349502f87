#!/usr/bin/env python3
"""Copy the judged summaries of a gpurun profiling pass (gpurun_out/prof_<tag>) into profiles/ (tracked).
usage: python scripts/save_profile.py <tag> [round]"""
import csv
import glob
import os
import shutil
import sys

tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r01"
root = os.path.join("gpurun_out", f"prof_{tag}")
os.makedirs("profiles", exist_ok=True)
for f in glob.glob(os.path.join(root, "trace", "*kernel_stats.csv")):
    with open(f) as fh, open(os.path.join("profiles", f"{rnd}_{tag}_kernel_stats.csv"), "w") as out:
        w = csv.writer(out)
        for row in csv.reader(fh):
            row[0] = row[0][:120]
            w.writerow(row)
src = os.path.join(root, "pmc_summary.txt")
if os.path.exists(src):
    keep = [l for l in open(src).read().splitlines() if l.startswith("==") or "coma::" in l or "sd::" in l or "_ZN2sd" in l or "seg::" in l]
    open(os.path.join("profiles", f"{rnd}_{tag}_pmc.txt"), "w").write(
        "# rocprofv3 --pmc passes (one counter group per pass), mean per dispatch; FETCH_SIZE/WRITE_SIZE in KiB.\n"
        "# gfx950: FETCH_SIZE reads 1/2 of the bytes of a coalesced stream (MI355X_MICROARCH.md, HBM) -> double it.\n"
        + "\n".join(keep) + "\n")
j = os.path.join(root, "bench_line_under_profiler.json")
if os.path.exists(j):
    shutil.copy(j, os.path.join("profiles", f"{rnd}_{tag}_bench_line_under_profiler.json"))
t = os.path.join(root, "time_seg.log")
if os.path.exists(t):          # the per-launch HIP-event table of scripts/time_seg.py taken in the same call
    shutil.copy(t, os.path.join("profiles", f"{rnd}_{tag}_per_launch.txt"))
print(open(os.path.join("profiles", f"{rnd}_{tag}_kernel_stats.csv")).read()[:1500])
