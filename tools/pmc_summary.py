#!/usr/bin/env python3
"""Summarise a tools/prof_pmc.sh output directory: per-launch averages of every counter for k_decode_step,
plus per-frame figures (frames per launch = grid size / 64)."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "k_decode_step"
acc = defaultdict(list)
grid = None
for f in sorted(glob.glob(root + "/pmc*/**/*counter_collection.csv", recursive=True)):
    per_dispatch = defaultdict(float)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if kern not in row["Kernel_Name"]:
                continue
            grid = int(row["Grid_Size"])
            per_dispatch[(row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
    for (_, name), v in per_dispatch.items():
        acc[name].append(v)
frames = grid // 64 if grid else 1
print(f"kernel {kern}: grid {grid} -> {frames} frames per launch")
for f in sorted(glob.glob(root + "/trace/**/*kernel_stats.csv", recursive=True)):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if kern in row["Name"]:
                print(f"kernel-trace: calls {row['Calls']} avg {float(row['AverageNs'])/1e6:.3f} ms  min {float(row['MinNs'])/1e6:.3f}  max {float(row['MaxNs'])/1e6:.3f}")
for name in sorted(acc):
    v = sum(acc[name]) / len(acc[name])
    print(f"{name:28s} {v:16.4g} per launch   {v/frames:12.2f} per frame   (n={len(acc[name])})")
