#!/usr/bin/env python3
"""Summarise a tools/prof_pmc.sh output directory: per-launch averages of every counter for the decode kernels,
plus per-frame figures.  usage: pmc_summary.py <dir> [frames_per_launch]"""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
kernels = ["k_celt_parse", "k_celt_recon", "k_decode_step"]
for f in sorted(glob.glob(root + "/trace/**/*kernel_stats.csv", recursive=True)):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            print(f"kernel-trace: {row['Name'].split('(')[0]:16s} calls {row['Calls']:>3s} avg {float(row['AverageNs'])/1e6:8.3f} ms"
                  f"  min {float(row['MinNs'])/1e6:8.3f}  max {float(row['MaxNs'])/1e6:8.3f}")
for kern in kernels:
    acc = defaultdict(list)
    for f in sorted(glob.glob(root + "/pmc*/**/*counter_collection.csv", recursive=True)):
        per_dispatch = defaultdict(float)
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if kern not in row["Kernel_Name"]:
                    continue
                per_dispatch[(row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
        for (_, name), v in per_dispatch.items():
            acc[name].append(v)
    if not acc:
        continue
    print(f"== {kern} ({frames} frames per launch)")
    for name in sorted(acc):
        v = sum(acc[name]) / len(acc[name])
        print(f"{name:28s} {v:16.4g} per launch   {v/frames:12.2f} per frame   (n={len(acc[name])})")
