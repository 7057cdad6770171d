#!/usr/bin/env python3
"""Summarise a tools/prof_pmc.sh output directory: per-launch averages of every counter for the decode kernels,
plus per-frame figures.  usage: pmc_summary.py <dir> [frames_per_launch]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
kernels = ["k_celt_parse", "k_celt_parse64", "k_celt_recon_fb", "k_celt_recon", "k_celt_post", "k_decode_step", "k_silk_parse", "k_silk_parse64", "k_silk_params", "k_silk_synth", "k_silk_synth_nb"]
for f in sorted(glob.glob(root + "/trace/**/*kernel_stats.csv", recursive=True)):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            print(f"kernel-trace: {row['Name'].split('(')[0]:16s} calls {row['Calls']:>3s} avg {float(row['AverageNs'])/1e6:8.3f} ms"
                  f"  min {float(row['MinNs'])/1e6:8.3f}  max {float(row['MaxNs'])/1e6:8.3f}")
traffic = {}
durations = {}
steps = int(os.environ.get("PROF_STEPS", "4"))  # steps of the traced run (bench --steps 3 --warmup 1)
for f in sorted(glob.glob(root + "/trace/**/*kernel_stats.csv", recursive=True)):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            # per STEP: a step with SILK frames launches its kernels twice (two halves on two streams, og_api.hip)
            durations[row["Name"].split("(")[0]] = float(row["TotalDurationNs"]) / steps / 1e6
for kern in kernels:
    acc = defaultdict(list)
    for f in sorted(glob.glob(root + "/pmc*/**/*counter_collection.csv", recursive=True)):
        per_dispatch = defaultdict(float)
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Kernel_Name"].split("(")[0].strip() != kern:
                    continue
                per_dispatch[(row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
        for (_, name), v in per_dispatch.items():
            acc[name].append(v)
    if not acc:
        continue
    if "FETCH_SIZE" in acc and "WRITE_SIZE" in acc:
        fetch = sum(acc["FETCH_SIZE"]) / len(acc["FETCH_SIZE"])
        write = sum(acc["WRITE_SIZE"]) / len(acc["WRITE_SIZE"])
        # rocprofv3 reports KiB; on gfx950 FETCH_SIZE reads half of what a wide stream fetches (MI355X_MICROARCH.md, HBM)
        traffic[kern] = {"fetch_size_kib": fetch, "write_size_kib": write, "hbm_bytes_per_launch": (2 * fetch + write) * 1024,
                         "avg_ms": durations.get(kern)}
        if "SQ_INSTS_VALU" in acc:  # vector-ALU wave-instructions per launch: what bounds the kernels that are not HBM-bound
            traffic[kern]["valu_insts_per_launch"] = sum(acc["SQ_INSTS_VALU"]) / len(acc["SQ_INSTS_VALU"])
    print(f"== {kern} ({frames} frames per launch)")
    for name in sorted(acc):
        v = sum(acc[name]) / len(acc[name])
        print(f"{name:28s} {v:16.4g} per launch   {v/frames:12.2f} per frame   (n={len(acc[name])})")

if traffic:
    out = {"frames_per_launch": frames, "kernels": traffic,
           "hbm_bytes_per_step": sum(k["hbm_bytes_per_launch"] for k in traffic.values()),
           "note": "counters per launch = per step (in-order steps, one launch per kernel and step), (2 x FETCH_SIZE + WRITE_SIZE) x 1024: "
                   "separate --pmc passes, gfx950 FETCH_SIZE correction; avg_ms = the kernel's total time per step in the kernel-trace "
                   "pass, which runs the bench as it is (pipelined / windowed / in halves: next to its neighbours)"}
    with open(root + "/traffic.json", "w") as fh:
        json.dump(out, fh, indent=1)
    print("traffic:", json.dumps(out))
