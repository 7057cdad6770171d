#!/bin/bash
# usage (GPU box): tools/host_path_timeline.sh <tag> [host_path_rate args...]   -- kernel + memory-copy trace of build_ab/host_path_rate
# (build it first, see its header) and the device timeline of the LAST opusgpu_decode_packets call: uploads, kernels, PCM pieces.
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p $out
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out -o t -- build_ab/host_path_rate "$@" > $out/run.log 2>&1 || { tail -5 $out/run.log; exit 1; }
tail -1 $out/run.log
python3 - <<PY
import csv
ev = []
for r in csv.DictReader(open('$out/t_kernel_trace.csv')):
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0]))
for r in csv.DictReader(open('$out/t_memory_copy_trace.csv')):
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Direction'].replace('MEMORY_COPY_', '').lower()))
ev.sort()
# the last call: everything behind the last gap of more than 0.2 ms in front of a host-to-device copy
start = 0
for i in range(1, len(ev)):
    if ev[i][2] == 'host_to_device' and ev[i][0] - max(e[1] for e in ev[:i]) > 200_000:
        start = i
last = ev[start:]
t0 = last[0][0]
print('device timeline of the last call (ms from its first upload): start -> end (duration)')
for a, b, nm in last:
    if b - a > 20_000 or nm.startswith('k_'):
        print(f'  {(a - t0) / 1e6:7.3f} -> {(b - t0) / 1e6:7.3f}  ({(b - a) / 1e6:6.3f})  {nm}')
print(f'first upload to last PCM piece: {(max(e[1] for e in last) - t0) / 1e6:.3f} ms')
PY
