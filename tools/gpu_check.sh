#!/bin/bash
# usage (GPU box): tools/gpu_check.sh <tag>  -- GPU parity tests, then a kernel-traced bench; prints per-kernel averages
tag=${1:-chk}
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/t_$tag.log 2>&1; tail -3 gpurun_out/t_$tag.log
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tr_$tag -o t -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs > gpurun_out/b_$tag.log 2>&1
grep "^{" gpurun_out/b_$tag.log | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('bench: ms/step %.3f  frames/s %.0f  roofline.frac %.4f' % (d['ms_per_step'], d['value'], d['roofline']['frac']))"
python3 - <<PY
import csv
for r in csv.DictReader(open('gpurun_out/tr_$tag/t_kernel_stats.csv')):
    print('%-16s calls %3s avg %.3f ms' % (r['Name'].split('(')[0], r['Calls'], float(r['AverageNs'])/1e6))
PY
