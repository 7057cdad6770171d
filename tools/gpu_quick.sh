#!/bin/bash
# usage (GPU box): tools/gpu_quick.sh <tag> [pytest selection...]  -- a subset of the GPU parity tests (default: the CELT
# ones), then a kernel-traced bench of the headline workload; prints per-kernel averages.  Output under gpurun_out/<tag>/
tag=${1:-quick}; shift
sel=${*:-tests/test_gpu_celt.py tests/test_gpu_fullsize.py::test_fullsize_celt tests/test_gpu_modes.py tests/test_golden.py}
mkdir -p gpurun_out/$tag
timeout -k 10 600 python -m pytest $sel -x -q -m gpu > gpurun_out/$tag/tests.log 2>&1; rc=$?; tail -4 gpurun_out/$tag/tests.log
[ $rc -ne 0 ] && exit $rc
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/trace -o t -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs > gpurun_out/$tag/bench.log 2>&1 || { tail -5 gpurun_out/$tag/bench.log; exit 1; }
grep "^{" gpurun_out/$tag/bench.log | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('bench: ms/step %.3f  frames/s %.0f  roofline.frac %.4f' % (d['ms_per_step'], d['value'], d['roofline']['frac']))"
python3 - <<PY
import csv, glob
for f in glob.glob('gpurun_out/$tag/trace/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        print('%-16s calls %3s avg %.3f ms' % (r['Name'].split('(')[0], r['Calls'], float(r['AverageNs'])/1e6))
PY
