#!/bin/bash
# usage (GPU box): tools/pmc_icache.sh   -- one --pmc pass of the in-order CELT bench: instruction-cache requests / hits / misses and
# instruction fetches per frame and kernel (OPUSGPU_PARSE_WIDE=2 for the 64-frame parse kernel).  Round 4: misses are zero for every kernel.
export TMPDIR=/tmp
rm -rf gpurun_out/pq_ic
timeout -k 5 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_IFETCH SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pq_ic -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --pipeline off > gpurun_out/pq_ic.log 2>&1
python3 - <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(set)
for f in glob.glob('gpurun_out/pq_ic/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].strip()
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
for k in acc:
    d = len(n[k]) * 65536.0
    print('%-18s' % k, '  '.join('%s %.1f' % (a, v / d) for a, v in sorted(acc[k].items())))
PY
