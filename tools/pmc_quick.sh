#!/bin/bash
# usage (GPU box): tools/pmc_quick.sh lib1.so lib2.so ...   -- one --pmc pass of the in-order CELT bench per build ("default" = the
# in-tree library): per-frame instruction and cycle counters of k_celt_recon_fb and k_celt_parse (PMC_KERNELS to name others).
export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename $lib .so)
  if [ "$lib" = default ]; then unset OPUSGPU_LIB; else export OPUSGPU_LIB=$PWD/$lib; fi
  rm -rf gpurun_out/pq_$tag
  timeout -k 5 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pq_$tag -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --pipeline off ${PQ_ARGS} > gpurun_out/pq_$tag.log 2>&1
  python3 - <<PY
import csv, glob, os
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(set)
want = os.environ.get('PMC_KERNELS', 'k_celt_recon_fb,k_celt_parse').split(',')
for f in glob.glob('gpurun_out/pq_$tag/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].strip()
        if k not in want: continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
for k in acc:
    d = len(n[k]) * float(os.environ.get('PQ_FRAMES', '65536'))
    c = {a.replace('SQ_', ''): v / d for a, v in acc[k].items()}
    print('%-10s %-16s ' % ('$tag', k) + '  '.join('%s %.0f' % kv for kv in sorted(c.items())) +
          '  | lanes %.1f%%' % (100 * c.get('THREAD_CYCLES_VALU', 0) / max(1, 64 * c.get('ACTIVE_INST_VALU', 1))))
PY
done
