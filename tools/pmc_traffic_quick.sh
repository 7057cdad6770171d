#!/bin/bash
# usage (GPU box): tools/pmc_traffic_quick.sh build_ab/lib_a.so build_ab/lib_b.so ...   -- HBM bytes per frame and CELT kernel of library
# variants, two --pmc passes each (FETCH_SIZE, WRITE_SIZE; (2 x FETCH_SIZE + WRITE_SIZE) x 1024 as in tools/prof_pmc.sh), in-order steps of the
# headline workload with the 64-frame parse kernel: the quick form of collect_profiles.sh pmc for same-box comparisons.
export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename $lib .so)
  export OPUSGPU_LIB=$PWD/$lib
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/tr_${tag}_$c
    OPUSGPU_PARSE_WIDE=2 timeout -k 5 200 rocprofv3 --pmc $c --output-format csv -d gpurun_out/tr_${tag}_$c -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --pipeline off > gpurun_out/tr_$tag.log 2>&1
  done
  python3 - <<PY
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(set)
for f in glob.glob('gpurun_out/tr_${tag}_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].strip()
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])].add(r['Dispatch_Id'])
tot = 0
for k in acc:
    if not k.startswith('k_celt'): continue
    f = 2 * acc[k]['FETCH_SIZE'] * 1024 / (len(n[(k, 'FETCH_SIZE')]) * 65536.0)
    w = acc[k]['WRITE_SIZE'] * 1024 / (len(n[(k, 'WRITE_SIZE')]) * 65536.0)
    tot += f + w
    print('%-10s %-16s fetch %.0f B  write %.0f B per frame' % ('$tag', k, f, w))
print('%-10s total %.0f B per frame = %.2f x 21,633' % ('$tag', tot, tot / 21633))
PY
done
