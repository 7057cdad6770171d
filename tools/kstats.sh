#!/bin/bash
# usage (GPU box): tools/kstats.sh lib1.so lib2.so ...  -- kernel-traced CELT bench (10 steps) for each experimental build; prints
# the step time and the per-kernel averages above 50 us.  "default" = the in-tree library.
export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename $lib .so)
  if [ "$lib" = default ]; then unset OPUSGPU_LIB; else export OPUSGPU_LIB=$PWD/$lib; fi
  rm -rf gpurun_out/ks_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_$tag -o t -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs ${KS_ARGS} > gpurun_out/ks_$tag.log 2>&1 || { echo "$tag: bench failed"; tail -3 gpurun_out/ks_$tag.log; continue; }
  python3 - <<PY
import csv, json, glob
line = '$tag: '
for l in open('gpurun_out/ks_$tag.log'):
    if l.startswith('{'):
        d = json.loads(l); line += '%.3f ms/step |' % d['ms_per_step']
for f in glob.glob('gpurun_out/ks_$tag/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r['AverageNs']) > 1e4 and 'init' not in r['Name']: line += ' %s %.3f' % (r['Name'].split('(')[0][2:], float(r['AverageNs'])/1e6)
print(line)
PY
done
