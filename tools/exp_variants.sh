#!/bin/bash
# usage (GPU box): tools/exp_variants.sh lib1.so lib2.so ...  -- for each experimental build of the library: the GPU parity
# suite, then kernel-traced benches of the CELT, SILK-NB and hybrid workloads at 65,536 streams (per-kernel averages).
# "default" = the in-tree library.
export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename $lib .so)
  if [ "$lib" = default ]; then unset OPUSGPU_LIB; else export OPUSGPU_LIB=$PWD/$lib; fi
  echo "== $tag"
  timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/xt_$tag.log 2>&1; tail -1 gpurun_out/xt_$tag.log
  for wl in celt_fb_stereo_64k silk_nb_stereo_64k hybrid_fb_stereo_256k; do
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/xtr_${tag}_$wl -o t -- python3 bench.py --workload $wl --streams 65536 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/xb_${tag}_$wl.log 2>&1 || { echo "bench failed"; tail -5 gpurun_out/xb_${tag}_$wl.log; exit 1; }
    python3 - <<PY
import csv, json
line = ''
for l in open('gpurun_out/xb_${tag}_$wl.log'):
    if l.startswith('{'):
        d = json.loads(l); line = '   %-22s %.3f ms/step |' % ('$wl', d['ms_per_step'])
for r in csv.DictReader(open('gpurun_out/xtr_${tag}_$wl/t_kernel_stats.csv')):
    if float(r['AverageNs']) > 5e4 and 'init' not in r['Name']: line += ' %s %.3f' % (r['Name'].split('(')[0][2:], float(r['AverageNs'])/1e6)
print(line)
PY
  done
done
