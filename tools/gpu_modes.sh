#!/bin/bash
# usage (GPU box): tools/gpu_modes.sh <tag>  -- kernel-traced bench of the SILK-NB and hybrid workloads (65,536 streams each);
# prints ms/step and the per-kernel averages.  The CELT workload is covered by tools/gpu_check.sh.
tag=${1:-modes}
export TMPDIR=/tmp
for wl in silk_nb_stereo_64k hybrid_fb_stereo_256k; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trm_${tag}_$wl -o t -- python3 bench.py --workload $wl --streams 65536 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bm_${tag}_$wl.log 2>&1 || { echo "$wl failed"; tail -5 gpurun_out/bm_${tag}_$wl.log; exit 1; }
  grep "^{" gpurun_out/bm_${tag}_$wl.log | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$wl: ms/step %.3f  frames/s %.0f' % (d['ms_per_step'], d['value']))"
  python3 - <<PY
import csv
for r in csv.DictReader(open('gpurun_out/trm_${tag}_$wl/t_kernel_stats.csv')):
    if float(r['AverageNs']) > 5e4: print('   %-16s calls %3s avg %.3f ms' % (r['Name'].split('(')[0], r['Calls'], float(r['AverageNs'])/1e6))
PY
done
