#!/bin/bash
# usage (GPU box): tools/pmc_lds_by_phase.sh lib1.so lib2.so ...   -- LDS counters of k_celt_recon_fb for ablated builds
# (tools/build_variant.sh <name> -DOG_RABL=n: the kernel returns after stage n -- 1 staging, 2 leaf pass, 3 band loop; "default" =
# the whole kernel).  The differences between consecutive builds attribute bank conflicts to the phases.  The ablated builds
# produce wrong PCM on purpose: the bench's parity check fails after the counters are in.
export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename $lib .so)
  if [ "$lib" = default ]; then unset OPUSGPU_LIB; else export OPUSGPU_LIB=$PWD/$lib; fi
  rm -rf gpurun_out/lds_$tag
  timeout -k 5 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/lds_$tag -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > gpurun_out/lds_$tag.log 2>&1
  python3 - <<PY
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(set)
for f in glob.glob('gpurun_out/lds_$tag/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].strip()
        if k != 'k_celt_recon_fb': continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
for k in acc:
    d = len(n[k]) * 65536.0
    print('%-8s %s: ' % ('$tag', k) + '  '.join('%s %.0f' % (c.replace('SQ_', ''), v / d) for c, v in sorted(acc[k].items())) + '  (per frame)')
PY
done
