#!/bin/bash
# usage (GPU box): tools/ab.sh <tag> "<lib names in build_ab/ (without lib_ / .so)>" "<workloads>" [steps]  -- same-box A/B of library
# variants: every variant x workload once through bench.py (no CPU baseline, no other configs), two rounds interleaved so that
# drift of the box shows; prints ms/step.  Output under gpurun_out/<tag>/.
tag=$1; libs=$2; wls=$3; steps=${4:-20}
mkdir -p gpurun_out/$tag
for round in 1 2; do
  for wl in $wls; do
    for l in $libs; do
      OPUSGPU_LIB=$PWD/build_ab/lib_$l.so timeout -k 10 200 python3 bench.py --workload $wl --steps $steps --warmup 3 --no-cpu-baseline --no-other-configs \
        > gpurun_out/$tag/b_${l}_${wl}_$round.json 2> gpurun_out/$tag/b_${l}_${wl}_$round.err || { echo "FAILED $l $wl"; tail -3 gpurun_out/$tag/b_${l}_${wl}_$round.err; exit 1; }
      python3 - "$l" "$wl" "$round" gpurun_out/$tag/b_${l}_${wl}_$round.json <<'PY'
import json, sys
for line in open(sys.argv[4]):
    if line.startswith("{"):
        d = json.loads(line)
        print("%-10s %-24s round %s: %.3f ms/step  %.2f M frames/s  check: %s" % (sys.argv[1], sys.argv[2], sys.argv[3], d["ms_per_step"], d["value"] / 1e6,
              str(d.get("parity_check", {}).get("result", "?"))[:40]))
PY
    done
  done
done
