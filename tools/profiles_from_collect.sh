#!/bin/bash
# usage (container, after both parts of tools/collect_profiles.sh came back): tools/profiles_from_collect.sh <round dir, e.g. profiles/r04> <prefix letter>
# copies what gpurun_out/collect/ holds into the round's directory under the names bench.py and DESIGN.md use
dst=$1; pre=${2:-n}; src=gpurun_out/collect
mkdir -p $dst
for w in celt:celt_fb_stereo_64k silk_nb:silk_nb_stereo_64k hybrid:hybrid_fb_stereo_256k mixed:mixed_pages_2m; do
  s=${w%%:*}; n=${w##*:}
  [ -f $src/$s/summary.txt ] || continue
  grep -v "^traffic:" $src/$s/summary.txt > $dst/k_${s}_pmc_summary.txt
  grep "^traffic:" $src/$s/summary.txt | sed 's/^traffic: //' | python3 -c "
import json, sys
d = json.loads(sys.stdin.read()); d['measured_on'] = '$dst (tools/collect_profiles.sh pmc)'; print(json.dumps(d, indent=1))" > $dst/traffic_$n.json
  f=$(ls $src/$s/trace/*/*kernel_stats.csv $src/$s/trace/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp $f $dst/k_${n}_kernel_stats.csv
done
[ -f $src/bench_default.json ] && cp $src/bench_default.json $dst/${pre}_bench_default_all_configs.json
[ -f $src/pytest_gpu.log ] && cp $src/pytest_gpu.log $dst/${pre}_pytest_gpu.log
[ -f $src/host_path_rate.txt ] && cp $src/host_path_rate.txt $dst/${pre}_host_path_rate.txt
[ -f $src/host_path_timeline.txt ] && cp $src/host_path_timeline.txt $dst/${pre}_host_path_timeline.txt
[ -f $src/launch_jitter.txt ] && cp $src/launch_jitter.txt $dst/${pre}_launch_jitter_pipelined_steps.txt
[ -f $src/soaks.log ] && cp $src/soaks.log $dst/${pre}_soaks.log
ls $dst
