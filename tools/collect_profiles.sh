#!/bin/bash
# usage (GPU box, from the repo root): tools/collect_profiles.sh <part>   -- what profiles/rNN/ of a round is made from; output under
# gpurun_out/collect/.  part = pmc: kernel-trace + --pmc passes of the four BASELINE workloads (tools/prof_pmc.sh: summary.txt and
# traffic.json each);  part = rest: the default bench line, the GPU test log, host-path rates and timeline, launch-jitter table,
# parity soaks.  Two gpurun calls: bench.py reads the traffic files of the first part from profiles/ in the second.
part=$1
out=gpurun_out/collect
mkdir -p $out
if [ "$part" = pmc ]; then
  PROF_PARSE_WIDE=2 tools/prof_pmc.sh collect/celt > $out/celt.log 2>&1 || { tail -5 $out/celt.log; exit 1; }
  PROF_PARSE_WIDE=2 tools/prof_pmc.sh collect/silk_nb --workload silk_nb_stereo_64k > $out/silk_nb.log 2>&1 || { tail -5 $out/silk_nb.log; exit 1; }
  PROF_PARSE_WIDE=2 PROF_FRAMES=262144 PROF_PASS_TIMEOUT=240 tools/prof_pmc.sh collect/hybrid --workload hybrid_fb_stereo_256k > $out/hybrid.log 2>&1 || { tail -5 $out/hybrid.log; exit 1; }
  PROF_PARSE_WIDE=2 PROF_FRAMES=262144 PROF_PASS_TIMEOUT=240 tools/prof_pmc.sh collect/mixed --workload mixed_pages_2m > $out/mixed.log 2>&1 || { tail -5 $out/mixed.log; exit 1; }
  rm -rf $out/*/pmc*/ $out/*/trace/*/*_trace.csv   # (keep the summaries and kernel stats; the raw counter files are large)
  ls $out
else
  timeout -k 10 900 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; tail -2 $out/pytest_gpu.log
  timeout -k 10 400 python bench.py --verbose > $out/bench_default.json 2> $out/bench_default.err || tail -3 $out/bench_default.err
  mkdir -p build_ab; g++ -O2 -std=c++17 -pthread tools/host_path_rate.cpp -Iinclude -Lesp32-opus-player_amd -lopusgpu -Wl,-rpath,'$ORIGIN/../esp32-opus-player_amd' -o build_ab/host_path_rate
  { for cfg in "0 8 1" "1 8 1" "0 8 1" "1 8 1" "1 2 0" "1 4 1" "1 16 1"; do set -- $cfg
      echo "page-locked PCM $1, slices $2 (OPUSGPU_HOST_PARTS), sliced flow $3 (OPUSGPU_HOST_SLICES): $(OPUSGPU_HOST_PARTS=$2 OPUSGPU_HOST_SLICES=$3 timeout -k 5 120 build_ab/host_path_rate 65536 14 $1 1 | tail -1)"
    done; } > $out/host_path_rate.txt 2>&1
  tools/host_path_timeline.sh collect/host_timeline 65536 6 1 1 > $out/host_path_timeline.txt 2>&1
  timeout -k 10 300 python tools/launch_jitter.py > $out/launch_jitter.txt 2>&1
  { timeout -k 10 500 python tools/soak_parity.py 32768 24 8 31; timeout -k 10 500 python tools/soak_parity.py --pipeline 32768 24 8 32; timeout -k 10 500 python tools/soak_parity.py --pipeline --masks 16384 32 4 77;
    timeout -k 10 400 python tools/soak_parity.py --host 32768 8 2 33; timeout -k 10 400 python tools/soak_parity.py --rfc 4096 12 4 34; } > $out/soaks.log 2>&1
  tail -4 $out/soaks.log
fi
