#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof_pmc.sh <tag> [bench args...]
# Runs the bench under rocprofv3: one kernel-trace/stats pass and several separate --pmc passes (never combined
# with other trace domains), then prints the per-launch averages of the decode kernels.  Output: gpurun_out/<tag>/
# The --pmc passes run the steps IN ORDER (--pipeline off): counter collection serialises kernel dispatches, and a windowed
# pipelined step holds its reconstruction behind a stream memory wait for the NEXT step's parse kernel to start -- which a
# serialised queue never lets happen (the pass would sit in that wait until its timeout).  Counters per kernel do not depend on
# what runs next to it; durations next to the neighbours come from the kernel-trace pass, which runs the default (pipelined) bench.
# For the same reason they run steps with SILK frames as ONE chain (OPUSGPU_HALVES=0): one launch of every kernel per step, so
# that "per launch" is "per step"; the counters of a kernel do not depend on how its frames are cut into launches.
# PROF_PARSE_WIDE=2: the counter passes use the parse kernel of PIPELINED steps (k_celt_parse64, 64 frames per wave) in their in-order
# steps -- for workloads whose bench run is pipelined (CELT-only, hybrid), so that the counters are those of the kernel that runs.
# PROF_FRAMES: frames per launch for the per-frame figures (default 65536).
tag=$1; shift
out=$PWD/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
export BENCH_E2E=0 # (mixed_pages_2m: no overlapped end-to-end run behind the timed steps -- its batches would count as launches)
args="--steps 3 --warmup 1 --no-cpu-baseline --no-other-configs $*"
# the kernel-trace pass: the bench as it is (pipelined, windowed, in halves), 10 + 2 steps
export PROF_STEPS=12
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs $* > $out/trace.log 2>&1 || exit 1
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVES" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU" \
  "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INST_CYCLES_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VSKIPPED SQ_THREAD_CYCLES_VALU" \
  "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_EA0_WRREQ" ; do
  i=$((i+1))
  echo "pmc pass $i: $set"
  OPUSGPU_HALVES=0 OPUSGPU_PARSE_WIDE=${PROF_PARSE_WIDE:-1} timeout -k 5 ${PROF_PASS_TIMEOUT:-150} rocprofv3 --pmc $set --output-format csv -d $out/pmc$i -o p -- python3 bench.py $args --pipeline off > $out/pmc$i.log 2>&1 || { echo "pmc pass $i failed"; tail -3 $out/pmc$i.log; exit 1; }
done
python3 tools/pmc_summary.py $out ${PROF_FRAMES:-65536} | tee $out/summary.txt
