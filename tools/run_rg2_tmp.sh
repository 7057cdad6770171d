KS_ARGS="--pipeline off" tools/kstats.sh default build_exp/lib_rg2np.so
echo "--- pipelined, library defaults"
tools/kstats.sh default build_exp/lib_rg2np.so
echo "--- pipelined, no head start, no post delay, no post pad"
export OPUSGPU_HEAD_START_US=0 OPUSGPU_POST_DELAY_US=0 OPUSGPU_POST_PAD=0
tools/kstats.sh default build_exp/lib_rg2np.so
