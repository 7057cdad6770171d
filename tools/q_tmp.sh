KS_ARGS="--pipeline off" tools/kstats.sh default 2>&1 | tail -1
KS_ARGS="" tools/kstats.sh default 2>&1 | tail -1
