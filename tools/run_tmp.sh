timeout -k 10 300 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py::test_fullsize_celt_pipelined_queued -x -q -m gpu 2>&1 | tail -2
echo "--- window on"; tools/kstats.sh default
echo "--- window off"; KS_ARGS="--window off" tools/kstats.sh default
echo "--- in order"; KS_ARGS="--pipeline off" tools/kstats.sh default
