timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for rep in 1 2; do
timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('celt', round(d['ms_per_step'],3), round(d['value']), d['parity_check']['pcm_crc32'])"
done
PMC_KERNELS=k_celt_recon_fb tools/pmc_quick.sh default 2>&1 | grep k_celt_recon_fb
