timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for w in hybrid_fb_stereo_256k mixed_pages_2m celt_fb_stereo_64k; do
timeout -k 10 200 python bench.py --workload $w --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w', round(d['ms_per_step'],3), d['parity_check']['pcm_crc32'])"
done
