timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for w in celt_fb_stereo_64k hybrid_fb_stereo_256k mixed_pages_2m silk_nb_stereo_64k; do
timeout -k 10 200 python bench.py --workload $w --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w', round(d['ms_per_step'],3), round(d['value']), d['parity_check']['pcm_crc32'])"
done
OPUSGPU_PARSE_WIDE=0 timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('celt narrow', round(d['ms_per_step'],3))"
timeout -k 10 300 python tools/soak_parity.py --pipeline --masks 16384 32 2 78 2>&1 | tail -1
