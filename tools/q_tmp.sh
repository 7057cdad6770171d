for rep in 1 2; do for v in default sp64; do
  if [ $v = default ]; then unset OPUSGPU_LIB; else export OPUSGPU_LIB=$PWD/build_exp/lib_$v.so; fi
  for w in silk_nb_stereo_64k hybrid_fb_stereo_256k; do
  timeout -k 10 200 python bench.py --workload $w --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v $w', round(d['ms_per_step'],3), d['parity_check']['pcm_crc32'])"
  done
done; done
export OPUSGPU_LIB=$PWD/build_exp/lib_sp64.so
timeout -k 10 200 python bench.py --workload mixed_pages_2m --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('sp64 mixed', round(d['ms_per_step'],3), d['parity_check']['pcm_crc32'])"
