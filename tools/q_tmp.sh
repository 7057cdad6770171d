for rep in 1 2; do for g in 1 2 3; do
OPUSGPU_PARSE_GROUPS=$g timeout -k 10 120 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('groups $g:', round(d['ms_per_step'],4))"
done; done
