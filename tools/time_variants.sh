#!/bin/bash
# usage: tools/time_variants.sh lib1.so lib2.so ...   (GPU box) -- prints ms/step for each experimental build
for lib in "$@"; do
  OPUSGPU_LIB=$PWD/$lib timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    line=line.strip()
    if line.startswith('{'):
        d=json.loads(line); print('$lib', 'ms/step=%.2f'%d['ms_per_step'], 'frames/s=%.0f'%d['value'])
    elif line: print('$lib', line[:200])
"
done
