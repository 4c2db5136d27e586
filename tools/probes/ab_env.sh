#!/bin/bash
# same-box A/B of environment settings on the default bench: ab_env.sh "VAR=val [VAR2=val]" ...   (first column: base)
set -e
for r in 1 2 3; do
  for v in "" "$@"; do
    env $v timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-hbm-rows 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('${v:-base}', d['ms_per_step'])"
  done
done
