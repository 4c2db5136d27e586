#!/bin/bash
# same-box A/B of environment settings on one bench workload: ab_env_wl.sh <workload> "VAR=val" ...
set -e
wl=$1; shift
for r in 1 2; do
  for v in "" "$@"; do
    env $v timeout -k 10 200 python bench.py --workload $wl --steps 60 --warmup 10 --no-cpu-baseline --no-hbm-rows 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$wl ${v:-base}', d['ms_per_step'])"
  done
done
