#!/bin/bash
# same-call A/B of environment settings on the Stage-I bench: ab_envs.sh "VAR=val [VAR2=val]" ...  ("X=0" = base)
for r in 1 2 3; do
  for v in "$@"; do
    env $v timeout -k 10 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-hbm-rows --no-pmc --no-gate-pass 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$v', d['ms_per_step'], d['launch'][:24], d['losses_finite'])"
  done
done
