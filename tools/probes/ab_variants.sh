#!/bin/bash
# same-box A/B of tools/probes/libfmri_<name>.so variants against the in-tree build: ab_variants.sh name [name ...]
set -e
for r in 1 2 3; do
for v in base "$@"; do
  if [ $v = base ]; then unset FMRI_LIB_PATH; else export FMRI_LIB_PATH=tools/probes/libfmri_$v.so; fi
  timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-hbm-rows 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'])"
done; done
