#!/bin/bash
# same-box A/B: tools/probes/libfmri_head.so (the library with one TU from the previous commit) against the in-tree build
set -e
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -q -x 2>&1 | tail -2
for r in 1 2 3; do
for v in head new; do
  if [ $v = head ]; then export FMRI_LIB_PATH=tools/probes/libfmri_head.so; else unset FMRI_LIB_PATH; fi
  timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'])"
done; done
for v in head new; do
  if [ $v = head ]; then export FMRI_LIB_PATH=tools/probes/libfmri_head.so; else unset FMRI_LIB_PATH; fi
  echo == $v; timeout -k 10 200 python tools/microbench_igemm.py 2>/dev/null | grep -E "dgrad|deconv"
done
