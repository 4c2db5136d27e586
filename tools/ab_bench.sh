#!/bin/bash
# A/B two builds of the library on the same box: ab/old.so vs ab/new.so
P=thesis-fmri-reconstruction_amd/fmri_hip
for r in 1 2; do
for v in old new; do
  cp ab/$v.so $P/libfmri_hip.so
  python bench.py --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'])"
done; done
for v in old new; do
  cp ab/$v.so $P/libfmri_hip.so
  echo == $v; python tools/microbench_igemm.py 2>/dev/null | grep TF; python tools/microbench_wgrad.py 2>/dev/null | grep TF
done
