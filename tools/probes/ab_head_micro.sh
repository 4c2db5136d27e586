#!/bin/bash
# same-box A/B against tools/probes/libfmri_head.so (build_head.sh): kernel tests, the default bench three times each, per-layer times
set -e
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -q -x 2>&1 | tail -2
bash tools/probes/ab_variants.sh head
for v in head new; do
  if [ $v = head ]; then export FMRI_LIB_PATH=tools/probes/libfmri_head.so; else unset FMRI_LIB_PATH; fi
  echo == $v; timeout -k 10 200 python tools/microbench_igemm.py 2>/dev/null | grep -E "TF"
done
