#!/bin/bash
# Weak-scaling sweep of the headline bench on one node: N = 1, 2, 4, 8 ranks (one per GPU, RCCL over xGMI), 256 images
# per GPU, per-rank BatchNorm statistics (the default: each rank normalises over its own 256 images, which is the batch
# the single-GPU reference normalises over; add --sync-bn for global-batch statistics, the mode the 2-rank parity tests
# run).  Prints one JSON line per N.  usage: tools/scale_sweep.sh [bench args...]
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd "$(dirname "$0")/.."
for n in 1 2 4 8; do
  if [ "$n" = 1 ]; then
    python bench.py --gpus 1 --no-cpu-baseline --no-hbm-rows "$@"
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) \
      bench.py --gpus $n --no-cpu-baseline "$@"
  fi
done
