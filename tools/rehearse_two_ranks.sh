#!/bin/bash
# 2-rank rehearsal of bench.py's multi-process control flow on ONE GPU (gloo over device tensors)
export FMRI_REHEARSE_ON_ONE_GPU=1 FMRI_DIST_BACKEND=gloo
for extra in "" "--sync-bn"; do
timeout -k 10 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 8 --warmup 3 --batch 64 $extra > gpurun_out/two.json 2> gpurun_out/two.err
echo "rc=$?"; cut -c1-260 gpurun_out/two.json; grep "probe\|fail\|Error" gpurun_out/two.err | head -5
done
