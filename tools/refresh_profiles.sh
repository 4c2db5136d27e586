#!/bin/bash
# refresh of the judged artefacts (run on the GPU box from the repo root): GPU tests, the default bench line, the other
# workloads' lines, the rocprofv3 kernel table of the same command (one stream), PMC passes, per-layer microbenchmarks.
# Results land in gpurun_out/final/; copy what is to be judged into profiles/ (named per round).
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd $R
timeout -k 10 700 python -m pytest tests -q -m gpu > $OUT/gpu_tests.log 2>&1; tail -3 $OUT/gpu_tests.log
timeout -k 10 400 python bench.py > $OUT/bench_stage1.json 2> $OUT/bench_stage1.err; cut -c1-200 $OUT/bench_stage1.json
for w in stage2 dual1 stage3_px128; do
  timeout -k 10 300 python bench.py --workload $w --steps 100 --warmup 20 > $OUT/bench_$w.json 2> $OUT/bench_$w.err
done
timeout -k 10 200 python tools/microbench_igemm.py > $OUT/microbench_igemm.log 2>&1
timeout -k 10 300 python tools/time_other_steps.py > $OUT/other_steps.log 2>&1
bash tools/prof_stats.sh final/prof > /dev/null 2>&1
bash tools/pmc_passes.sh final/pmc > $OUT/pmc.log 2>&1
ls $OUT $OUT/prof $OUT/pmc
