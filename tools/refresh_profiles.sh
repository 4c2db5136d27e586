#!/bin/bash
# refresh of the judged artefacts: GPU tests, bench line, rocprof kernel table of the same command, PMC traffic
R=$PWD
timeout -k 10 600 python -m pytest tests -q -m gpu > gpurun_out/gpu_tests.log 2>&1; tail -3 gpurun_out/gpu_tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; cat gpurun_out/bench.json
cd /tmp && export TMPDIR=/tmp; cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -o x -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --eager --serial > gpurun_out/bench_rocprof.json 2> gpurun_out/bench_rocprof.err
cp $(find $R/gpurun_out/prof_final -name "*kernel_stats.csv" | head -1) gpurun_out/final_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -o x -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --eager --serial > gpurun_out/pmc_$c.log 2>&1
done
hipcc --offload-arch=gfx950 -O3 -o gpurun_out/mfma_peak tools/probes/mfma_peak.hip && timeout -k 5 120 gpurun_out/mfma_peak > gpurun_out/mfma_peak.log; cat gpurun_out/mfma_peak.log
