#!/bin/bash
# per-kernel A/B under rocprofv3: ab/old.so vs ab/new.so
P=thesis-fmri-reconstruction_amd/fmri_hip
R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in old new; do
  cp $R/ab/$v.so $R/$P/libfmri_hip.so
  cd $R
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab_$v -o x -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --eager > /dev/null 2>&1
  f=$(find $R/gpurun_out/ab_$v -name "*kernel_stats.csv" | head -1)
  cp $f $R/gpurun_out/ab_${v}_stats.csv
done
