#!/bin/bash
# rocprofv3 PMC passes of the bench workload (eager, one stream): MFMA utilisation and the traffic leaving L2.
# Each pass is its own run (--pmc with --kernel-trace only; FETCH_SIZE and WRITE_SIZE do not fit one pass).
# usage: tools/pmc_passes.sh <out dir under gpurun_out/> [bench args...]
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --steps 3 --warmup 2 --eager --serial --no-cpu-baseline --no-hbm-rows $@"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/mfma -o x -- python3 $B > $OUT/mfma.json 2> $OUT/mfma.err
python3 $R/tools/pmc_mfma.py $(find $OUT/mfma -name "*counter_collection.csv" | head -1) > $OUT/pmc_mfma.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o x -- python3 $B > $OUT/fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o x -- python3 $B > $OUT/write.json 2> $OUT/write.err
python3 $R/tools/pmc_traffic.py $(find $OUT/fetch -name "*counter_collection.csv" | head -1) $(find $OUT/write -name "*counter_collection.csv" | head -1) > $OUT/pmc_traffic.json
rm -rf $OUT/mfma $OUT/fetch $OUT/write
