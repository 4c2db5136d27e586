#!/bin/bash
# rocprofv3 kernel table of the bench workload issued eagerly on ONE stream (per-kernel durations are the kernels' own).
# usage: tools/prof_stats.sh <out-prefix under gpurun_out/> [bench args...]
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o x -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-pmc --no-gate-pass --eager --serial "$@" > $OUT/bench.json 2> $OUT/bench.err
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv
t=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
# keep the trace small: one step's worth of rows from the middle is enough for per-launch inspection
python3 - "$t" "$OUT/kernel_trace_tail.csv" <<'PY'
import sys, csv
rows = list(csv.reader(open(sys.argv[1])))
hdr, body = rows[0], rows[1:]
keep = body[-1500:]
w = csv.writer(open(sys.argv[2], "w"))
cols = [hdr.index(c) for c in ("Kernel_Name", "Start_Timestamp", "End_Timestamp", "Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z", "Workgroup_Size_X", "LDS_Block_Size", "VGPR_Count") if c in hdr]
w.writerow([hdr[c] for c in cols])
for r in keep:
    w.writerow([r[c][:90] for c in cols])
PY
rm -rf $OUT/trace
