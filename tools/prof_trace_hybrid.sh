#!/bin/bash
# rocprofv3 kernel trace of the bench workload in its default launch mode (recorded forward, two-stream backward):
# keeps the last ~3 steps' rows with their queue ids so that the overlap of the two streams can be read off.
# usage: tools/prof_trace_hybrid.sh <out-prefix under gpurun_out/> [bench args...]
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o x -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hbm-rows "$@" > $OUT/bench.json 2> $OUT/bench.err
t=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 - "$t" "$OUT/kernel_trace_tail.csv" <<'PY'
import sys, csv
rows = list(csv.reader(open(sys.argv[1])))
hdr, body = rows[0], rows[1:]
body.sort(key=lambda r: int(r[hdr.index("Start_Timestamp")]))
keep = body[-1100:]
w = csv.writer(open(sys.argv[2], "w"))
cols = [hdr.index(c) for c in ("Kernel_Name", "Start_Timestamp", "End_Timestamp", "Queue_Id", "Stream_Id", "Grid_Size_X") if c in hdr]
w.writerow([hdr[c] for c in cols])
for r in keep:
    w.writerow([r[c][:70] for c in cols])
PY
rm -rf $OUT/trace
