#!/bin/bash
# Where do a kernel's wave cycles go?  One PMC pass (SQ block, 8 slots) over a microbenchmark script:
#   WAIT_ANY (parked on s_waitcnt / barrier) + WAIT_INST_ANY (issue stalls) + ACTIVE_INST_ANY ~= WAVE_CYCLES.
# usage: tools/pmc_wait_breakdown.sh <out dir under gpurun_out/> <script.py> [args...]
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc -o x -- python3 $R/"$@" > $OUT/run.log 2> $OUT/run.err
python3 - $(find $OUT/pmc -name "*counter_collection.csv" | head -1) <<'PY'
import csv, sys
acc = {}
for r in csv.DictReader(open(sys.argv[1])):
    a = acc.setdefault(r["Kernel_Name"], {})
    a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k, a in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    w = a.get("SQ_WAVE_CYCLES", 0)
    if w <= 0 or "fmri" not in k:
        continue
    f = lambda n: a.get(n, 0.0) / w
    print(f"{k[:60]:60s} wait_any {f('SQ_WAIT_ANY'):.2f} wait_inst {f('SQ_WAIT_INST_ANY'):.2f} (lds {f('SQ_WAIT_INST_LDS'):.2f}) "
          f"active {f('SQ_ACTIVE_INST_ANY'):.2f} (lds {f('SQ_ACTIVE_INST_LDS'):.2f})  mfma_busy/busy {a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(a.get('SQ_BUSY_CYCLES', 1), 1):.2f}")
PY
rm -rf $OUT/pmc
