#!/bin/bash
# rocprofv3 kernel table of tools/microbench_igemm.py with the wide stride-2 kernel on every eligible layer / off
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/c5w_ab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for m in all off; do
  export FMRI_C5W=$m
  REP=20 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t_$m -o x -- python3 $R/tools/microbench_igemm.py > $OUT/mb_$m.log 2>&1
  cp $(find $OUT/t_$m -name "*kernel_stats.csv" | head -1) $OUT/stats_$m.csv
  python3 - $(find $OUT/t_$m -name "*kernel_trace.csv" | head -1) > $OUT/c5_$m.txt <<'PY'
import sys, csv, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    if "igemm_c5" not in n: continue
    k = (n[:60], r["Grid_Size_X"], r["Grid_Size_Y"])
    d.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v = sorted(v)
    print(k, "n", len(v), "median us %.1f" % v[len(v) // 2], "min %.1f" % v[0])
PY
  rm -rf $OUT/t_$m
done
