#!/bin/bash
# rocprofv3 kernel trace of the default (hybrid, two-stream) bench run; per-queue busy time, overlap and gaps of the last steps
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/timeline
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o x -- python3 $R/bench.py --steps 12 --warmup 8 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/bench.err
t=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 - "$t" > $OUT/summary.txt <<'PY'
import sys, csv, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 6 steps: find the optimizer kernel occurrences? simpler: take the last 40% of rows by time
t_end = int(rows[-1]["End_Timestamp"]); t_beg = int(rows[0]["Start_Timestamp"])
cut = t_end - int(6 * 7.0e6)          # ~6 steps of 7 ms
sel = [r for r in rows if int(r["Start_Timestamp"]) >= cut]
span = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e6
qs = collections.defaultdict(list)
for r in sel: qs[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
print(f"window {span:.2f} ms, {len(sel)} kernels, queues {[(q, len(v)) for q, v in qs.items()]}")
ev = []
for q, v in qs.items():
    busy = sum(e - s for s, e, _ in v) / 1e6
    print(f"queue {q}: busy {busy:.2f} ms = {busy / span:.2%} of the window")
    for s, e, _ in v: ev.append((s, 1)); ev.append((e, -1))
ev.sort()
cur = 0; last = ev[0][0]; t = collections.Counter()
for ts, d in ev:
    t[min(cur, 2)] += ts - last; last = ts; cur += d
tot = sum(t.values())
print("no kernel running %.2f%%, one %.2f%%, two or more %.2f%%" % tuple(100.0 * t[i] / tot for i in (0, 1, 2)))
# per kernel family: time while alone vs overlapped is too detailed; list the top kernels by total time in the window
fam = collections.Counter()
for r in sel: fam[r["Kernel_Name"].split("(")[0][:70]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for k, v in fam.most_common(14): print(f"{v / 6:8.3f} ms/step  {k}")
PY
head -c 300000 "$t" > /dev/null
python3 - "$t" "$OUT/last_step.csv" <<'PY'
import sys, csv
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_end = int(rows[-1]["End_Timestamp"])
sel = [r for r in rows if int(r["Start_Timestamp"]) >= t_end - int(8.0e6)]
t0 = int(sel[0]["Start_Timestamp"])
w = csv.writer(open(sys.argv[2], "w"))
w.writerow(["start_us", "dur_us", "queue", "grid", "wg", "lds", "vgpr", "kernel"])
for r in sel:
    w.writerow(["%.1f" % ((int(r["Start_Timestamp"]) - t0) / 1e3), "%.1f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3), r["Queue_Id"],
                r.get("Grid_Size_X", ""), r.get("Workgroup_Size_X", ""), r.get("LDS_Block_Size", ""), r.get("VGPR_Count", ""), r["Kernel_Name"].split("(")[0][:60]])
PY
rm -rf $OUT/trace
