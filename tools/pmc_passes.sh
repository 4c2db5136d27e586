#!/bin/bash
# rocprofv3 PMC passes of the bench workload (eager, one stream): MFMA utilisation and the traffic leaving L2.
# Each pass is its own run (--pmc with --kernel-trace only; FETCH_SIZE and WRITE_SIZE do not fit one pass).
# usage: tools/pmc_passes.sh <out dir under gpurun_out/> [bench args...]
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B=("$R/bench.py" --steps 3 --warmup 2 --eager --serial --no-cpu-baseline --no-hbm-rows --no-gate-pass --no-pmc "$@")

# one counter pass: <name> <counters...>; prints the path of its counter CSV, or fails with the run's stderr
pass() {
    local name=$1; shift
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -o x -- python3 "${B[@]}" \
        > "$OUT/$name.json" 2> "$OUT/$name.err" || { echo "pass $name failed:" >&2; tail -20 "$OUT/$name.err" >&2; return 1; }
    local csv
    csv=$(find "$OUT/$name" -name "*counter_collection.csv" | head -1)
    if [ -z "$csv" ]; then
        echo "pass $name wrote no counter CSV:" >&2; tail -20 "$OUT/$name.err" >&2
        rm -rf "$OUT/$name"
        return 1
    fi
    echo "$csv"
}

mfma=$(pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE)
python3 "$R/tools/pmc_mfma.py" "$mfma" > "$OUT/pmc_mfma.json"
fetch=$(pass fetch FETCH_SIZE)
write=$(pass write WRITE_SIZE)
python3 "$R/tools/pmc_traffic.py" "$fetch" "$write" > "$OUT/pmc_traffic.json"
rm -rf "$OUT/mfma" "$OUT/fetch" "$OUT/write"
