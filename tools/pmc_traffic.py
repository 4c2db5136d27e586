"""Per-launch HBM traffic of every kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; one counter per pass,
as the gfx950 TCC slots require).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_FETCH_SIZE -o x -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_WRITE_SIZE -o x -- python3 bench.py ...
    python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE/x_counter_collection.csv \
                                gpurun_out/pmc_WRITE_SIZE/x_counter_collection.csv > profiles/rNN_pmc_traffic.json

Units and corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950 FETCH_SIZE tallies the
128-B requests of wide coalesced reads (16 B per lane, plain loads and LDS-DMA alike) at 64 B, so the read side is
doubled; WRITE_SIZE is exact for 16-B-per-lane stores and for float atomics.  Infinity-Cache hits are included in both
(the counters sit on the L2's fabric side), so this is traffic leaving L2, an upper bound on HBM bytes.
"""
import csv
import json
import sys


def per_kernel(path, counter):
    acc = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        a = acc.setdefault(r["Kernel_Name"], [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return acc


def main(fetch_csv, write_csv):
    f = per_kernel(fetch_csv, "FETCH_SIZE")
    w = per_kernel(write_csv, "WRITE_SIZE")
    out = {}
    for name in sorted(set(f) | set(w)):
        nf, kf = f.get(name, (0, 0.0))
        nw, kw = w.get(name, (0, 0.0))
        rd = 2.0 * 1024.0 * kf / max(nf, 1)
        wr = 1024.0 * kw / max(nw, 1)
        out[name] = {"launches": max(nf, nw), "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
                     "bytes_per_launch": round(rd + wr)}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes); bytes = (2*FETCH_SIZE + "
                         "WRITE_SIZE) * 1024 per launch", "kernels": out}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
