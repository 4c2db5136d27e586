"""MFMA utilisation per kernel from a rocprofv3 PMC pass of bench.py (one pass: SQ and GRBM counters share no slots).

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
        -d gpurun_out/pmc_mfma -o x -- python3 bench.py --steps 3 --warmup 2 --eager --serial --no-cpu-baseline --no-hbm-rows
    python tools/pmc_mfma.py gpurun_out/pmc_mfma/**/x_counter_collection.csv > profiles/rNN_pmc_mfma.json

Units (MI355X_MICROARCH.md): SQ_VALU_MFMA_BUSY_CYCLES counts cycles of every SIMD's matrix pipe (16 per
v_mfma_f32_16x16x32_f16), summed over the 1024 SIMDs; GRBM_GUI_ACTIVE is the sum over the 8 XCDs of the cycles the
dispatch was resident.  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024): the fraction of all matrix
pipes' cycles spent in MFMAs while the kernel ran (1.0 = every SIMD back-to-back in MFMAs = the dense peak AT THE CLOCK
THE CHIP HELD, which under MFMA load is below the 2.4 GHz the 2.5 PFLOP/s datasheet figure assumes).
"""
import csv
import json
import sys


def main(path):
    acc = {}
    for r in csv.DictReader(open(path)):
        a = acc.setdefault(r["Kernel_Name"], {"n": {}, "v": {}})
        c = r["Counter_Name"]
        a["n"][c] = a["n"].get(c, 0) + 1
        a["v"][c] = a["v"].get(c, 0.0) + float(r["Counter_Value"])
    out = {}
    tot_m = tot_g = 0.0
    for name, a in acc.items():
        m = a["v"].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        g = a["v"].get("GRBM_GUI_ACTIVE", 0.0)
        b = a["v"].get("SQ_BUSY_CYCLES", 0.0)
        n = max(a["n"].values())
        tot_m += m
        tot_g += g
        out[name] = {"launches": n, "mfma_busy_cycles_per_launch": round(m / n), "gui_active_per_launch": round(g / n),
                     "sq_busy_cycles_per_launch": round(b / n),
                     "mfma_util": round(m / (g / 8.0 * 1024.0), 4) if g > 0 else None}
    ranked = dict(sorted(out.items(), key=lambda kv: -kv[1]["gui_active_per_launch"] * kv[1]["launches"]))
    json.dump({"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace; "
                         "mfma_util = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024 SIMDs)",
               "whole_run_mfma_util": round(tot_m / (tot_g / 8.0 * 1024.0), 4) if tot_g > 0 else None,
               "kernels": ranked}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1])
