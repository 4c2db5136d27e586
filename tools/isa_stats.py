#!/usr/bin/env python3
"""Per-kernel register / spill / instruction-mix summary of a gfx950 assembly file (hipcc --save-temps *.s).

    python tools/isa_stats.py /tmp/isa/igemm_c5-hip-amdgcn-amd-amdhsa-gfx950.s [...]

For every kernel: the metadata the assembler printed (.sgpr_count, .sgpr_spill_count, .vgpr_count, .vgpr_spill_count,
scratch bytes, LDS) and counts of the instructions that matter in an MFMA loop: v_mfma, v_readlane / v_writelane (SGPR
spill traffic), s_nop, s_load (kernarg re-loads), scratch_ / buffer_ spill traffic, v_readfirstlane, ds_read, s_waitcnt.
"""
import re
import subprocess
import sys
from collections import OrderedDict


def demangle(n):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip()
    except OSError:
        return n


def parse(path):
    txt = open(path).read()
    kernels = OrderedDict()
    # bodies: from "<name>:" after ".type <name>,@function" to ".Lfunc_end"
    for m in re.finditer(r"^\s*\.type\s+(\S+),@function\n(.*?)^\.Lfunc_end\d+:", txt, re.S | re.M):
        name, body = m.group(1), m.group(2)
        c = dict(mfma=0, readlane=0, writelane=0, s_nop=0, s_load=0, scratch=0, readfirstlane=0, ds_read=0, waitcnt=0,
                 lines=0)
        for line in body.splitlines():
            s = line.strip()
            if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
                continue
            op = s.split()[0]
            c["lines"] += 1
            if op.startswith("v_mfma"): c["mfma"] += 1
            elif op.startswith("v_readlane"): c["readlane"] += 1
            elif op.startswith("v_writelane"): c["writelane"] += 1
            elif op == "s_nop": c["s_nop"] += 1
            elif op.startswith("s_load"): c["s_load"] += 1
            elif op.startswith("scratch_"): c["scratch"] += 1
            elif op.startswith("v_readfirstlane"): c["readfirstlane"] += 1
            elif op.startswith("ds_read"): c["ds_read"] += 1
            elif op == "s_waitcnt": c["waitcnt"] += 1
        kernels[name] = c
    # metadata
    for m in re.finditer(r"- \.agpr_count:.*?\.name:\s+(\S+).*?(?=\n  - \.agpr_count:|\namdhsa\.target|\Z)", txt, re.S):
        blk, name = m.group(0), m.group(1)
        if name not in kernels:
            continue
        for key in ("sgpr_count", "sgpr_spill_count", "vgpr_count", "vgpr_spill_count", "private_segment_fixed_size",
                    "group_segment_fixed_size", "agpr_count"):
            mm = re.search(r"\.%s:\s+(\d+)" % key, blk)
            if mm:
                kernels[name][key] = int(mm.group(1))
    return kernels


def main():
    for path in sys.argv[1:]:
        for name, c in parse(path).items():
            if "sgpr_count" not in c:
                continue
            d = demangle(name)
            d = d.replace("fmri::", "").replace("void ", "")
            d = re.sub(r"\(.*\)$", "", d)
            print(f"{d:46s} sgpr {c.get('sgpr_count'):3d} spill {c.get('sgpr_spill_count'):3d} | vgpr {c.get('vgpr_count'):3d} "
                  f"agpr {c.get('agpr_count', 0):3d} spill {c.get('vgpr_spill_count'):3d} scratch {c.get('private_segment_fixed_size'):4d} | "
                  f"mfma {c['mfma']:5d} readlane {c['readlane']:4d} writelane {c['writelane']:4d} s_nop {c['s_nop']:4d} "
                  f"s_load {c['s_load']:3d} rfl {c['readfirstlane']:4d} scratch_op {c['scratch']:3d} lines {c['lines']}")


if __name__ == "__main__":
    main()
