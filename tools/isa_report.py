#!/usr/bin/env python3
"""Register / spill / instruction-mix table of every kernel in libfmri_hip.so, from the code objects' own metadata and
disassembly (llvm-readelf --notes, llvm-objdump -d).  Dev container or GPU box; writes plain text to stdout.

usage: tools/isa_report.py [path/to/libfmri_hip.so] > profiles/rNN_isa.txt
"""
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import scan_store_hazard as ssh     # noqa: E402  (code-object extraction)

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
FILT = "c++filt"
FIELDS = [".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count",
          ".private_segment_fixed_size", ".group_segment_fixed_size"]
COUNT = [("mfma", r"v_mfma_"), ("readlane", r"v_readlane_b32"), ("writelane", r"v_writelane_b32"),
         ("scratch", r"scratch_(load|store)"), ("s_nop", r"s_nop"), ("waitcnt", r"s_waitcnt"), ("barrier", r"s_barrier")]


def short(name):
    m = re.match(r"_ZN4fmri(\d+)", name)          # (c++filt does not know the _Float16 mangling: keep the bare name)
    if m:
        n0 = m.end()
        name = name[n0:n0 + int(m.group(1))] + ("<" + ",".join(re.findall(r"Li(\d+)E", name)) + ">" if "ILi" in name else "")
    name = re.sub(r"\(.*$", "", name)
    return name.replace("void ", "").replace("fmri::", "")


def main(argv):
    here = os.path.dirname(os.path.abspath(__file__))
    lib = argv[0] if argv else os.path.join(here, "..", "thesis-fmri-reconstruction_amd", "fmri_hip", "libfmri_hip.so")
    tmp, objs = ssh.code_objects(lib)
    rows = []
    try:
        for o in objs:
            notes = subprocess.run([READELF, "--notes", o], check=True, capture_output=True, text=True).stdout
            meta = {}
            for blk in notes.split("\n  - .agpr_count:")[1:]:
                blk = ".agpr_count:" + blk
                nm = re.search(r"\.name:\s+(\S+)", blk)
                if not nm:
                    continue
                meta[nm.group(1)] = {f: int(m.group(1)) if (m := re.search(re.escape(f) + r":\s+(\d+)", blk)) else 0 for f in FIELDS}
            dis = subprocess.run([ssh.OBJDUMP, "-d", "--no-show-raw-insn", o], check=True, capture_output=True, text=True).stdout
            cur, counts = None, {}
            for line in dis.splitlines():
                lm = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if lm:
                    cur = lm.group(1) if lm.group(1) in meta else cur
                    counts.setdefault(cur, {k: 0 for k, _ in COUNT} | {"instr": 0})
                    continue
                if cur is None or not line.strip() or line.lstrip().startswith((";", ".")):
                    continue
                c = counts[cur]
                c["instr"] += 1
                for k, pat in COUNT:
                    if re.search(r"^\s*" + pat, line):
                        c[k] += 1
            for k, m in meta.items():
                rows.append((k, m, counts.get(k, {})))
    finally:
        for f in os.listdir(tmp):
            os.unlink(os.path.join(tmp, f))
        os.rmdir(tmp)
    names = subprocess.run([FILT], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
    print("# libfmri_hip.so: per-kernel registers, spills and instruction mix (code-object metadata + disassembly)")
    print(f"{'kernel':58s} {'vgpr':>4s} {'agpr':>4s} {'sgpr':>4s} {'vspill':>6s} {'sspill':>6s} {'scratchB':>8s} {'ldsB':>6s} "
          f"{'instr':>6s} {'mfma':>5s} {'rdlane':>6s} {'wrlane':>6s} {'scr.ops':>7s} {'s_nop':>5s} {'barrier':>7s}")
    for (k, m, c), nm in sorted(zip(rows, names), key=lambda t: -t[0][2].get("mfma", 0)):
        print(f"{short(nm)[:58]:58s} {m['.vgpr_count']:4d} {m['.agpr_count']:4d} {m['.sgpr_count']:4d} {m['.vgpr_spill_count']:6d} "
              f"{m['.sgpr_spill_count']:6d} {m['.private_segment_fixed_size']:8d} {m['.group_segment_fixed_size']:6d} "
              f"{c.get('instr', 0):6d} {c.get('mfma', 0):5d} {c.get('readlane', 0):6d} {c.get('writelane', 0):6d} "
              f"{c.get('scratch', 0):7d} {c.get('s_nop', 0):5d} {c.get('barrier', 0):7d}")


if __name__ == "__main__":
    main(sys.argv[1:])
