#!/usr/bin/env python3
"""gfx950 store-data hazard scan of the built library (dev container or GPU box; needs only llvm-objdump).

Observed on MI355X (round 3, `igemm_tc5w_kernel<16,1,true>`): a `buffer_store_dwordx4 v[a:a+3], …, sN offen` whose NEXT
instruction is a packed-fp32 VALU op (`v_pk_add_f32` / `v_pk_mul_f32` …) writing v[a:a+1] stores the NEW value of v[a+1]:
the second dword of the 16 bytes is the VALU result, not the register's content at issue.  The compiler's hazard recogniser
(ROCm 7.2 clang) leaves no wait state there when the store carries an SGPR offset.  One `s_nop` between the two cures it
(tools/probes/tc5w_check.py, variants ss1/ss4 in DESIGN.md §6).

This script disassembles every gfx950 code object of libfmri_hip.so and reports each VMEM store of more than 64 bits that
is followed, within WINDOW instructions (default 1: the observed case), by a VALU instruction writing any of its data
registers.  Exit status 1 if any is found.  tests/test_abi.py runs it so that a scheduling change cannot ship the pattern.

usage: tools/scan_store_hazard.py [path/to/libfmri_hip.so] [--window N] [--all-valu]
"""
import os
import re
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
STORE = re.compile(r"^\s*((?:buffer|global|flat|scratch)_store_dwordx[34])\s+(.*)$")
VREG = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")


def vregs(tok):
    m = VREG.fullmatch(tok.strip())
    if not m:
        return set()
    if m.group(3) is not None:
        return {int(m.group(3))}
    return set(range(int(m.group(1)), int(m.group(2)) + 1))


def store_data(op, args):
    a = [t.strip() for t in args.split(",")]
    # buffer_store: vdata first; global/flat/scratch_store: vaddr first, vdata second
    return vregs(a[0]) if op.startswith("buffer") else (vregs(a[1]) if len(a) > 1 else set())


def valu_dst(line):
    m = re.match(r"^\s*(v_[a-z0-9_]+)\s+([^,]+)", line)
    if not m or m.group(1).startswith(("v_cmp", "v_cmpx")):
        return None, set()
    return m.group(1), vregs(m.group(2))


def code_objects(lib):
    """gfx950 code objects of the fat binary, extracted into a scratch directory (llvm-objdump --offloading writes
    them beside its input, so the input is a copy)."""
    tmp = tempfile.mkdtemp(prefix="fmri_scan_")
    cp = os.path.join(tmp, "lib.so")
    with open(lib, "rb") as f, open(cp, "wb") as g:
        g.write(f.read())
    subprocess.run([OBJDUMP, "--offloading", cp], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return tmp, sorted(os.path.join(tmp, f) for f in os.listdir(tmp) if "gfx950" in f)


def scan_text(dis, window=1, packed_only=True):
    """Hits in one llvm-objdump -d listing: ([(kernel, store, valu)], wide stores seen, symbols seen)."""
    hits, nstores, nkern = [], 0, 0
    kern, pend = "?", []          # pend: [(store text, data regs, instructions left)]
    for line in dis.splitlines():
        lm = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if lm:
            kern, pend = lm.group(1), []
            nkern += not kern.startswith(("L", "."))
            continue
        body = line.split("//")[0]
        if not body.strip() or body.strip().startswith((";", ".")):
            continue
        op, dst = valu_dst(body)
        if op and dst and (not packed_only or op.startswith("v_pk_") or len(dst) > 1):
            for st, regs, _ in pend:
                if regs & dst:
                    hits.append((kern, st.strip(), body.strip()))
        pend = [(s, r, n - 1) for s, r, n in pend if n > 1]
        sm = STORE.match(body)
        if sm:
            nstores += 1
            pend.append((body, store_data(sm.group(1), sm.group(2)), window))
    return hits, nstores, nkern


def scan(lib, window=1, packed_only=True):
    tmp, objs = code_objects(lib)
    hits, nstores, nkern = [], 0, 0
    try:
        for o in objs:
            dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", o], check=True, capture_output=True, text=True).stdout
            h, ns, nk = scan_text(dis, window, packed_only)
            hits += h
            nstores += ns
            nkern += nk
    finally:
        for f in os.listdir(tmp):
            os.unlink(os.path.join(tmp, f))
        os.rmdir(tmp)
    return hits, nstores, nkern, len(objs)


def main(argv):
    here = os.path.dirname(os.path.abspath(__file__))
    lib = os.path.join(here, "..", "thesis-fmri-reconstruction_amd", "fmri_hip", "libfmri_hip.so")
    window, packed_only = 1, True
    it = iter(argv)
    for a in it:
        if a == "--window":
            window = int(next(it))
        elif a == "--all-valu":
            packed_only = False
        else:
            lib = a
    hits, nstores, nkern, nobj = scan(lib, window, packed_only)
    print(f"{nobj} code objects, {nkern} symbols, {nstores} wide VMEM stores scanned, window {window}, "
          f"{'multi-register VALU writers' if packed_only else 'all VALU writers'}: {len(hits)} hazards")
    for k, s, v in hits:
        print(f"  {k}\n      {s}\n      {v}")
    return 1 if hits else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
