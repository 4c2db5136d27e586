#!/usr/bin/env python3
"""gfx950 store-data hazard scan of the built library (dev container or GPU box; needs only llvm-objdump).

A VMEM store of more than 64 bits (``*_store_dwordx3/x4``) reads its data VGPRs after it has issued; a VALU write of one
of them in the next two issue slots can land first.  hipcc pads such pairs with wait states EXCEPT when the store carries
an SGPR offset.  Observed on MI355X:
  round 3, `igemm_tc5w_kernel<16,1,true>`: `buffer_store_dwordx4 v[14:17], ..., s20 offen` + `v_pk_add_f32 v[14:15], ...`
           back to back -> wrong second dword in every launch; one wait state did not cure it, two did;
  round 4, `igemm_tc5w_kernel<8,0,true>`:  `buffer_store_dwordx4 v[24:27], v28, ..., s15 offen` + `v_cndmask_b32 v24, ...`
           back to back -> wrong FIRST dword in ~1 launch of a few hundred, only with another stream's kernel running
           (tools/probes/store_hazard_stress.py); found by the bit-exact self-comparison tests of deterministic mode.

This script disassembles every gfx950 code object of libfmri_hip.so and counts, for each wide VMEM store, the wait states
(instructions issued, `s_nop N` = N + 1) in front of the first VALU instruction that writes one of its data registers.
Fewer than NEED (default 2: what hipcc itself leaves behind stores without an SGPR offset, and what cured both observed
cases) is a hazard.  Exit status 1 if any is found.  fmri_hip/build.py runs it as a build step and tests/test_abi.py on
every CPU test run, so that a scheduling change cannot ship the pattern.  (VMEM loads into the data registers are not
hazards: the memory pipeline executes a wave's VMEM instructions in order.)

usage: tools/scan_store_hazard.py [path/to/libfmri_hip.so] [--need N]
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


def scan_text(dis, need=2):
    """Hits in one llvm-objdump -d listing: ([(kernel, store, valu, wait states)], wide stores seen, symbols seen)."""
    hits, nstores, nkern = [], 0, 0
    kern, pend = "?", []          # pend: [(store text, data regs, wait states elapsed since the store)]
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
        if op and dst:
            for st, regs, ws in pend:
                if regs & dst:
                    hits.append((kern, st.strip(), body.strip(), ws))
        nm = re.match(r"^\s*s_nop\s+(\d+)", body)
        inc = int(nm.group(1)) + 1 if nm else 1
        pend = [(s, r, w + inc) for s, r, w in pend if w + inc < need]
        sm = STORE.match(body)
        if sm:
            nstores += 1
            pend.append((body, store_data(sm.group(1), sm.group(2)), 0))
    return hits, nstores, nkern


def scan(lib, need=2):
    tmp, objs = code_objects(lib)
    hits, nstores, nkern = [], 0, 0
    try:
        for o in objs:
            dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", o], check=True, capture_output=True, text=True).stdout
            h, ns, nk = scan_text(dis, need)
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
    need = 2
    it = iter(argv)
    for a in it:
        if a == "--need":
            need = int(next(it))
        else:
            lib = a
    hits, nstores, nkern, nobj = scan(lib, need)
    print(f"{nobj} code objects, {nkern} symbols, {nstores} wide VMEM stores scanned, {need} wait states required in front "
          f"of a VALU write of the store data: {len(hits)} hazards")
    for k, s, v, ws in hits:
        print(f"  {k}\n      {s}\n      {v}      ({ws} wait states)")
    return 1 if hits else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
