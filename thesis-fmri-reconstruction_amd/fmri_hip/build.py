"""Build libfmri_hip.so (gfx950 only) in-tree with hipcc.

    python -m fmri_hip.build            # from thesis-fmri-reconstruction_amd/
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.abspath(os.path.join(HERE, "..", "csrc"))
LIB = os.path.join(HERE, "libfmri_hip.so")
SOURCES = ["igemm.hip", "igemm_narrow.hip", "igemm_tc32.hip", "igemm_tc5.hip", "igemm_tc5w.hip", "igemm_c5.hip", "igemm_c5w.hip", "wgrad.hip", "wgrad_win.hip", "wgrad_narrow.hip", "layout.hip", "norm.hip", "loss.hip", "mlp.hip", "metrics.hip", "ingest.hip", "api.hip"]
HEADERS = ["common.h", "kernels.h", os.path.join("..", "..", "include", "fmri_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wno-unused-result"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + hdrs):
            jobs.append([hipcc, *FLAGS, "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
        _scan_store_hazard(verbose)
    return LIB


def _scan_store_hazard(verbose: bool):
    """Build step: the gfx950 store-data hazard scan of the linked library (tools/scan_store_hazard.py; csrc/common.h
    FMRI_STORE_FENCE).  A hit fails the build -- the library is removed, so that nothing can load it."""
    import importlib.util
    tool = os.path.abspath(os.path.join(HERE, "..", "..", "tools", "scan_store_hazard.py"))
    if not os.path.exists(tool):
        return
    spec = importlib.util.spec_from_file_location("scan_store_hazard", tool)
    scan = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(scan)
    if not os.path.exists(scan.OBJDUMP):
        if verbose:
            print("store-hazard scan skipped: llvm-objdump not found", file=sys.stderr)
        return
    hits, nstores, _, _ = scan.scan(LIB)
    if hits:
        os.unlink(LIB)
        raise RuntimeError(f"gfx950 store-data hazard: {len(hits)} wide VMEM stores of the build are followed by a VALU "
                           "write of their data registers with fewer than two wait states (FMRI_STORE_FENCE):\n"
                           + "\n".join(f"  {k}: {s}  ->  {v}  ({ws} wait states)" for k, s, v, ws in hits[:12]))
    if verbose:
        print(f"store-hazard scan: {nstores} wide VMEM stores, 0 hazards")


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
