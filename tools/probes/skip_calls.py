"""GPU box: where the wall clock of the two-stream Stage-I step goes -- the step timed with whole FAMILIES of library
launches skipped (wrong numbers, right clock): an upper bound of what fusing / removing a family could buy in the real
schedule (a family's kernel time is not that bound: the side stream hides part of it, and a launch's gaps come on top).

    python tools/probes/skip_calls.py            (one process, same box, three rounds)
"""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "thesis-fmri-reconstruction_amd"))
import numpy as np, torch
from fmri_hip import lib, ops
from fmri_hip.params import ArchConfig
from fmri_hip.steps import Stage1Step
dev, B = "cuda:0", 256
rs = np.random.RandomState(0)
x = torch.from_numpy(rs.uniform(-1, 1, (B, 3, 64, 64)).astype(np.float32)).to(dev)
e, z = (torch.from_numpy(rs.standard_normal((B, 128)).astype(np.float32)).to(dev) for _ in range(2))
st = Stage1Step(ArchConfig.px64(), dev); st.load_recipe(0, False)
for _ in range(5): st.step(x, e, z)
real = lib.call
SKIP = set()
def call(name, *a):
    if name in SKIP: return
    return real(name, *a)
lib.call = call
FAM = {
    "nothing": (),
    "BN forward apply": ("fmri_bn_apply",),
    "BN forward folds": ("fmri_bn_fold_finalize", "fmri_bn_stats_finalize", "fmri_bn_finalize", "fmri_bn_finalize_s", "fmri_bn_fold", "fmri_bn_stats"),
    "BN backward reduce": ("fmri_bn_bwd_reduce", "fmri_bn_bwd_reduce2"),
    "BN backward folds": ("fmri_bn_bwd_fold",),
    "BN backward apply": ("fmri_bn_bwd_apply", "fmri_bn_bwd_apply2"),
    "BN cols (dense)": ("fmri_bn_cols_fwd", "fmri_bn_cols_fwd_s", "fmri_bn_cols_bwd"),
    "act_bwd+colsum+permute": ("fmri_act_bwd", "fmri_colsum_acc", "fmri_colsum_rows", "fmri_permute_chw"),
    "reduce_slabs": ("fmri_reduce_slabs",),
    "all weight gradients": ("fmri_wgrad",),
    "apply+pack": ("fmri_apply_batch", "fmri_pack_weight_batch", "fmri_pack_weight", "fmri_transpose_f16"),
}
def timed(n=50):
    for _ in range(6): st.step(x, e, z)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): st.step(x, e, z)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
res = {k: [] for k in FAM}
for rnd in range(3):
    for k, names in FAM.items():
        SKIP.clear(); SKIP.update(names)
        res[k].append(timed())
base = sorted(res["nothing"])[1]
for k, v in res.items():
    m = sorted(v)[1]
    print(f"skip {k:26s} {m:.3f} ms/step  ({m - base:+.3f})   runs {['%.3f' % t for t in v]}", flush=True)
