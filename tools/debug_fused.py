"""Debug helper (GPU box): fused Stage1Step.step vs separate calls, and run-to-run spread of each."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "thesis-fmri-reconstruction_amd"))
import torch
from oracle import vaegan_oracle as O
from fmri_hip import ops
from fmri_hip.params import ArchConfig
from fmri_hip.steps import Stage1Step
DEV = "cuda:0"
B = 8
data = O.synth_batch(B, O.ArchCfg.px64(), seed=1234, steps=3)
x = data["x"].to(DEV)
def run(fused, side, nsteps=3):
    st = Stage1Step(ArchConfig.px64(), DEV); st.load_recipe(0, True)
    ops._SIDE["on"] = side
    for s in range(nsteps):
        e, zp = data["noise"][s, 0].to(DEV), data["noise"][s, 1].to(DEV)
        if fused: st.step(x, e, zp)
        else:
            st.forward(x, e, zp); st.gate(B); st.backward(); st.apply()
    ops.join_side(); torch.cuda.synchronize()
    return st.logs(), {k: v.float().cpu() for k, v in st.state_dict().items()}
for n in (1, 2, 3):
    runs = {name: run(f, s, n) for name, f, s in (("fused+side", True, True), ("sep+side", False, True), ("sep", False, False), ("sep again", False, False), ("fused noside", True, False))}
    base = runs["sep"]
    for name, (l, sd) in runs.items():
        worst = max(((sd[k] - base[1][k]).norm() / (base[1][k].norm() + 1e-20)).item() for k in sd)
        print(n, f"{name:14s} kl {l['kl']:.6f} enc {l['loss_encoder']:.4f} worst state diff vs sep {worst:.2e}")
