"""Debug helper (GPU box): per-tensor gradient / forward errors of the HIP Stage-I step vs the CPU oracle."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "thesis-fmri-reconstruction_amd"))
import torch
from oracle import vaegan_oracle as O
from fmri_hip.params import ArchConfig
from fmri_hip.steps import Stage1Step

arch = sys.argv[1] if len(sys.argv) > 1 else "px64"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cfg_o, cfg_e = getattr(O.ArchCfg, arch)(), getattr(ArchConfig, arch)()
data = O.synth_batch(B, cfg_o, seed=1234, steps=1)
st = Stage1Step(cfg_e, "cuda:0"); st.load_recipe(0, True)
st.forward(data["x"].cuda(), data["noise"][0, 0].cuda(), data["noise"][0, 1].cuda()); st.gate(B); st.backward()
P = O.fill_state(O.vaegan_spec(cfg_o), 0, True)
opts = {n: O.OptState(kind="rmsprop", lr=1e-4) for n in ("encoder", "decoder", "discriminator")}
ref = O.stage1_step(P, opts, data["x"], data["noise"][0, 0], data["noise"][0, 1], cfg_o, keep_grads=True)
print("logs", st.logs()); print("ref ", ref["logs"])
def err(a, b):
    a, b = a.detach().float().cpu().reshape(-1), b.detach().float().cpu().reshape(-1)
    return ((a - b).norm() / (b.norm() + 1e-30)).item(), (torch.dot(a, b) / (b.norm() ** 2 + 1e-30)).item()
for k, v in st.outputs().items():
    print("fw %-16s relerr %.3e proj %.5f" % ((k,) + err(v, ref["fw"][k])))
g = st.named_grads()
for k, r in ref["grads"].items():
    e, p = err(g[k], r)
    print("grad %-40s relerr %.3e proj %.5f  |ref| %.3e" % (k, e, p, r.norm().item()))
