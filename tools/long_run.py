"""Debug helper (GPU box): Stage-I losses over many steps on one repeated batch (does the fp16 engine stay finite where
the fp32 oracle does?).  usage: python tools/long_run.py [steps] [batch] [rotate]"""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "thesis-fmri-reconstruction_amd"))
import torch
from oracle import vaegan_oracle as O
from fmri_hip.params import ArchConfig
from fmri_hip.steps import Stage1Step
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
rotate = int(sys.argv[3]) if len(sys.argv) > 3 else 0
cfg = O.ArchCfg.px64()
st = Stage1Step(ArchConfig.px64(), "cuda:0")
st.load_recipe(0, False)
data = O.synth_batch(B, cfg, seed=1234, steps=1)
x, e, zp = data["x"].cuda(), data["noise"][0, 0].cuda(), data["noise"][0, 1].cuda()
for i in range(steps):
    if rotate:
        d = O.synth_batch(B, cfg, seed=1234 + i, steps=1)
        x, e, zp = d["x"].cuda(), d["noise"][0, 0].cuda(), d["noise"][0, 1].cuda()
    st.step(x, e, zp)
    if i % 5 == 0 or i == steps - 1:
        l = st.logs()
        amax = {k: float(v.float().abs().max()) for k, v in (("feat", st.fw["feat"]), ("head", st.fw["head32"]))}
        print(i, {k: round(l[k], 4) for k in ("loss_encoder", "loss_decoder", "loss_discriminator", "kl", "mse", "nle")},
              l["train_dis"], l["train_dec"], amax, flush=True)
