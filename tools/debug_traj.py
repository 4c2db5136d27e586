"""Debug helper (GPU box): loss trajectory of the HIP Stage-I step over repeated-batch steps (finite-ness, scale health)."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "thesis-fmri-reconstruction_amd"))
import torch
from oracle import vaegan_oracle as O
from fmri_hip.params import ArchConfig
from fmri_hip.steps import Stage1Step

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16
cfg_o = O.ArchCfg.px64()
data = O.synth_batch(B, cfg_o, seed=1234, steps=1)
st = Stage1Step(ArchConfig.px64(), "cuda:0"); st.load_recipe(0, False)
x, e, z = data["x"].cuda(), data["noise"][0, 0].cuda(), data["noise"][0, 1].cuda()
for s in range(N):
    st.step(x, e, z)
    l = st.logs(); sc = st.scal.tolist()
    print(s, {k: round(l[k] / B, 4) for k in ("loss_encoder", "loss_discriminator", "kl", "mse", "bce_orig", "bce_pred", "bce_samp")},
          l["train_dis"], l["train_dec"], "nA %.3g nB %.3g" % (sc[10], sc[11]),
          "gmax", {n: float(g.group.grad.abs().max()) for n, g in (("e", st.enc), ("d", st.dec), ("s", st.dis))}, flush=True)
