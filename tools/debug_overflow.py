"""Debug helper (GPU box): print max|.| of every fp16 cotangent produced during the backward of a given step."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "thesis-fmri-reconstruction_amd"))
import torch
from oracle import vaegan_oracle as O
from fmri_hip import ops
from fmri_hip.params import ArchConfig
from fmri_hip.steps import Stage1Step

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
WATCH = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2, 3]
on = [False]
def wrap(cls, name, pick):
    f = getattr(cls, name)
    def g(self, *a, **k):
        r = f(self, *a, **k)
        if on[0]:
            t = pick(r)
            if t is not None:
                tf = t.float()
                tag = getattr(self, "prefix", None) or getattr(self, "kind", "dense")
                print("   %-10s %-28s shape %-22s max %.4g rms %.4g finite %s" % (name, tag, tuple(t.shape), tf.abs().max().item(), tf.pow(2).mean().sqrt().item(), bool(torch.isfinite(tf).all())))
        return r
    setattr(cls, name, g)
wrap(ops.ConvLayer, "dgrad", lambda r: r)
wrap(ops.DenseLayer, "dgrad", lambda r: r[0] if r[0] is not None else r[1])
wrap(ops.BatchNorm, "backward", lambda r: r[0])
cfg_o = O.ArchCfg.px64()
data = O.synth_batch(B, cfg_o, seed=1234, steps=1)
st = Stage1Step(ArchConfig.px64(), "cuda:0"); st.load_recipe(0, False)
x, e, z = data["x"].cuda(), data["noise"][0, 0].cuda(), data["noise"][0, 1].cuda()
for s in range(max(WATCH) + 1):
    fw = st.forward(x, e, z); st.gate(B)
    on[0] = s in WATCH
    if on[0]:
        print("== step", s, "scal", [round(v, 4) for v in st.scal.tolist()[:13]])
        h = fw["head32"]; print("   head32 max", h.abs().max().item(), "mu max", h[:, :128].abs().max().item(), "lv max", h[:, 128:].max().item(), "lv min", h[:, 128:].min().item())
        for i, r in enumerate(fw["ectx"]["raws"]): print("   enc raw", i, "max", r.float().abs().max().item())
        print("   enc raw_fc max", fw["ectx"]["raw_fc"].float().abs().max().item())
    st.backward(); on[0] = False
    st.apply()
