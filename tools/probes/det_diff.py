"""Deterministic mode: which tensors of two identical engines differ after each of STEPS steps (forward tensors, gradients,
parameters)?  A long run (B = 256, side, 100 steps) is the soak test for rare, timing-dependent corruption.
python tools/probes/det_diff.py [B] [side|serial] [STEPS]"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
for p in (ROOT, os.path.join(ROOT, "thesis-fmri-reconstruction_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from fmri_hip import ops  # noqa: E402
from fmri_hip.params import ArchConfig  # noqa: E402
from fmri_hip.steps import Stage1Step  # noqa: E402
from oracle import vaegan_oracle as O  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
side = len(sys.argv) > 2 and sys.argv[2] == "side"
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ops.set_deterministic(True)
ops._SIDE["on"] = side
DEV = "cuda:0"
data = O.synth_batch(B, O.ArchCfg.px64(), seed=1234, steps=1)
x, e, zp = data["x"].to(DEV), data["noise"][0, 0].to(DEV), data["noise"][0, 1].to(DEV)
a = Stage1Step(ArchConfig.px64(), DEV)
a.load_recipe(0, True)
b = Stage1Step(ArchConfig.px64(), DEV)
b.load_state_dict(a.state_dict())


def diff(name, ta, tb):
    if ta is None or tb is None or not torch.is_tensor(ta):
        return
    if not torch.equal(ta, tb):
        d = (ta.double() - tb.double()).abs()
        print(f"  DIFF {name}: {int((d > 0).sum())}/{ta.numel()} max {float(d.max()):.3e} (ref max {float(tb.double().abs().max()):.3e})")


for step in range(STEPS):
    if step % 10 == 0:
        print("step", step, flush=True)
    for st in (a, b):
        st.forward(x, e, zp)
        st.gate(B)
    torch.cuda.synchronize()
    for k in ("disc_in", "head32", "feat", "logit32", "prob"):
        diff("fw." + k, a.fw[k], b.fw[k])
    diff("scal", a.scal, b.scal)
    for st in (a, b):
        st.backward()
        ops.join_side()
    torch.cuda.synchronize()
    diff("scal.bwd", a.scal, b.scal)
    for (pre, na, nb) in (("enc.", a.enc, b.enc), ("dec.", a.dec, b.dec), ("dis.", a.dis, b.dis)):
        for k in na.group.grads:
            diff("grad." + pre + k, na.group.grads[k], nb.group.grads[k])
    for st in (a, b):
        st.apply()
    torch.cuda.synchronize()
    sa, sb = a.state_dict(), b.state_dict()
    for k in sa:
        diff("state." + k, sa[k], sb[k])
print("done")
