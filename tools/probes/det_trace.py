"""Deterministic mode + side stream: record the output of every dgrad / BatchNorm-backward / act_backward call of the
backward pass of two identical engines and report the first calls whose outputs differ, with the differing indices.
python tools/probes/det_trace.py [B] [steps]"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
for p in (ROOT, os.path.join(ROOT, "thesis-fmri-reconstruction_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from fmri_hip import nets, ops  # noqa: E402
from fmri_hip.params import ArchConfig  # noqa: E402
from fmri_hip.steps import Stage1Step  # noqa: E402
from oracle import vaegan_oracle as O  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ops.set_deterministic(True)
ops._SIDE["on"] = True
DEV = "cuda:0"
data = O.synth_batch(B, O.ArchCfg.px64(), seed=1234, steps=1)
x, e, zp = data["x"].to(DEV), data["noise"][0, 0].to(DEV), data["noise"][0, 1].to(DEV)
a = Stage1Step(ArchConfig.px64(), DEV)
a.load_recipe(0, True)
b = Stage1Step(ArchConfig.px64(), DEV)
b.load_state_dict(a.state_dict())

TRACE = None


def wrap(cls, name, pick=lambda r: r):
    orig = getattr(cls, name)

    def f(self, *args, **kw):
        r = orig(self, *args, **kw)
        if TRACE is not None:
            t = pick(r)
            if torch.is_tensor(t):
                TRACE.append((f"{cls.__name__}.{name}#{len(TRACE)} {tuple(t.shape)}", t.clone() if os.environ.get("CLONE") == "1" else t))
        return r
    setattr(cls, name, f)


wrap(ops.ConvLayer, "dgrad")
wrap(ops.ConvLayer, "forward")
wrap(ops.BatchNorm, "backward", lambda r: r[0])
wrap(ops.BatchNorm, "backward2", lambda r: r[0])
wrap(ops.BatchNorm, "forward", lambda r: r[0])
wrap(ops.DenseLayer, "dgrad", lambda r: r[0] if r[0] is not None else r[1])
wrap(ops.DenseLayer, "forward", lambda r: r[0] if r[0] is not None else r[1])

for step in range(steps):
    tr = []
    for st in (a, b):
        TRACE = []
        st.forward(x, e, zp)
        st.gate(B)
        st.backward()
        ops.join_side()
        torch.cuda.synchronize()
        tr.append(TRACE)
        TRACE = None
        st.apply()
        torch.cuda.synchronize()
    nd = 0
    for (na, ta), (nb, tb) in zip(*tr):
        if not torch.equal(ta, tb):
            d = (ta.float() - tb.float()).abs()
            idx = torch.nonzero(d > 0)
            print(f"step {step} DIFF {na}: {idx.shape[0]} elements, max {float(d.max()):.3e}; first {idx[:6].tolist()} last {idx[-3:].tolist()}")
            nd += 1
            if nd >= 4:
                break
    for (pre, na_, nb_) in (("enc.", a.enc, b.enc), ("dec.", a.dec, b.dec), ("dis.", a.dis, b.dis)):
        for k in na_.group.grads:
            ga, gb = na_.group.grads[k], nb_.group.grads[k]
            if not torch.equal(ga, gb):
                d = (ga - gb).abs()
                idx = torch.nonzero(d > 0)
                print(f"step {step} GRAD {pre}{k}: {idx.shape[0]}/{ga.numel()} max {float(d.max()):.3e} first {idx[:4].tolist()} last {idx[-2:].tolist()}")
    print(f"step {step}: {len(tr[0])} traced calls, {'no' if nd == 0 else nd} differing (first shown)")
