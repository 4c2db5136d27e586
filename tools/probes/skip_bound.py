"""GPU box: upper bound of what the fused Stage-I step could gain from the gradient unpack / optimizer / weight re-pack
chain: the hybrid step timed with those launches SKIPPED (wrong numbers, right clock).  One process, same box."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "thesis-fmri-reconstruction_amd"))
import numpy as np, torch
from fmri_hip import ops, steps, nets
from fmri_hip.params import ArchConfig
from fmri_hip.steps import Stage1Step
dev, B = "cuda:0", 256
rs = np.random.RandomState(0)
x = torch.from_numpy(rs.uniform(-1, 1, (B, 3, 64, 64)).astype(np.float32)).to(dev)
e, z = (torch.from_numpy(rs.standard_normal((B, 128)).astype(np.float32)).to(dev) for _ in range(2))
st = Stage1Step(ArchConfig.px64(), dev); st.load_recipe(0, False)
for _ in range(5): st.step(x, e, z)
run = st.capture_forward(x, e, z)
real = dict(unpack=ops.unpack_grad, repack=ops._repack_group, opt=steps._Optim.step, zero=type(st.enc.group).zero_grad)
def fake_repack(group):
    for pw in group.packed: pw.version = group.version
def fake_opt(self, flag=None, clamp=0.0, gdev=None): self.g.version += 1
def setup(skip):
    ops.unpack_grad = (lambda *a, **k: None) if "unpack" in skip else real["unpack"]
    ops._repack_group = fake_repack if "repack" in skip else real["repack"]
    steps._Optim.step = fake_opt if "opt" in skip else real["opt"]
    type(st.enc.group).zero_grad = (lambda self: None) if "zero" in skip else real["zero"]
def timed(n=60):
    for _ in range(8): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): run()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
variants = [(), ("unpack",), ("repack",), ("opt",), ("zero",), ("unpack", "repack"), ("unpack", "repack", "opt", "zero")]
for rnd in range(3):
    for v in variants:
        setup(v)
        print(f"round {rnd} skip {'+'.join(v) or 'nothing':28s} {timed():.3f} ms/step", flush=True)
