"""Probe (GPU box): how long does the main stream wait for the side stream at the joins of an eager Stage-I step?
Events are recorded on the main stream right before and right after every ops.join_side(); elapsed = the wait."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "thesis-fmri-reconstruction_amd"))
import numpy as np, torch
from fmri_hip import ops, nets, steps
from fmri_hip.params import ArchConfig
B = 256
st = steps.Stage1Step(ArchConfig.px64(), "cuda:0"); st.load_recipe(0, False)
rs = np.random.RandomState(0)
x = torch.from_numpy(rs.uniform(-1, 1, (B, 3, 64, 64)).astype(np.float32)).cuda()
e, z = (torch.from_numpy(rs.standard_normal((B, 128)).astype(np.float32)).cuda() for _ in range(2))
for _ in range(5): st.step(x, e, z)
torch.cuda.synchronize()
rec = []
orig = ops.join_side
def wrapped(device=None):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    pending = len(ops._SIDE["pending"])
    a.record(); orig(device); b.record()
    rec.append((a, b, pending))
ops.join_side = wrapped; nets.join_side = wrapped; steps.ops.join_side = wrapped
N = 10
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
marks = []
t0.record()
for _ in range(N):
    s = torch.cuda.Event(enable_timing=True); s.record(); marks.append((len(rec), s))
    st.step(x, e, z)
t1.record(); torch.cuda.synchronize()
print(f"step {t0.elapsed_time(t1) / N:.3f} ms")
per = len(rec) // N
for i in range(per):
    w = [rec[k * per + i][0].elapsed_time(rec[k * per + i][1]) for k in range(N)]
    at = [marks[k][1].elapsed_time(rec[k * per + i][0]) for k in range(N)]
    print(f"join {i}: pending {rec[i][2]:3d}  reached at {np.median(at):7.3f} ms into the step, main waits {np.median(w):7.3f} ms")
