"""GPU box: is the hybrid launch mode (recorded forward + eagerly issued two-stream backward) bound by the host?  Times the
Python side of N steps (no synchronisation inside) against the wall time of the same N steps."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "thesis-fmri-reconstruction_amd"))
import numpy as np, torch
from fmri_hip.params import ArchConfig
from fmri_hip.steps import Stage1Step
dev = "cuda:0"; B = 256
rs = np.random.RandomState(0)
x = torch.from_numpy(rs.uniform(-1, 1, (B, 3, 64, 64)).astype(np.float32)).to(dev)
e, z = (torch.from_numpy(rs.standard_normal((B, 128)).astype(np.float32)).to(dev) for _ in range(2))
st = Stage1Step(ArchConfig.px64(), dev); st.load_recipe(0, False)
for _ in range(3): st.step(x, e, z)
run = st.capture_forward(x, e, z)
for _ in range(20): run()
torch.cuda.synchronize()
N = 50
t0 = time.perf_counter()
for _ in range(N): run()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"hybrid: host issue {1e3 * (t1 - t0) / N:.2f} ms/step, wall {1e3 * (t2 - t0) / N:.2f} ms/step, GPU still busy after the last issue: {1e3 * (t2 - t1):.2f} ms")
g = st.capture(x, e, z)
for _ in range(10): g()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N): g()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"one-stream graph: wall {1e3 * (t2 - t0) / N:.2f} ms/step")
