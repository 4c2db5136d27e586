"""GPU box: host time to ISSUE one fused Stage-I step (no synchronisation inside the loop) vs its GPU time, eager and
hybrid (recorded forward): how far the Python side is from becoming the bound."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "thesis-fmri-reconstruction_amd"))
import numpy as np, torch
from fmri_hip.params import ArchConfig
from fmri_hip.steps import Stage1Step
dev, B = "cuda:0", 256
rs = np.random.RandomState(0)
x = torch.from_numpy(rs.uniform(-1, 1, (B, 3, 64, 64)).astype(np.float32)).to(dev)
e, z = (torch.from_numpy(rs.standard_normal((B, 128)).astype(np.float32)).to(dev) for _ in range(2))
st = Stage1Step(ArchConfig.px64(), dev); st.load_recipe(0, False)
for _ in range(5): st.step(x, e, z)
run = st.capture_forward(x, e, z)
for name, fn in (("eager", lambda: st.step(x, e, z)), ("hybrid", run)):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    host = []
    t0 = time.perf_counter()
    for i in range(30):
        h0 = time.perf_counter(); fn(); host.append(time.perf_counter() - h0)
        if i % 3 == 2: torch.cuda.synchronize()          # keep the queue short: issue time, not back-pressure
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 30
    host.sort()
    print(f"{name}: host issue median {1e3 * host[len(host) // 2]:.2f} ms/step (min {1e3 * host[0]:.2f}), wall {1e3 * wall:.2f} ms/step")
