"""Debug helper (GPU box): Stage-I step replayed from a captured HIP graph vs eager launches."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "thesis-fmri-reconstruction_amd"))
import numpy as np, torch
from fmri_hip import lib
from fmri_hip.params import ArchConfig
from fmri_hip.steps import Stage1Step
lib.load()
dev = torch.device("cuda:0")
cfg = ArchConfig.px64(); B = 256
st = Stage1Step(cfg, dev); st.load_recipe(0, False)
x = torch.from_numpy(np.random.RandomState(1234).uniform(-1, 1, (B, 3, 64, 64)).astype(np.float32)).to(dev)
nz = torch.from_numpy(np.random.RandomState(1236).standard_normal((2, B, cfg.latent_dim)).astype(np.float32)).to(dev)
for _ in range(5): st.step(x, nz[0], nz[1])
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): st.step(x, nz[0], nz[1])
torch.cuda.synchronize()
print("eager ms/step", (time.perf_counter() - t0) / 20 * 1e3, flush=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): st.step(x, nz[0], nz[1])
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    st.step(x, nz[0], nz[1])
torch.cuda.synchronize()
for _ in range(3): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): g.replay()
torch.cuda.synchronize()
print("graph ms/step", (time.perf_counter() - t0) / 20 * 1e3, flush=True)
print(st.logs())
