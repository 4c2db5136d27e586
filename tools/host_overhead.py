"""Debug helper (GPU box): host time to ISSUE one Stage-I step (Python + ctypes + torch allocator) vs its GPU time."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "thesis-fmri-reconstruction_amd"))
import numpy as np, torch
from fmri_hip import lib
from fmri_hip.params import ArchConfig
from fmri_hip.steps import Stage1Step
lib.load()
dev = torch.device("cuda:0")
cfg = ArchConfig.px64(); B = int(os.environ.get("B", "256"))
st = Stage1Step(cfg, dev); st.load_recipe(0, False)
x = torch.from_numpy(np.random.RandomState(1234).uniform(-1, 1, (B, 3, 64, 64)).astype(np.float32)).to(dev)
nz = torch.from_numpy(np.random.RandomState(1236).standard_normal((2, B, cfg.latent_dim)).astype(np.float32)).to(dev)
for _ in range(5): st.step(x, nz[0], nz[1])
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): st.step(x, nz[0], nz[1])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"B={B}: host issue {1e3*(t1-t0)/20:.2f} ms/step, total {1e3*(t2-t0)/20:.2f} ms/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5): st.step(x, nz[0], nz[1])
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
