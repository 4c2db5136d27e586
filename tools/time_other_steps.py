"""GPU box: ms/step of the other fused steps at B = 256 (BASELINE configs 3-5 shapes), eager launches."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "thesis-fmri-reconstruction_amd"))
import numpy as np, torch
from fmri_hip import lib
from fmri_hip.params import ArchConfig
from fmri_hip.steps import CognitiveStep
from fmri_hip.wae_steps import WaeStep, DualStage1Step
lib.load()
dev = torch.device("cuda:0")
B = 256
def bench(tag, fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    print(f"{tag:34s} {1e3*(time.perf_counter()-t0)/n:8.2f} ms/step  {B*n/(time.perf_counter()-t0):9.0f} samples/s", flush=True)
rs = np.random.RandomState(0)
def t(*s): return torch.from_numpy(rs.standard_normal(s).astype(np.float32)).to(dev)
for name, cfg, V in (("px64 V=4096", ArchConfig.px64(), 4096), ("px128 V=3620", ArchConfig.px128(), 3620)):
    x = torch.tanh(t(B, 3, cfg.image_size, cfg.image_size)); fm = t(B, V); e, z, et = t(B, 128), t(B, 128), t(B, 128)
    for stage in (2, 3):
        st = CognitiveStep(cfg, V, dev, stage); st.load_recipe(1, True)
        bench(f"Stage-{stage} {name}", lambda: st.step(fm, x, e, z, et))
        del st; torch.cuda.empty_cache()
cfg = ArchConfig.px64(); x = torch.tanh(t(B, 3, 64, 64)); fm = t(B, 4096); e, z, zf = t(B, 128), t(B, 128), t(B, 128)
for stage in (1, 2, 3):
    st = WaeStep(cfg, dev, stage, 4096 if stage > 1 else 0); st.load_recipe(5, False)
    bench(f"WAE Stage-{stage} px64", (lambda: st.step(x, zf)) if stage == 1 else (lambda: st.step(x, fmri=fm)))
    del st
st = DualStage1Step(cfg, dev); st.load_recipe(8, True)
bench("Dual Stage-I px64", lambda: st.step(x, e, z, zf))
