"""GPU box: why does lib.ptr() show 27 us per call under cProfile?  Times data_ptr() per call inside the API-path step."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "thesis-fmri-reconstruction_amd"))
import numpy as np, torch
import configs.models_config as mc
mc.use_px64()
from fmri_hip import lib, ops, nets
import models.vae_gan as vg
rec = []
_orig = lib.ptr
def timed_ptr(t):
    if t is None:
        return None
    t0 = time.perf_counter()
    p = t.data_ptr()
    dt = time.perf_counter() - t0
    rec.append((dt, type(t).__name__, tuple(t.shape), t.dtype, t.requires_grad, t.grad_fn is not None, t._is_view()))
    return p
for m in (lib, ops, nets):
    if hasattr(m, "_P"): m._P = timed_ptr
lib.ptr = timed_ptr
dev = "cuda:0"; B = 256
rs = np.random.RandomState(0)
x = torch.from_numpy(rs.uniform(-1, 1, (B, 3, 64, 64)).astype(np.float32)).to(dev)
model = vg.VaeGan(device=dev, z_size=128).to(dev); model.train()
for _ in range(2):
    out = model(x)
torch.cuda.synchronize(); rec.clear()
t0 = time.perf_counter(); out = model(x); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"forward: host {1e3*(t1-t0):.2f} ms, wall {1e3*(t2-t0):.2f} ms; ptr calls {len(rec)}, ptr time {1e3*sum(r[0] for r in rec):.2f} ms")
rec.sort(key=lambda r: -r[0])
for r in rec[:12]:
    print(f"{1e6*r[0]:8.1f} us", r[1:])
import collections
by = collections.defaultdict(lambda: [0, 0.0])
for r in rec:
    k = (r[1], r[4], r[5], r[6]); by[k][0] += 1; by[k][1] += r[0]
for k, v in by.items():
    print(k, v[0], f"{1e6*v[1]/v[0]:.1f} us avg")
