"""Probe (GPU box): the conv-2 data gradient (128 <- 256 channels, 16x16 -> 32x32, igemm_win) at different batch sizes,
as one launch and as back-to-back launches over image ranges -- separates a per-launch size effect (cache footprint of
the 403 MB output at N = 1536) from a sustained-load effect (clock under MFMA load)."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "thesis-fmri-reconstruction_amd"))
import torch
from fmri_hip import ops
class G:
    def __init__(s, t): s.views = t; s.grads = {k: torch.zeros_like(v) for k, v in t.items()}; s.version = 0; s.device = torch.device("cuda:0")
cin, cout, H = 128, 256, 32
g = G({"w": torch.randn(cout, cin, 5, 5, device="cuda") * 0.05})
L = ops.ConvLayer(g, "w", None, "conv", cin, cout, 5, 2, 2)
def timeit(fn, rep=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rep): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rep
NMAX = 1536
dy_all = torch.randn(NMAX, H // 2, H // 2, cout, device="cuda").half()
out_all = torch.empty(NMAX, H, H, cin, device="cuda").half()
for N in (256, 512, 768, 1536):
    fl = 2.0 * N * H * H * cin * cout * 25 / 4
    ms = timeit(lambda: L.dgrad(dy_all[:N], H, H, out=out_all[:N]))
    print(f"one launch   N={N:5d}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s", flush=True)
for parts in (2, 3, 6):
    n = NMAX // parts
    fl = 2.0 * NMAX * H * H * cin * cout * 25 / 4
    def run():
        for i in range(parts):
            L.dgrad(dy_all[i * n:(i + 1) * n], H, H, out=out_all[i * n:(i + 1) * n])
    ms = timeit(run)
    print(f"{parts} launches of N={n:4d}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s", flush=True)
