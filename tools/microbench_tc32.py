"""Debug helper (GPU box): discriminator.conv.1 data gradient (128 -> 32 transposed conv, csrc/igemm_tc32.hip) vs batch."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "thesis-fmri-reconstruction_amd"))
import torch
from fmri_hip import ops
class G:
    def __init__(s, t): s.views = t; s.grads = {k: torch.zeros_like(v) for k, v in t.items()}; s.version = 0; s.device = torch.device("cuda:0")
g = G({"w": torch.randn(128, 32, 5, 5, device="cuda") * 0.05})
L = ops.ConvLayer(g, "w", None, "conv", 32, 128, 5, 2, 2)
H = 64
for N in (64, 128, 256, 512, 768, 1536):
    dy = torch.randn(N, 32, 32, 128, device="cuda").half()
    x = torch.empty(N, H, H, 32, device="cuda").half()
    y0 = torch.relu(torch.randn(N, H, H, 32, device="cuda")).half()
    masked = os.environ.get("MASK") == "1"
    f = (lambda: L.dgrad(dy, H, H, out=x, relu_y=y0)) if masked else (lambda: L.dgrad(dy, H, H, out=x))
    for _ in range(3): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    fl = L._flops(N, H, H, 32, 32)
    print(f"N={N:5d}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s  {ms*1e3/N:6.3f} us/image", flush=True)
