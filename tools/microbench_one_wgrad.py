"""Debug helper (GPU box): run ONE conv weight gradient a few times (for rocprofv3 --pmc passes)."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "thesis-fmri-reconstruction_amd"))
import torch
from fmri_hip import ops
class G:
    def __init__(s, t): s.views = t; s.grads = {k: torch.zeros_like(v) for k, v in t.items()}; s.version = 0; s.device = torch.device("cuda:0")
cin, cout, N, H = [int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (128, 256, 768, 32))]
g = G({"w": torch.randn(cout, cin, 5, 5, device="cuda") * 0.05})
L = ops.ConvLayer(g, "w", None, "conv", cin, cout, 5, 2, 2)
x = torch.randn(N, H, H, ops.pad8(cin), device="cuda").half()
y = L.forward(x); dy = torch.randn_like(y)
for _ in range(3): L._wgrad(x, dy, 1.0)
torch.cuda.synchronize()
