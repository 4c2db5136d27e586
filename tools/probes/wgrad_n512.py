"""Probe (GPU box): decoder weight gradients as two launches over 256 images each (one per cotangent block, as the step
issues them) vs one launch over 512 images."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "thesis-fmri-reconstruction_amd"))
import torch
from fmri_hip import ops
class G:
    def __init__(s, t): s.views = t; s.grads = {k: torch.zeros_like(v) for k, v in t.items()}; s.version = 0; s.device = torch.device("cuda:0")
ops._SIDE["on"] = False
def run(tag, cin, cout, stride, N, H, kind):
    shape = (cout, cin, 5, 5) if kind == "conv" else (cin, cout, 5, 5)
    g = G({"w": torch.randn(*shape, device="cuda") * 0.05})
    L = ops.ConvLayer(g, "w", None, kind, cin, cout, 5, stride, 2, 1 if kind == "deconv" else 0)
    x = torch.randn(N, H, H, ops.pad8(cin), device="cuda").half()
    y = L.forward(x); dy = torch.randn_like(y)
    h = N // 2
    one = lambda: L._wgrad(x, dy, 1.0)
    two = lambda: (L._wgrad(x[:h], dy[:h], 1.0), L._wgrad(x[h:], dy[h:], 1.0))
    res = []
    for f in (two, one):
        for _ in range(2): f()
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 10 * 1e3)
    print(f"{tag:14s} 2 x N={h}: {res[0]:7.1f} us   1 x N={N}: {res[1]:7.1f} us", flush=True)
run("dec.deconv0", 256, 256, 2, 512, 8, "deconv")
run("dec.deconv1", 256, 128, 2, 512, 16, "deconv")
run("dec.deconv2", 128, 32, 2, 512, 32, "deconv")
run("dec.conv3", 32, 3, 1, 512, 64, "conv")
