"""Debug helper (GPU box): time one igemm shape under the FMRI_IGEMM_DEBUG ablation flags."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "thesis-fmri-reconstruction_amd"))
import torch
from fmri_hip import ops
class G:
    def __init__(s, t): s.views = t; s.grads = {k: torch.zeros_like(v) for k, v in t.items()}; s.version = 0; s.device = torch.device("cuda:0")
def run(cin, cout, stride, N, H, kind="conv"):
    shape = (cout, cin, 5, 5) if kind == "conv" else (cin, cout, 5, 5)
    g = G({"w": torch.randn(*shape, device="cuda") * 0.05})
    L = ops.ConvLayer(g, "w", None, kind, cin, cout, 5, stride, 2, 1 if kind == "deconv" else 0)
    x = torch.randn(N, H, H, ops.pad8(cin), device="cuda").half()
    y = L.forward(x)
    fl = L._flops(N, H, H, y.shape[1], y.shape[2])
    for mode in (0, 1, 2, 3, 4, 8, 12):
        os.environ["FMRI_IGEMM_DEBUG"] = str(mode)
        for _ in range(3): L.forward(x, out=y)
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): L.forward(x, out=y)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{kind} {cin}->{cout} s{stride} N{N} H{H} mode {mode:2d}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s", flush=True)
    os.environ["FMRI_IGEMM_DEBUG"] = "0"
run(128, 256, 2, 768, 32)     # disc conv2 fwd
run(256, 128, 2, 768, 16, "deconv")   # ~ dgrad-like tconv
run(32, 128, 2, 768, 64)      # disc conv1 fwd (Cin=32)
