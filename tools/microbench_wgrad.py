"""Debug helper (GPU box): time the weight-gradient GEMMs of the B=256 Stage-I step per layer."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "thesis-fmri-reconstruction_amd"))
import torch
from fmri_hip import ops
class G:
    def __init__(s, t): s.views = t; s.grads = {k: torch.zeros_like(v) for k, v in t.items()}; s.version = 0; s.device = torch.device("cuda:0")
REP = int(os.environ.get("REP", "10"))
def run(tag, cin, cout, stride, N, H, kind="conv"):
    shape = (cout, cin, 5, 5) if kind == "conv" else (cin, cout, 5, 5)
    g = G({"w": torch.randn(*shape, device="cuda") * 0.05})
    L = ops.ConvLayer(g, "w", None, kind, cin, cout, 5, stride, 2, 1 if kind == "deconv" else 0)
    x = torch.randn(N, H, H, ops.pad8(cin), device="cuda").half()
    y = L.forward(x)
    fl = L._flops(N, H, H, y.shape[1], y.shape[2])
    dy = torch.randn_like(y)
    f = lambda: L._wgrad(x, dy, 1.0)   # current stream (L.wgrad would go to the side stream)
    for _ in range(2): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REP): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / REP
    print(f"{tag:18s} {kind} {cin}->{cout} s{stride} N{N} H{H} wgrad: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s (incl. unpack)", flush=True)
run("disc.conv0", 3, 32, 1, 768, 64)
run("disc.conv1", 32, 128, 2, 768, 64)
run("disc.conv2", 128, 256, 2, 768, 32)
run("disc.conv3", 256, 256, 2, 768, 16)
run("dec.deconv0", 256, 256, 2, 256, 8, "deconv")
run("dec.deconv1", 256, 128, 2, 256, 16, "deconv")
run("dec.deconv2", 128, 32, 2, 256, 32, "deconv")
run("dec.conv3", 32, 3, 1, 256, 64)
run("enc.conv0", 3, 64, 2, 256, 64)
run("enc.conv1", 64, 128, 2, 256, 32)
run("enc.conv2", 128, 256, 2, 256, 16)
