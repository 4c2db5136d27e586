"""Debug helper (GPU box): time representative conv layers (forward / data-gradient) of the B=256 Stage-I step."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "thesis-fmri-reconstruction_amd"))
import torch
from fmri_hip import ops
class G:
    def __init__(s, t): s.views = t; s.grads = {k: torch.zeros_like(v) for k, v in t.items()}; s.version = 0; s.device = torch.device("cuda:0")
REP = int(os.environ.get("REP", "10"))
ZERO = os.environ.get("ZERO") == "1"      # all-zero operands: the clock the chip holds then is its unloaded one
ONLY = os.environ.get("ONLY")
def run(tag, cin, cout, stride, N, H, kind="conv", what="fwd"):
    if ONLY and ONLY not in tag: return
    shape = (cout, cin, 5, 5) if kind == "conv" else (cin, cout, 5, 5)
    g = G({"w": torch.randn(*shape, device="cuda") * (0.0 if ZERO else 0.05)})
    L = ops.ConvLayer(g, "w", None, kind, cin, cout, 5, stride, 2, 1 if kind == "deconv" else 0)
    x = torch.randn(N, H, H, ops.pad8(cin), device="cuda").half()
    if ZERO: x.zero_()
    y = L.forward(x)
    fl = L._flops(N, H, H, y.shape[1], y.shape[2])
    dy = torch.zeros_like(y) if ZERO else torch.randn_like(y)
    f = (lambda: L.forward(x, out=y)) if what == "fwd" else (lambda: L.dgrad(dy, H, H, out=x))
    for _ in range(2): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REP): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / REP
    print(f"{tag:22s} {kind} {cin}->{cout} s{stride} N{N} H{H} {what}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s", flush=True)
run("disc.conv2 dgrad", 128, 256, 2, 1536, 32, "conv", "dgrad")
run("disc.conv3 dgrad", 256, 256, 2, 1536, 16, "conv", "dgrad")
run("disc.conv1 dgrad", 32, 128, 2, 1536, 64, "conv", "dgrad")
run("dec.deconv1 fwd", 256, 128, 2, 512, 16, "deconv", "fwd")
run("dec.deconv2 fwd", 128, 32, 2, 512, 32, "deconv", "fwd")
run("disc.conv0 fwd", 3, 32, 1, 768, 64, "conv", "fwd")
run("disc.conv2 fwd", 128, 256, 2, 768, 32, "conv", "fwd")
run("disc.conv0 dgrad", 3, 32, 1, 1536, 64, "conv", "dgrad")
run("dec.conv3 fwd", 32, 3, 1, 512, 64, "conv", "fwd")
run("dec.conv3 dgrad", 32, 3, 1, 512, 64, "conv", "dgrad")
run("dec.deconv0 fwd", 256, 256, 2, 512, 8, "deconv", "fwd")
run("dec.deconv1 dgrad", 256, 128, 2, 512, 16, "deconv", "dgrad")
run("disc.conv3 fwd", 256, 256, 2, 768, 16, "conv", "fwd")
run("enc.conv2 fwd", 128, 256, 2, 256, 16, "conv", "fwd")
run("dec.deconv0 dgrad", 256, 256, 2, 768, 8, "deconv", "dgrad")
run("disc.conv1 fwd", 32, 128, 2, 768, 64, "conv", "fwd")
run("dec.deconv2 dgrad", 128, 32, 2, 768, 32, "deconv", "dgrad")
run("enc.conv1 fwd", 64, 128, 2, 256, 32, "conv", "fwd")
run("enc.conv2 dgrad", 128, 256, 2, 256, 16, "conv", "dgrad")
run("enc.conv1 dgrad", 64, 128, 2, 256, 32, "conv", "dgrad")
