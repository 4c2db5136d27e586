"""Debug helper (GPU box): rounding-level error of a stride-2 convolution / its data gradient against an fp64 reference of the
same fp16 operands: mean and max of |got - ref| / (fp16 ulp of ref), and the mean SIGNED error (a bias shows here)."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "thesis-fmri-reconstruction_amd"))
import torch, torch.nn.functional as F
from fmri_hip import ops
class G:
    def __init__(s, t): s.views = t; s.grads = {k: torch.zeros_like(v) for k, v in t.items()}; s.version = 0; s.device = torch.device("cuda:0")
def ulp16(v):
    e = torch.floor(torch.log2(v.abs().clamp_min(6.1e-5)))
    return torch.pow(2.0, e - 10)
def run(cin, cout, H, N, what):
    torch.manual_seed(1)
    w = (torch.randn(cout, cin, 5, 5) * 0.05).half()
    g = G({"w": w.float().cuda()})
    L = ops.ConvLayer(g, "w", None, "conv", cin, cout, 5, 2, 2)
    if what == "fwd":
        x = torch.randn(N, H, H, cin).half().cuda()
        y = L.forward(x)
        ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double().cuda(), None, 2, 2).permute(0, 2, 3, 1)
        lab = ops.igemm_route(N, H, H, cin, H // 2, H // 2, cout, cout, 5, 2, 2, ops.MODE_CONV, ops.ACT_NONE, False, 1, 128, cout * 25 * cin)
    else:
        dy = torch.randn(N, H // 2, H // 2, cout).half().cuda()
        y = L.dgrad(dy, H, H)
        ref = F.conv_transpose2d(dy.double().permute(0, 3, 1, 2), w.double().cuda(), None, 2, 2, output_padding=1).permute(0, 2, 3, 1)
        lab = "dgrad"
    y = y[..., :ref.shape[-1]].double()
    u = ulp16(ref)
    e = (y - ref) / u
    print(f"{what} {cin}->{cout} H{H} N{N} [{lab}]: mean |err| {e.abs().mean():.4f} ulp  max {e.abs().max():.2f} ulp  mean signed {e.mean():+.5f} ulp  "
          f"rel-norm {((y - ref).norm() / ref.norm()):.3e}", flush=True)
run(128, 256, 32, 16, "fwd")
run(32, 128, 64, 8, "fwd")
run(64, 128, 32, 8, "fwd")
run(256, 256, 16, 16, "fwd")
run(128, 256, 32, 16, "dgrad")
run(256, 256, 16, 16, "dgrad")
run(32, 128, 64, 8, "dgrad")
