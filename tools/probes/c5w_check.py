"""Debug helper (GPU box): igemm_c5w against torch's convolution on one geometry, per-image / per-row error map."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "thesis-fmri-reconstruction_amd"))
import torch, torch.nn.functional as F
from fmri_hip import ops
class G:
    def __init__(s, t): s.views = t; s.grads = {k: torch.zeros_like(v) for k, v in t.items()}; s.version = 0; s.device = torch.device("cuda:0")
cin, cout, H, W, N = [int(v) for v in sys.argv[1:6]]
torch.manual_seed(0)
w = (torch.randn(cout, cin, 5, 5) * 0.05).half().float()
x = torch.randn(N, cin, H, W).half().float()
g = G({"w": w.cuda()})
L = ops.ConvLayer(g, "w", None, "conv", cin, cout, 5, 2, 2)
x16 = x.cuda().permute(0, 2, 3, 1).contiguous().half()
y = torch.full((N, (H + 1) // 2, (W + 1) // 2, cout), float("nan"), device="cuda", dtype=torch.half)
L.forward(x16, out=y)
torch.cuda.synchronize()
ref = F.conv2d(x.cuda(), w.cuda(), None, 2, 2).permute(0, 2, 3, 1)
err = (y.float() - ref).abs()
bad = ~(err < 2e-2 * (1 + ref.abs()))
print("label", ops.igemm_route(N, H, W, cin, y.shape[1], y.shape[2], cout, cout, 5, 2, 2, ops.MODE_CONV, ops.ACT_NONE, False, 1, 128, L.pw_f.buf.numel()))
print("bad elements", int(bad.sum()), "of", bad.numel(), "nan", int(torch.isnan(y).sum()))
for n in range(N):
    rows = bad[n].any(dim=2).cpu()
    if rows.any():
        print("image", n, "bad pixel map:\n" + "\n".join("".join("X" if v else "." for v in r) for r in rows.tolist()))
        chans = bad[n].any(dim=0).any(dim=0).cpu().nonzero().flatten().tolist()
        print("  bad channels:", chans[:8], "...", chans[-4:], len(chans))
