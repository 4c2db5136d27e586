"""Debug helper (GPU box): transposed-convolution forward against torch on one geometry, per-class / per-position error map."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "thesis-fmri-reconstruction_amd"))
import torch, torch.nn.functional as F
from fmri_hip import ops
class G:
    def __init__(s, t): s.views = t; s.grads = {k: torch.zeros_like(v) for k, v in t.items()}; s.version = 0; s.device = torch.device("cuda:0")
cin, cout, H, op, N = [int(v) for v in sys.argv[1:6]]
stats = len(sys.argv) > 6
torch.manual_seed(0)
w = (torch.randn(cin, cout, 5, 5) * 0.05).half().float()
x = torch.randn(N, cin, H, H).half().float()
g = G({"w": w.cuda()})
L = ops.ConvLayer(g, "w", None, "deconv", cin, cout, 5, 2, 2, op)
x16 = x.cuda().permute(0, 2, 3, 1).contiguous().half()
Ho = 2 * H - 1 + op
y = torch.full((N, Ho, Ho, cout), float("nan"), device="cuda", dtype=torch.half)
L.forward(x16, out=y, bn_groups=1 if stats else 0)
torch.cuda.synchronize()
ref = F.conv_transpose2d(x.cuda(), w.cuda(), None, 2, 2, output_padding=op).permute(0, 2, 3, 1)
err = (y.float() - ref).abs()
bad = ~(err < 2e-2 * (1 + ref.abs()))
print("bad elements", int(bad.sum()), "of", bad.numel(), "nan", int(torch.isnan(y).sum()))
for n in range(N):
    for cy in range(2):
        for cx in range(2):
            b = bad[n, cy::2, cx::2]
            if b.any():
                rows = sorted(set(b.any(dim=2).nonzero()[:, 0].tolist())); cols = sorted(set(b.any(dim=2).nonzero()[:, 1].tolist()))
                ch = b.any(dim=0).any(dim=0).nonzero().flatten().tolist()
                print(f"image {n} class ({cy},{cx}): rows {rows} cols {cols[:6]}..{len(cols)} channels {ch[:8]}..{len(ch)}")
if os.environ.get("DUMP") and bad.any():
    # raw bits of the first wrong elements beside the expected ones (a register overwritten before the store read it shows up
    # as the halves of an fp32 number)
    import struct
    idx = bad.nonzero()[: int(os.environ["DUMP"])]
    for n, yy, xx, c in idx.tolist():
        c0 = c & ~7
        got = y[n, yy, xx, c0:c0 + 8].view(torch.int16).tolist()
        exp = ref[n, yy, xx, c0:c0 + 8].half().view(torch.int16).tolist()
        f = lambda v: " ".join(f"{u & 0xffff:04x}" for u in v)
        lo = struct.unpack("<I", struct.pack("<f", ref[n, yy, xx, c0 + 2].half().float().item()))[0]
        print(f"n{n} y{yy} x{xx} c{c0}: got {f(got)} | exp {f(exp)} | f32(ch2) {lo:08x}")
