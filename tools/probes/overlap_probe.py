"""Probe (GPU box): do an HBM-bound kernel (BatchNorm backward on a 3B x 32 x 32 x 128 map) and an MFMA-bound kernel
(weight gradient of conv-2) overlap when issued on two HIP streams?  alone / alone / together."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "thesis-fmri-reconstruction_amd"))
import torch
from fmri_hip import ops
class G:
    def __init__(s, t): s.views = t; s.grads = {k: torch.zeros_like(v) for k, v in t.items()}; s.version = 0; s.device = torch.device("cuda:0")
ops._SIDE["on"] = False
g = G({"w": torch.randn(256, 128, 5, 5, device="cuda") * 0.05, "bn.weight": torch.ones(128, device="cuda"), "bn.bias": torch.zeros(128, device="cuda"),
       "bn.running_mean": torch.zeros(128, device="cuda"), "bn.running_var": torch.ones(128, device="cuda"), "bn.num_batches_tracked": torch.zeros((), dtype=torch.long, device="cuda")})
g.bufs = {k: g.views[k] for k in ("bn.running_mean", "bn.running_var", "bn.num_batches_tracked")}
L = ops.ConvLayer(g, "w", None, "conv", 128, 256, 5, 2, 2)
bn = ops.BatchNorm(g, "bn.", 128)
N = 768
x = torch.randn(N, 32, 32, 128, device="cuda").half()
y = L.forward(x); dy = torch.randn_like(y)
raw = torch.randn(N, 32, 32, 128, device="cuda").half()
a, sv = bn.forward(raw, relu=True, updates=0)
da = torch.randn_like(raw)
outb = torch.empty_like(da)
def gemm(): L._wgrad(x, dy, 1.0)
def hbm(): bn.backward(raw, da, sv, True, None, out=outb)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def timed(fa, fb, rep=20):
    for f, s in ((fa, s1), (fb, s2)):
        if f:
            with torch.cuda.stream(s):
                f(); f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(rep):
        if fa:
            with torch.cuda.stream(s1): fa()
        if fb:
            with torch.cuda.stream(s2): fb()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / rep * 1e3
tg, th, tb = timed(gemm, None), timed(None, hbm), timed(gemm, hbm)
print(f"wgrad alone {tg:.3f} ms | BN backward alone {th:.3f} ms | both on two streams {tb:.3f} ms (sum {tg+th:.3f}, max {max(tg,th):.3f})")
def gemm2(): L.dgrad(dy, 32, 32)
td = timed(gemm2, None); tdd = timed(gemm2, gemm)
print(f"dgrad alone {td:.3f} ms | dgrad + wgrad on two streams {tdd:.3f} ms (sum {td+tg:.3f})")
tdh = timed(gemm2, hbm)
print(f"dgrad + BN backward on two streams {tdh:.3f} ms (sum {td+th:.3f}, max {max(td,th):.3f})")
