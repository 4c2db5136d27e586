"""Probe (GPU box): does a BatchNorm pass on one stream run BESIDE a resident weight-gradient kernel on another?
wgrad_win holds 2 x 200..208 registers per SIMD (of 512) and 70 KB of LDS per CU: a 256-thread block of a kernel with at
most 96 registers per wave still fits on the same CU, a block with more does not and waits for a free CU.
Times (a) the weight gradient of discriminator.conv.2 alone, (b) the pass alone, (c) both started together."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "thesis-fmri-reconstruction_amd"))
import torch
from fmri_hip import ops, lib
_P = lambda t: t.data_ptr()
class G:
    def __init__(s, t): s.views = t; s.grads = {k: torch.zeros_like(v) for k, v in t.items()}; s.version = 0; s.device = torch.device("cuda:0")
g = G({"w": torch.randn(256, 128, 5, 5, device="cuda") * 0.05})
L = ops.ConvLayer(g, "w", None, "conv", 128, 256, 5, 2, 2, 0)
x = torch.randn(768, 32, 32, 128, device="cuda").half()
y = L.forward(x); dy = torch.randn_like(y)
wg = lambda: L._wgrad(x, dy, 1.0)
M, C = 768 * 32 * 32, 128
t = torch.randn(M, C, device="cuda").half(); gt = torch.randn(M, C, device="cuda").half(); o = torch.empty_like(t)
sc = torch.rand(C, device="cuda") + 0.5; sh = torch.randn(C, device="cuda"); sums = torch.randn(2, C, device="cuda")
passes = {
    "bn_apply (bn_stream<0>, 52 regs)": lambda: lib.call("fmri_bn_apply", _P(t), _P(o), M, C, _P(sc), _P(sh), 1),
    "bn_bwd_apply (bn_stream<1>, 114 regs)": lambda: lib.call("fmri_bn_bwd_apply", _P(t), _P(gt), _P(o), M, C, float(M), _P(sh), _P(sc), _P(sc), _P(sh), 1, _P(sums)),
}
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def timed(fa, fb, rep=10):
    """fa on s1 and fb on s2 (either may be None), started together; ms per repetition until both are done"""
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    s1.wait_event(e0); s2.wait_event(e0)
    for _ in range(rep):
        if fa:
            with torch.cuda.stream(s1): fa()
        if fb:
            with torch.cuda.stream(s2): fb()
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rep
for _ in range(3): wg()
for f in passes.values(): f()
timed(None, wg); [timed(f, wg) for f in passes.values()]      # (per-stream allocator pools warm)
tw = timed(None, wg)
print(f"weight gradient alone: {tw*1e3:.1f} us")
for name, f in passes.items():
    ta = timed(f, None); tb = timed(f, wg)
    print(f"{name}: alone {ta*1e3:.1f} us, with the weight gradient {tb*1e3:.1f} us (sum {1e3*(ta+tw):.1f}, overlap {1e3*(ta+tw-tb):.1f})")
