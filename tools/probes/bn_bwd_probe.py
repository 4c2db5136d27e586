"""GPU box: BatchNorm backward passes (two-stream reduce + apply, single-stream reduce + apply) on the step's largest
tensors, GB/s of their algorithmic bytes (FMRI_BN_BLOCKS varies the reduction grid)."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "thesis-fmri-reconstruction_amd"))
import torch
from fmri_hip.ops import BatchNorm
class G:
    def __init__(s, t): s.views = t; s.grads = {k: torch.zeros_like(v) for k, v in t.items()}; s.version = 0; s.device = torch.device("cuda:0"); s.bufs = {}
DEV = "cuda:0"
def run(N, H, C, two):
    g = G({"bn.weight": (torch.rand(C) + 0.5).to(DEV), "bn.bias": (torch.randn(C) * 0.3).to(DEV)})
    g.bufs = {"bn.running_mean": torch.zeros(C, device=DEV), "bn.running_var": torch.ones(C, device=DEV), "bn.num_batches_tracked": torch.zeros((), dtype=torch.int64, device=DEV)}
    bn = BatchNorm(g, "bn.", C)
    raw = torch.randn(N, H, H, C, device=DEV).half()
    _, sv = bn.forward(raw, True, 0)
    dy = torch.randn((2 if two else 1) * N, H, H, C, device=DEV).half()
    out = torch.empty_like(dy)
    f = (lambda: bn.backward2(raw, dy, sv, True, 1.0, out=out)) if two else (lambda: bn.backward(raw, dy, sv, True, 1.0, out=out))
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    el = N * H * H * C
    byts = (2 * (3 if two else 2) + 2 * (5 if two else 3)) * el     # reduce: x + dy(s); apply: x + dy(s) + dx(s)
    print(f"N={N} H={H} C={C} streams={2 if two else 1}: reduce + apply {us:7.1f} us  {byts / us / 1e3:7.1f} GB/s", flush=True)
run(768, 32, 128, True)
run(768, 16, 256, True)
run(512, 32, 128, False)
run(256, 16, 128, False)
run(256, 8, 256, False)
