"""GPU box: dense forward / data gradient per split-K block target (ops._choose_splits aims at `target` blocks)."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "thesis-fmri-reconstruction_amd"))
import torch
from fmri_hip import ops


class G:
    def __init__(s, t):
        s.views = t; s.grads = {k: torch.zeros_like(v) for k, v in t.items()}; s.version = 0; s.device = torch.device("cuda:0")


def timed(f, rep=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rep): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rep * 1e3


CASES = [("cog.fc1", 256, 4096, 1024, None, None), ("enc.fc", 256, 16384, 1024, (256, 64), None),
         ("disc.fc0", 768, 16384, 512, (256, 64), None), ("dec.fc", 512, 128, 16384, None, (256, 64)),
         ("enc.heads", 256, 1024, 256, None, None)]
orig = ops._choose_splits
for target in (512, 256, 128, 64):
    def choose(blocks, ksteps, target=target):
        if blocks >= 192 or ksteps < 8:
            return 1
        s = max(1, min(ksteps // 4, (target + blocks - 1) // blocks))
        per = (ksteps + s - 1) // s
        return (ksteps + per - 1) // per
    ops._choose_splits = choose
    for name, M, K, N, ip, op in CASES:
        g = G({"w": torch.randn(N, K, device="cuda") * 0.02})
        L = ops.DenseLayer(g, "w", None, K, N, in_perm=ip, out_perm=op)
        x = torch.randn(M, L.kp, device="cuda").half()
        dy = torch.randn(M, L.np_, device="cuda").half()
        tf = timed(lambda: L.forward(x)); td = timed(lambda: L.dgrad(dy))
        nb = 2.0 * (M * K + K * N + M * N)
        print(f"target {target:4d} {name:10s}: fwd {tf:6.1f} us ({nb / tf / 1e3:6.0f} GB/s)  dgrad {td:6.1f} us", flush=True)
