"""GPU box: the dense (fc) layers of the B = 256 Stage-I step, forward and data gradient, per output-tile width of the
generic igemm kernel (split-K slabs + reduce_slabs): time per call and the slab traffic it implies."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "thesis-fmri-reconstruction_amd"))
import torch
from fmri_hip import ops


class G:
    def __init__(s, t):
        s.views = t; s.grads = {k: torch.zeros_like(v) for k, v in t.items()}; s.version = 0; s.device = torch.device("cuda:0")


def timed(f, rep=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rep): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rep * 1e3


CASES = [("enc.fc", 256, 16384, 1024, (256, 64), None), ("disc.fc0", 768, 16384, 512, (256, 64), None),
         ("dec.fc", 512, 128, 16384, None, (256, 64)), ("enc.heads", 256, 1024, 256, None, None),
         ("cog.fc1", 256, 4096, 1024, None, None)]
for name, M, K, N, ip, op in CASES:
    for t_out, t_in in ((None, None), (64, 64), (32, 32)):
        g = G({"w": torch.randn(N, K, device="cuda") * 0.02})
        L = ops.DenseLayer(g, "w", None, K, N, in_perm=ip, out_perm=op)
        if t_out is not None:
            # rebuild the packed weights with the narrower tile
            L.t_out, L.t_in = min(t_out, L.t_out), min(t_in, L.t_in)
            g.packed = []
            spec_f, spec_d = L.pw_f.specs[0], L.pw_d.specs[0]
            L.pw_f = ops._single(L.w, g, spec_f, L.t_out)
            L.pw_d = ops._single(L.w, g, spec_d, L.t_in)
        x = torch.randn(M, L.kp, device="cuda").half()
        dy = torch.randn(M, L.np_, device="cuda").half()
        tf = timed(lambda: L.forward(x))
        td = timed(lambda: L.dgrad(dy))
        print(f"{name:10s} M{M} K{K} N{N} tiles fwd {L.t_out:3d} dgrad {L.t_in:3d}: fwd {tf:7.1f} us  dgrad {td:7.1f} us", flush=True)
