"""gfx950 store-data hazard under contention: a data-gradient kernel whose output stores carry an SGPR offset
(igemm_tc5w / igemm_c5w) is launched ITER times on the main stream while a weight-gradient kernel runs in a loop on a
second stream; every output is compared bit for bit with the result of the same launch on an idle GPU.

    python tools/probes/store_hazard_stress.py [iters]        (FMRI_LIB_PATH selects the library build)
"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
for p in (ROOT, os.path.join(ROOT, "thesis-fmri-reconstruction_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from fmri_hip import lib, ops  # noqa: E402
from fmri_hip.ops import ConvLayer  # noqa: E402

DEV = "cuda:0"
ITER = int(sys.argv[1]) if len(sys.argv) > 1 else 300


class G:
    def __init__(self, tensors):
        self.views = {k: v.to(DEV) for k, v in tensors.items()}
        self.grads = {k: torch.zeros_like(v) for k, v in self.views.items()}
        self.version = 0
        self.device = torch.device(DEV)


def layer(kind, cin, cout):
    shape = (cout, cin, 5, 5) if kind == "conv" else (cin, cout, 5, 5)
    g = G({"w": (torch.randn(*shape) * 0.05).half().float()})
    return ConvLayer(g, "w", None, kind, cin, cout, 5, 2, 2, 1 if kind == "deconv" else 0)


# (label, kind, cin, cout, H of the layer input, N, what): conv dgrad = transposed conv (igemm_tc5w), deconv dgrad and conv
# forward = stride-2 conv (igemm_c5w)
CASES = [
    ("disc.conv3 dgrad B=8   tc5w<8,0,solo>", "conv", 256, 256, 16, 40, "dgrad"),
    ("disc.conv3 dgrad B=256 tc5w<8,0>", "conv", 256, 256, 16, 1280, "dgrad"),
    ("disc.conv2 dgrad B=256 tc5w<16,0>", "conv", 128, 256, 32, 1536, "dgrad"),
    ("dec.conv0 fwd   B=512 tc5w<16,1>", "deconv", 256, 256, 8, 512, "fwd_stats"),
    ("disc.conv2 fwd  B=768 c5w<16,1>", "conv", 128, 256, 32, 768, "fwd_stats"),
    ("disc.conv3 fwd  B=768 c5w<8,1>", "conv", 256, 256, 16, 768, "fwd_stats"),
    ("dec.conv1 dgrad B=768 c5w<16,0>", "deconv", 256, 128, 16, 768, "dgrad"),
]
torch.manual_seed(0)
side = torch.cuda.Stream()
# the contending kernel: discriminator conv.3 weight gradient at 3B = 24 (what ran beside the failing launch) and at 768
wl = layer("conv", 256, 256)
wx = torch.randn(192, 16, 16, 256, device=DEV).half()
wdy = torch.randn(192, 8, 8, 256, device=DEV).half()
ops._SIDE["on"] = False
print("library:", lib.LIB_PATH)
for label, kind, cin, cout, H, N, what in CASES:
    L = layer(kind, cin, cout)
    Ho, Wo = L.out_hw(H, H)
    if what == "dgrad":
        inp = (torch.randn(N, Ho, Wo, L.coutp, device=DEV) * 0.5).half()
        run = lambda out: L.dgrad(inp, H, H, out=out)
        oshape = (N, H, H, L.cinp)
    else:
        inp = torch.randn(N, H, H, L.cinp, device=DEV).half()
        run = lambda out: L.forward(inp, out=out, bn_groups=1)
        oshape = (N, Ho, Wo, L.coutp)
    ref = torch.empty(oshape, dtype=torch.float16, device=DEV)
    run(ref)
    torch.cuda.synchronize()
    ring = [torch.empty_like(ref) for _ in range(4)]
    bad_iters = bad_elems = 0
    first = None
    for it in range(ITER):
        with torch.cuda.stream(side):
            for _ in range(2):
                wl._wgrad(wx, wdy, 1.0)
        out = ring[it % 4]
        run(out)
        if it % 4 == 3 or it == ITER - 1:
            torch.cuda.synchronize()
            for j, o in enumerate(ring[:(it % 4) + 1]):
                if not torch.equal(o, ref):
                    d = torch.nonzero(o != ref)
                    bad_iters += 1
                    bad_elems += d.shape[0]
                    if first is None:
                        first = d[:4].tolist()
    print(f"{label}: {bad_iters}/{ITER} launches differ ({bad_elems} elements){'' if first is None else ' first ' + str(first)}",
          flush=True)
