"""Diagnostic (GPU box): where a wave of igemm_c5 spends its cycles inside a K-step (s_memtime stamps, see FMRI_STAMP in
csrc/igemm_c5.hip).  Uses tools/probes/libfmri_stamp.so = the library with igemm_c5.hip compiled with -DFMRI_STAMP
(built in the dev container: tools/probes/build_variant.sh stamp igemm_c5 -DFMRI_STAMP).  The stamp
build's fences forbid overlaps the real kernel has: read the SHARES, not the run time."""
import ctypes, os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "thesis-fmri-reconstruction_amd"))
import torch
from fmri_hip import lib
lib.LIB_PATH = os.path.join(ROOT, "tools", "probes", "libfmri_stamp.so")
from fmri_hip import ops
L = lib.load()
L.fmri_debug_c5_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.fmri_debug_c5_stamps.restype = ctypes.c_int


class G:
    def __init__(s, t):
        s.views = t; s.grads = {k: torch.zeros_like(v) for k, v in t.items()}; s.version = 0; s.device = torch.device("cuda:0")


def run(cin, cout, N, H, stats):
    g = G({"w": torch.randn(cout, cin, 5, 5, device="cuda") * 0.05})
    layer = ops.ConvLayer(g, "w", None, "conv", cin, cout, 5, 2, 2)
    x = torch.randn(N, H, H, cin, device="cuda").half()
    out = (ctypes.c_ulonglong * 8)()
    for _ in range(2):
        layer.forward(x, bn_groups=1 if stats else 0)
    torch.cuda.synchronize()
    L.fmri_debug_c5_stamps(out, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        layer.forward(x, bn_groups=1 if stats else 0)
    e1.record(); torch.cuda.synchronize()
    L.fmri_debug_c5_stamps(out, 1)
    sync, dma, comp, epi, steps, waves = [int(v) for v in out[:6]]
    tot = sync + dma + comp + epi
    ms = e0.elapsed_time(e1) / 5
    fl = 2.0 * N * (H // 2) ** 2 * cin * cout * 25
    print(f"conv {cin}->{cout} N={N} {H}px stats={stats}: {ms*1e3:7.1f} us ({fl/ms/1e9:6.1f} TF/s, stamped build)  per step per wave: "
          f"sync {sync/steps:7.1f}  dma-issue {dma/steps:7.1f}  reads+mfma {comp/steps:7.1f}  epilogue/step {epi/steps:6.1f} cycles "
          f"| shares sync {sync/tot:.2f} dma {dma/tot:.2f} comp {comp/tot:.2f} epi {epi/tot:.2f}", flush=True)


run(128, 256, 768, 32, False)
run(128, 256, 768, 32, True)
run(256, 256, 768, 16, True)
run(32, 128, 768, 64, True)
run(64, 128, 256, 32, True)
