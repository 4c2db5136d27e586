"""Diagnostic (GPU box): where a wave of igemm_c5w spends its cycles inside a K-step, and the clock the chip holds
(s_memtime / s_memrealtime stamps, FMRI_STAMP in csrc/igemm_c5w.hip).  Uses tools/probes/libfmri_stamp.so
(tools/probes/build_variant.sh stamp igemm_c5w -DFMRI_STAMP=2; VARIANT=<name> picks another build).  The stamp build's fences forbid overlaps the real kernel has: read the
SHARES, not the run time."""
import ctypes, os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "thesis-fmri-reconstruction_amd"))
os.environ.setdefault("FMRI_C5W", "all")
import torch
from fmri_hip import lib
lib.LIB_PATH = os.path.join(ROOT, "tools", "probes", "libfmri_%s.so" % os.environ.get("VARIANT", "stamp"))
from fmri_hip import ops
L = lib.load()
L.fmri_debug_c5w_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.fmri_debug_c5w_stamps.restype = ctypes.c_int
ZERO = os.environ.get("ZERO") == "1"


class G:
    def __init__(s, t):
        s.views = t; s.grads = {k: torch.zeros_like(v) for k, v in t.items()}; s.version = 0; s.device = torch.device("cuda:0")


def run(cin, cout, N, H, stats, reps=200):
    g = G({"w": torch.randn(cout, cin, 5, 5, device="cuda") * (0.0 if ZERO else 0.05)})
    layer = ops.ConvLayer(g, "w", None, "conv", cin, cout, 5, 2, 2)
    x = torch.randn(N, H, H, cin, device="cuda").half()
    if ZERO: x.zero_()
    out = (ctypes.c_ulonglong * 8)()
    for _ in range(reps):
        layer.forward(x, bn_groups=1 if stats else 0)
    torch.cuda.synchronize()
    L.fmri_debug_c5w_stamps(out, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        layer.forward(x, bn_groups=1 if stats else 0)
    e1.record(); torch.cuda.synchronize()
    L.fmri_debug_c5w_stamps(out, 1)
    sync, pend, first, epi, steps, waves, kc, kr = [int(v) for v in out[:8]]
    tot = max(sync + pend + first + epi, 1); steps = max(steps, 1)
    ms = e0.elapsed_time(e1) / 20
    fl = 2.0 * N * (H // 2) ** 2 * cin * cout * 25
    print(f"conv {cin}->{cout} N={N} {H}px stats={stats} zero={ZERO}: {ms*1e3:7.1f} us ({fl/ms/1e9:6.1f} TF/s, stamped build) clock {kc/kr*0.1:.2f} GHz "
          f"wave cycles/kernel {kc/waves:9.0f}  per step per wave: sync {sync/steps:7.1f}  pending phase {pend/steps:7.1f}  first phase {first/steps:7.1f}  "
          f"epilogue/step {epi/steps:6.1f} cycles | shares sync {sync/tot:.2f} pend {pend/tot:.2f} first {first/tot:.2f} epi {epi/tot:.2f}", flush=True)


run(128, 256, 768, 32, False)
if os.environ.get("ALL"):
    run(128, 256, 768, 32, True)
    run(32, 128, 768, 64, True)
