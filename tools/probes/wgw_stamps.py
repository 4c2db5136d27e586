"""Diagnostic (GPU box): wave cycles of wgrad_win in the kernel and the clock the chip holds (s_memtime / s_memrealtime,
FMRI_STAMP in csrc/wgrad_win.hip), for the build tools/probes/libfmri_$VARIANT.so
(tools/probes/build_variant.sh <name> wgrad_win -DFMRI_STAMP=1 [-DWGW_ABL=n])."""
import ctypes, os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "thesis-fmri-reconstruction_amd"))
import torch
from fmri_hip import lib
lib.LIB_PATH = os.path.join(ROOT, "tools", "probes", "libfmri_%s.so" % os.environ.get("VARIANT", "stamp"))
from fmri_hip import ops
L = lib.load()
L.fmri_debug_wgw_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.fmri_debug_wgw_stamps.restype = ctypes.c_int


class G:
    def __init__(s, t):
        s.views = t; s.grads = {k: torch.zeros_like(v) for k, v in t.items()}; s.version = 0; s.device = torch.device("cuda:0")


def run(cin, cout, N, H, reps=100):
    g = G({"w": torch.randn(cout, cin, 5, 5, device="cuda") * 0.05})
    layer = ops.ConvLayer(g, "w", None, "conv", cin, cout, 5, 2, 2)
    x = torch.randn(N, H, H, cin, device="cuda").half()
    y = layer.forward(x)
    dy = torch.randn_like(y)
    out = (ctypes.c_ulonglong * 8)()
    for _ in range(reps):
        layer._wgrad(x, dy, 1.0)
    torch.cuda.synchronize()
    L.fmri_debug_wgw_stamps(out, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        layer._wgrad(x, dy, 1.0)
    e1.record(); torch.cuda.synchronize()
    L.fmri_debug_wgw_stamps(out, 1)
    c9, c6, c4, waves, kc, kr = int(out[0]), int(out[1]), int(out[2]), int(out[5]), int(out[6]), int(out[7])
    ms = e0.elapsed_time(e1) / 20
    fl = 2.0 * N * (H // 2) ** 2 * cin * cout * 25
    print(f"wgrad of conv {cin}->{cout} N={N} {H}px: {ms*1e3:7.1f} us incl. unpack ({fl/ms/1e9:6.1f} TF/s, stamped build) clock {kc/max(kr,1)*0.1:.2f} GHz "
          f"wave cycles per launch: all {kc/20/1e6:8.2f} M  (9-shift planes {c9/20/1e6:.2f} M, 6-shift {c6/20/1e6:.2f} M, 4-shift {c4/20/1e6:.2f} M)  "
          f"waves {waves//20}", flush=True)


run(128, 256, 768, 32)
run(256, 256, 768, 16)
