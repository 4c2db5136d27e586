"""GPU box: upper bound of what the fused Stage-I step could gain from faster dense layers (fc.0 of the three
sub-networks: M = 256..1536 rows against 16384-wide weights): the hybrid step timed with those GEMM launches SKIPPED
(wrong numbers, right clock), one process, same box."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "thesis-fmri-reconstruction_amd"))
import numpy as np, torch
from fmri_hip import ops
from fmri_hip.params import ArchConfig
from fmri_hip.steps import Stage1Step
dev, B = "cuda:0", 256
rs = np.random.RandomState(0)
x = torch.from_numpy(rs.uniform(-1, 1, (B, 3, 64, 64)).astype(np.float32)).to(dev)
e, z = (torch.from_numpy(rs.standard_normal((B, 128)).astype(np.float32)).to(dev) for _ in range(2))
st = Stage1Step(ArchConfig.px64(), dev); st.load_recipe(0, False)
for _ in range(5): st.step(x, e, z)
real_gemm, real_wg = ops.DenseLayer._gemm, ops.DenseLayer._wgrad
cache = {}
def fake_gemm(self, x, pw, M, Ci, Co, CoStore, tile, bias, act, want16, want32):
    k = (id(self), id(pw), M, want16, want32)
    if k not in cache: cache[k] = real_gemm(self, x, pw, M, Ci, Co, CoStore, tile, bias, act, want16, want32)
    return cache[k]
def fake_wg(self, x, dy, scale):
    k = (id(self), "w")
    if k in cache:
        packed, ldo = cache[k]
        ops.emit_grad(self.group, packed, self.wg, self.gspec, ldo, 1.0 / scale)
        return
    M = x.shape[0]
    packed, ldo = ops.run_wgrad(dy, x, M, 1, 1, self.np_, 1, 1, self.kp, 1, 1, 0, hold=ops._hold_of(self))
    cache[k] = (packed, ldo)
    ops.emit_grad(self.group, packed, self.wg, self.gspec, ldo, 1.0 / scale)
def timed(run, n=60):
    for _ in range(8): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): run()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
run = lambda: st.step(x, e, z)
for rnd in range(3):
    for name, g, w in (("nothing", real_gemm, real_wg), ("dense fwd+dgrad", fake_gemm, real_wg),
                       ("dense wgrad", real_gemm, fake_wg), ("all dense GEMMs", fake_gemm, fake_wg)):
        ops.DenseLayer._gemm, ops.DenseLayer._wgrad = g, w
        print(f"round {rnd} skip {name:18s} {timed(run):.3f} ms/step (eager)", flush=True)
