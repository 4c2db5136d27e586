"""Debug helper (GPU box): HIP-event timing of the segments of the hybrid Stage-I step (recorded forward | eager
two-stream backward | encoder update), and of the one-stream eager step for comparison."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "thesis-fmri-reconstruction_amd"))
import numpy as np, torch
from fmri_hip import lib, ops
from fmri_hip.nets import refresh_net
from fmri_hip.params import ArchConfig
from fmri_hip.steps import Stage1Step
lib.load()
dev = torch.device("cuda:0")
cfg = ArchConfig.px64(); B = 256
st = Stage1Step(cfg, dev); st.load_recipe(0, False)
x = torch.from_numpy(np.random.RandomState(1234).uniform(-1, 1, (B, 3, 64, 64)).astype(np.float32)).to(dev)
nz = torch.from_numpy(np.random.RandomState(1236).standard_normal((2, B, cfg.latent_dim)).astype(np.float32)).to(dev)
for _ in range(5): st.step(x, nz[0], nz[1])
run = st.capture_forward(x, nz[0], nz[1])
for _ in range(5): run()
torch.cuda.synchronize()
graph = st._fwd_graph
nets = (st.enc, st.dec, st.dis)
def ev(): e = torch.cuda.Event(enable_timing=True); e.record(); return e
acc = np.zeros(4)
R = 30
for _ in range(R):
    for n in nets: refresh_net(n)
    e0 = ev(); graph.replay(); e1 = ev()
    st.backward(early_apply=True); e2 = ev()
    st.apply(); e3 = ev()
    torch.cuda.synchronize()
    acc += [e0.elapsed_time(e1), e1.elapsed_time(e2), e2.elapsed_time(e3), e0.elapsed_time(e3)]
print("hybrid: forward+gate %.3f ms | backward (two streams, early updates) %.3f | encoder update %.3f | total %.3f" % tuple(acc / R))
ops.join_side(); ops._SIDE["on"] = False
acc = np.zeros(4)
for _ in range(R):
    e0 = ev(); st.forward(x, nz[0], nz[1]); st.gate(B); e1 = ev()
    st.backward(); e2 = ev()
    st.apply(); e3 = ev()
    torch.cuda.synchronize()
    acc += [e0.elapsed_time(e1), e1.elapsed_time(e2), e2.elapsed_time(e3), e0.elapsed_time(e3)]
print("one stream eager: forward+gate %.3f ms | backward %.3f | updates %.3f | total %.3f" % tuple(acc / R))
