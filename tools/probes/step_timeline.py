"""GPU box: timeline of ONE hybrid-mode Stage-I step (recorded forward + eagerly issued two-stream backward): every library
launch of the backward bracketed by HIP events on its own stream; per-stream busy time, main-stream gaps, the launches the
main stream spends its time in, and how much of the side stream's work overlaps main-stream launches."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "thesis-fmri-reconstruction_amd"))
import numpy as np, torch
from fmri_hip import lib
from fmri_hip.params import ArchConfig
from fmri_hip.steps import Stage1Step
dev = "cuda:0"; B = 256
rs = np.random.RandomState(0)
x = torch.from_numpy(rs.uniform(-1, 1, (B, 3, 64, 64)).astype(np.float32)).to(dev)
e, z = (torch.from_numpy(rs.standard_normal((B, 128)).astype(np.float32)).to(dev) for _ in range(2))
st = Stage1Step(ArchConfig.px64(), dev); st.load_recipe(0, False)
for _ in range(3): st.step(x, e, z)
run = st.capture_forward(x, e, z)
for _ in range(20): run()
torch.cuda.synchronize()
recs = []
orig_call = lib.call
def call(name, *args):
    nt = lib._NOTE
    lib._NOTE = None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    sid = torch.cuda.current_stream().stream_id
    e0.record()
    orig_call(name, *args)
    e1.record()
    recs.append((name, (nt or {}).get("kernel", ""), sid, e0, e1))
lib.PROFILE = None
import fmri_hip.ops as ops, fmri_hip.nets as nets, fmri_hip.steps as steps
for m in (lib, ops, nets, steps):
    if getattr(m, "call", None) is orig_call: m.call = call
lib.call = call
base = torch.cuda.Event(enable_timing=True); endev = torch.cuda.Event(enable_timing=True)
steps_n = 3
allrecs = []
for i in range(steps_n):
    recs.clear()
    torch.cuda.synchronize()
    base.record()
    run()
    endev.record()
    torch.cuda.synchronize()
    allrecs = [(n, k, s, base.elapsed_time(a), base.elapsed_time(b)) for n, k, s, a, b in recs]
    total = base.elapsed_time(endev)
print(f"step (isolated, events around every launch): {total:.2f} ms, {len(allrecs)} eager launches")
main = torch.cuda.current_stream().stream_id
by = {}
for n, k, s, a, b in allrecs: by.setdefault(s, []).append((a, b, k or n))
for s, v in by.items():
    v.sort()
    busy = sum(b - a for a, b, _ in v)
    print(f"stream {s}{' (main)' if s == main else ''}: {len(v)} launches, first at {v[0][0]:.2f} ms, last ends {v[-1][1]:.2f} ms, sum of launch spans {busy:.2f} ms")
mv = by.get(main, [])
if mv:
    print(f"forward graph + gate (before the first eager main-stream launch): {mv[0][0]:.2f} ms")
    gaps = sum(max(0.0, mv[i + 1][0] - mv[i][1]) for i in range(len(mv) - 1))
    print(f"main stream: gaps between consecutive eager launches {gaps:.2f} ms")
    import collections
    fam = collections.Counter()
    for a, b, k in mv: fam[k.split("<")[0]] += b - a
    for k, v in fam.most_common(14): print(f"   {v:7.3f} ms  {k}")
for s, v in by.items():
    if s == main: continue
    import collections
    fam = collections.Counter()
    for a, b, k in v: fam[k.split("<")[0]] += b - a
    print("side stream launches:")
    for k, vv in fam.most_common(8): print(f"   {vv:7.3f} ms  {k}")
