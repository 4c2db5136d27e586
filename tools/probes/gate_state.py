import os, sys
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "thesis-fmri-reconstruction_amd"))
import numpy as np, torch
import bench
st, run, fl, _ = bench.build_stage1("cuda:0", 256, 0, False, False)
seq = []
for i in range(130):
    run(i)
    if i % 1 == 0:
        l = st.logs()
        seq.append((int(l["train_dis"]), int(l["train_dec"])))
print("".join("%d%d " % s for s in seq))
print({k: round(v, 3) if isinstance(v, float) else v for k, v in st.logs().items()})
