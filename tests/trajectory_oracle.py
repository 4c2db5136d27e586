#!/usr/bin/env python3
"""Test helper (not collected): the fp32 CPU oracle's Stage-I trajectory on bench.py's data sequence -- per step the
loss block, the gate decisions and max |mu| / max logvar.  Used to tell the reference arithmetic's own excursions from
the engine's (tests/test_trajectory_gpu.py).

    python tests/trajectory_oracle.py --batch 256 --steps 150 --out /tmp/traj/oracle_b256.jsonl
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from oracle import vaegan_oracle as O  # noqa: E402

NBATCH = 8


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    if a.threads:
        torch.set_num_threads(a.threads)
    cfg = O.ArchCfg.px64()
    P = O.fill_state(O.vaegan_spec(cfg), 0, False)
    B, Z = a.batch, cfg.latent_dim
    xs = [torch.from_numpy(np.random.RandomState(1234 + 97 * i).uniform(-1, 1, (B, 3, 64, 64)).astype(np.float32))
          for i in range(NBATCH)]
    nz = [torch.from_numpy(np.random.RandomState(1236 + 97 * i).standard_normal((2, B, Z)).astype(np.float32))
          for i in range(NBATCH)]
    opts = {n: O.OptState(kind="rmsprop", lr=1e-4) for n in ("encoder", "decoder", "discriminator")}
    f = open(a.out, "w") if a.out else sys.stdout
    for i in range(a.steps):
        t = time.time()
        j = i % NBATCH
        r = O.stage1_step(P, opts, xs[j], nz[j][0], nz[j][1], cfg)
        fw = r["fw"]
        row = dict(step=i, **r["logs"], max_mu=float(fw["mus"].abs().max()), max_logvar=float(fw["log_variances"].max()),
                   min_logvar=float(fw["log_variances"].min()), sec=round(time.time() - t, 2))
        f.write(json.dumps(row) + "\n")
        f.flush()


if __name__ == "__main__":
    main()
