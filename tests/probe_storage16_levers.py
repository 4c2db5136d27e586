#!/usr/bin/env python3
"""Test helper (not collected): which 16-bit storage points carry the distance of the step-1 losses from fp32?

VERDICT r4 item 3 names two levers for the feature term after one update (2-3e-3 at B = 256 against the 1e-3 bar): the raw
'REC' features in fp32 and the images in fp32.  The oracle's 16-bit-storage model (STORAGE16) answers without touching the
engine: two steps at batch B with every storage point rounded, and with the tagged points kept in fp32.

    python tests/probe_storage16_levers.py --batch 256
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from oracle import vaegan_oracle as O  # noqa: E402

KEYS = ("mse", "loss_encoder", "kl", "nle", "bce_orig", "bce_pred", "bce_samp")


def two_steps(cfg, data, storage16, keep=frozenset()):
    P = O.fill_state(O.vaegan_spec(cfg), 0, True)
    opts = {n: O.OptState(kind="rmsprop", lr=1e-4) for n in ("encoder", "decoder", "discriminator")}
    O.STORAGE16, O.STORAGE16_KEEP32 = storage16, frozenset(keep)
    try:
        return [O.stage1_step(P, opts, data["x"], data["noise"][s, 0], data["noise"][s, 1], cfg)["logs"] for s in range(2)]
    finally:
        O.STORAGE16, O.STORAGE16_KEEP32 = False, frozenset()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    a = ap.parse_args()
    cfg = O.ArchCfg.px64()
    data = O.synth_batch(a.batch, cfg, seed=1234, steps=2)
    plain = two_steps(cfg, data, False)
    for name, keep in (("all 16-bit", ()), ("REC features fp32", ("rec_feat",)), ("images fp32", ("images",)),
                       ("REC features + images fp32", ("rec_feat", "images")),
                       ("every discriminator conv output fp32", ("rec_feat", "disc_raw"))):
        m = two_steps(cfg, data, True, keep)
        rel = lambda s, k: abs(m[s][k] - plain[s][k]) / max(abs(plain[s][k]), 1e-12)
        print(f"B {a.batch} {name:40s} step 0: " + " ".join(f"{k} {rel(0, k):.1e}" for k in KEYS[:2])
              + " | step 1: " + " ".join(f"{k} {rel(1, k):.1e}" for k in KEYS), flush=True)


if __name__ == "__main__":
    main()
