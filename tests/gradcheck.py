"""Gradient parity helpers for the fused-step tests (GPU engine vs the CPU oracle).

Two references per tensor:
  * the oracle run under its 16-bit storage model (``oracle.vaegan_oracle.STORAGE16``): same fp32 arithmetic, but every
    tensor the engine stores in fp16 is rounded where the engine rounds it, so the ReLU masks agree with the engine's
    and the gradients can be compared tightly (relative L2 error per tensor);
  * the plain fp32 oracle (what the golden vectors pin): fp16 activations flip the ReLU mask of the ~1e-3 of elements
    whose pre-activation lies within rounding of zero, which adds unbiased noise to back-propagated gradients -- there the
    engine's gradient must keep the direction (cosine) and the length along the reference (projection).
"""
import contextlib
import os

import torch

TOL16 = 1e-2        # relative L2 error against the 16-bit-storage oracle, per tensor
COS32 = 0.95        # cosine against the fp32 oracle
PROJ32 = 0.08       # |<g, r> / <r, r> - 1| against the fp32 oracle (B = 4: mask flips move small tensors by 3-7 %: the 64-element
                    # decoder.conv.2.bn gradients of the 100-px config read 0.952 / 0.933 with igemm_tc5 / igemm_tc5w at err16 2e-3)


@contextlib.contextmanager
def storage16(O):
    O.STORAGE16 = True
    try:
        yield
    finally:
        O.STORAGE16 = False


def report(got, ref32, ref16, skip=()):
    rows = []
    for k, r16 in ref16.items():
        if r16 is None or k in skip:
            continue
        g = got[k].detach().double().cpu().reshape(-1)
        a = r16.detach().double().reshape(-1)
        b = ref32[k].detach().double().reshape(-1)
        e16 = ((g - a).norm() / (a.norm() + 1e-30)).item()
        cos = (g @ b / (g.norm() * b.norm() + 1e-30)).item()
        proj = (g @ b / (b @ b + 1e-30)).item()
        rows.append((k, e16, cos, proj, g.numel()))
    return rows


def check(got, ref32, ref16, what="", skip=(), tol16=TOL16, cos32=COS32, proj32=PROJ32, small=64, loose16=0.15):
    """Assert the criteria for every tensor.  ``ref16`` comes from an oracle run with the engine's ReLU masks pinned
    (``oracle.RELU_MASKS``) -> bound ``tol16``; without pinned masks (pass ``tol16=None``) the 16-bit-storage run only has
    masks as far from the engine's as the fp32 run's and is held to ``loose16``.  Tensors with fewer than ``small``
    elements (the 3-element bias of the decoder's last convolution, scalar biases: sums that cancel over the batch) are
    held to 5x the 16-bit bound only."""
    if tol16 is None:
        tol16 = loose16
    rows = report(got, ref32, ref16, skip)
    rows.sort(key=lambda r: -r[1])
    log = os.environ.get("FMRI_GRADLOG")
    if log:
        with open(log, "a") as f:
            f.write(f"# {what}: worst err16 {max(r[1] for r in rows):.3e}, min cos32 "
                    f"{min(r[2] for r in rows if r[4] >= small):.4f}, max |proj32-1| "
                    f"{max(abs(r[3] - 1) for r in rows if r[4] >= small):.4f} (bounds {tol16} / {cos32} / {proj32})\n")
            for k, e16, cos, proj, n in rows[:5]:
                f.write(f"{what} {k} err16 {e16:.3e} cos32 {cos:.4f} proj32 {proj:.4f} n {n}\n")
    for k, e16, cos, proj, n in rows[:8]:
        print(f"{what} grad {k}: err16 {e16:.2e} cos32 {cos:.4f} proj32 {proj:.4f} n {n}")
    print(f"{what} worst err16 {max(r[1] for r in rows):.2e}  min cos32 {min(r[2] for r in rows if r[4] >= small):.4f}")
    bad = []
    for k, e16, cos, proj, n in rows:
        if n < small:
            if e16 > 5 * tol16:
                bad.append((k, e16, cos, proj, n))
            continue
        if e16 > tol16 or cos < cos32 or abs(proj - 1.0) > proj32:
            bad.append((k, e16, cos, proj, n))
    assert not bad, (what, bad[:6])
    return rows
