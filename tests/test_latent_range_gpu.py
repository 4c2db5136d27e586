"""Range-safe latent path (fmri_latent_fwd_ranged, fmri_bn_cols_fwd_s / fmri_bn_finalize_s, DecoderNet(zscale=...)).

`VaeGan.reparameterize` (models/vae_gan.py:266-269) computes sigma = exp(0.5 logvar) in fp32; a training run does reach
logvar > 22.2 (one outlier row of a BatchNorm1d batch is enough), where sigma no longer fits fp16 while the reference's
arithmetic is still finite.  The engine stores such a latent batch as s * z, s a power of two, and undoes the scale
exactly in the BatchNorm1d behind `Decoder.fc` (eps * s^2) and in the data gradient (times s).
References: plain PyTorch fp32 on the CPU; tolerances as in test_kernels_gpu.py (2e-3 of the output RMS).
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _close(got, ref, what, tol=2e-3):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), what
    rms = ref.pow(2).mean().sqrt().item() + 1e-12
    err = (got - ref).abs()
    bad = err > tol * rms + tol * ref.abs()
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} off; max err {err.max():.3e} rms {rms:.3e}"


@pytest.mark.parametrize("top_logvar", [1.0, 12.0, 30.0, 60.0])
def test_latent_ranged_matches_fp32_and_picks_a_power_of_two(top_logvar):
    from fmri_hip import ops
    torch.manual_seed(3)
    B, Z = 7, 128
    head = torch.randn(B, 2 * Z) * 0.5
    head[2, Z + 5] = top_logvar                      # one outlier coordinate
    head[4, Z:] = top_logvar * 0.9                   # one outlier row
    eps = torch.randn(B, Z)
    mu, lv = head[:, :Z], head[:, Z:]
    z_ref = eps * torch.exp(0.5 * lv) + mu
    kl_ref = (-0.5 * torch.sum(-lv.exp() - mu.pow(2) + lv + 1, 1)).sum()
    zp = ops.pad8(Z)
    z16 = torch.full((B, zp), float("nan"), dtype=torch.float16, device=DEV)
    st = torch.zeros(3, device=DEV)
    z32 = ops.latent_ranged(head.to(DEV), eps.to(DEV), B, Z, z16, st[0:1], st[1:2], kl_total=st[2:3], sample=True)
    zmax, s, kl = st.tolist()
    assert zmax == pytest.approx(z_ref.abs().max().item(), rel=1e-5)
    assert math.log2(s) == int(math.log2(s)) and s <= 1.0
    if zmax <= ops.LATENT_CAP:
        assert s == 1.0
    else:
        assert zmax * s <= ops.LATENT_CAP < zmax * s * 2          # the LARGEST such power of two
    _close(z32, z_ref, "z32", tol=1e-5)
    assert kl == pytest.approx(kl_ref.item(), rel=1e-4)
    got = z16.float().cpu()
    assert torch.isfinite(got).all()
    assert torch.equal(got[:, :Z], (z_ref * s).half().float()) or (got[:, :Z] - (z_ref * s).half().float()).abs().max() \
        <= 2e-3 * (z_ref * s).abs().max()
    assert (got[:, Z:] == 0).all()


def test_latent_ranged_is_the_plain_kernel_inside_the_cap():
    """Below the cap the scaled path stores bit-identical rows (s = 1): nothing changes for a healthy step."""
    from fmri_hip import lib, ops
    P = lib.ptr
    torch.manual_seed(4)
    B, Z = 33, 128
    head, eps = (torch.randn(B, 2 * Z) * 0.7).to(DEV), torch.randn(B, Z).to(DEV)
    a = torch.empty(B, Z, dtype=torch.float16, device=DEV)
    b = torch.empty(B, Z, dtype=torch.float16, device=DEV)
    k1, st = torch.zeros(1, device=DEV), torch.zeros(3, device=DEV)
    lib.call("fmri_latent_fwd", P(head), P(eps), B, Z, Z, P(a), None, P(k1), 1)
    ops.latent_ranged(head, eps, B, Z, b, st[0:1], st[1:2], kl_total=st[2:3])
    assert st[1].item() == 1.0
    assert torch.equal(a, b)
    assert k1.item() == pytest.approx(st[2].item(), rel=1e-6)


def test_non_finite_latent_is_carried_on_not_hidden():
    from fmri_hip import ops
    B, Z = 4, 128
    head = torch.zeros(B, 2 * Z)
    head[1, Z + 3] = 200.0                      # exp(100) overflows fp32: the reference's z is inf here too
    eps = torch.ones(B, Z)
    z16 = torch.empty(B, Z, dtype=torch.float16, device=DEV)
    st = torch.zeros(2, device=DEV)
    ops.latent_ranged(head.to(DEV), eps.to(DEV), B, Z, z16, st[0:1], st[1:2])
    assert st[1].item() == 1.0
    assert torch.isinf(z16[1, 3]) and torch.isfinite(z16[0]).all()


@pytest.mark.parametrize("M,C", [(256, 1024), (6, 64), (2500, 256)])
def test_batchnorm_of_range_scaled_rows_equals_batchnorm(M, C):
    """BN_eps(x) == BN_{eps s^2}(s x): output, running statistics and backward through the stored rows."""
    from fmri_hip.ops import BatchNorm
    from test_kernels_gpu import _G
    torch.manual_seed(5)
    s = 2.0 ** -9
    x = (torch.randn(M, C) * 40.0).half().float()
    x[1] *= 30.0                                                    # an outlier row
    x = x.half().float()
    gam, bet = 1 + 0.2 * torch.randn(C), 0.1 * torch.randn(C)
    bn_ref = torch.nn.BatchNorm1d(C, momentum=0.9)
    with torch.no_grad():
        bn_ref.weight.copy_(gam); bn_ref.bias.copy_(bet)
    xr = x.clone().requires_grad_(True)
    y_ref = F.relu(bn_ref(xr))
    dy = torch.randn(M, C).half().float()
    y_ref.backward(dy)

    def engine(scale):
        g = _G({"bn.weight": gam, "bn.bias": bet})
        g.bufs = {"bn.running_mean": torch.zeros(C, device=DEV), "bn.running_var": torch.ones(C, device=DEV),
                  "bn.num_batches_tracked": torch.zeros((), dtype=torch.int64, device=DEV)}
        bn = BatchNorm(g, "bn.", C)
        xs = (x * scale).half().to(DEV)
        ins = None if scale == 1.0 else torch.tensor([scale], device=DEV)
        y, sv = bn.forward(xs, relu=True, updates=1, in_scale=ins)
        dx, _ = bn.backward(xs, dy.half().to(DEV), sv, True, 1.0)
        return y, dx, g

    y, dx_stored, g = engine(s)
    _close(y, y_ref, "y")
    _close(g.bufs["bn.running_mean"], bn_ref.running_mean, "running_mean", tol=1e-3)
    _close(g.bufs["bn.running_var"], bn_ref.running_var, "running_var", tol=1e-3)
    assert int(g.bufs["bn.num_batches_tracked"]) == 1
    # d/dx = s * d/d(stored rows); the parameter gradients need no correction
    _close(dx_stored.float() * s, xr.grad, "dx", tol=4e-3)
    _close(g.grads["bn.weight"], bn_ref.weight.grad, "dgamma", tol=3e-3)
    _close(g.grads["bn.bias"], bn_ref.bias.grad, "dbeta", tol=3e-3)


def test_decoder_on_range_scaled_latent_matches_unscaled(monkeypatch):
    """The whole decoder (forward, weight gradients, dz) on a latent batch stored at s = 2^-k against the same batch
    stored at scale 1 (both fit fp16 here): the scale must cancel everywhere."""
    from fmri_hip import ops
    from fmri_hip.nets import DecoderNet
    from fmri_hip.params import ArchConfig
    cfg = ArchConfig.px64()
    torch.manual_seed(6)
    B, Z = 8, cfg.latent_dim
    z = torch.randn(2 * B, Z) * 3.0
    z[3] *= 40.0                                   # outlier row: |z| up to ~500
    z = z.to(DEV)
    cot = (torch.randn(2 * B, 64, 64, 8, device=DEV) * 0.05).half()
    cot[..., 3:] = 0

    def run(cap):
        monkeypatch.setattr(ops, "LATENT_CAP", cap)
        net = DecoderNet(cfg, DEV)
        net.group.load_recipe(np.random.RandomState(11), True)
        z16 = torch.empty(2 * B, ops.pad8(Z), dtype=torch.float16, device=DEV)
        st = torch.zeros(4, device=DEV)
        zs = torch.ones(2, device=DEV)
        ops.latent_ranged(None, None, B, Z, z16[:B], st[0:1], zs[0:1], z32=z[:B].contiguous())
        ops.latent_ranged(None, None, B, Z, z16[B:], st[1:2], zs[1:2], z32=z[B:].contiguous())
        y, ctx = net.forward(z16, 2, zscale=zs)
        net.group.zero_grad()
        dz = net.backward(ctx, cot, [dict(g=0, scale=1.0, train=True, need_dz=True),
                                     dict(g=1, scale=1.0, train=True, need_dz=True)])
        torch.cuda.synchronize()
        sd = net.group.state_dict()
        return (zs.tolist(), y.float().cpu(), {k: v.clone().cpu() for k, v in net.group.grads.items()},
                dz[0].cpu(), dz[1].cpu(), sd["fc.1.running_mean"].cpu(), sd["fc.1.running_var"].cpu())

    s_a, y_a, g_a, dz0_a, dz1_a, rm_a, rv_a = run(1.0e4)          # scale 1 for both groups
    s_b, y_b, g_b, dz0_b, dz1_b, rm_b, rv_b = run(8.0)            # group 0 (outlier) and group 1 scaled differently
    assert s_a == [1.0, 1.0] and s_b[0] < s_b[1] < 1.0
    _close(y_b, y_a, "decoder output", tol=4e-3)
    _close(dz0_b, dz0_a, "dz group 0", tol=2e-2)
    _close(dz1_b, dz1_a, "dz group 1", tol=2e-2)
    _close(rm_b, rm_a, "fc.1.running_mean", tol=2e-3)
    _close(rv_b, rv_a, "fc.1.running_var", tol=2e-3)
    for k in g_a:
        _close(g_b[k], g_a[k], "grad " + k, tol=2e-2)


def test_stage1_step_survives_a_latent_excursion():
    """A step whose encoder emits logvar ~ 30 on one row (sigma = 3e6: beyond fp16) stays finite and updates finite
    weights -- the reference's fp32 arithmetic does -- and matches the oracle's losses of that step."""
    from fmri_hip.params import ArchConfig
    from fmri_hip.steps import Stage1Step
    from oracle import vaegan_oracle as O
    B = 8
    cfg_o, cfg_e = O.ArchCfg.px64(), ArchConfig.px64()
    data = O.synth_batch(B, cfg_o, seed=77, steps=1)
    P = O.fill_state(O.vaegan_spec(cfg_o), 5, True)
    # push l_var's bias so that logvar sits near 30 for every sample (a whole-batch excursion), one row further out
    P["encoder.l_var.bias"] = P["encoder.l_var.bias"] + 24.0
    st = Stage1Step(cfg_e, DEV)
    st.load_state_dict({k: (v.reshape(()) if k.endswith("num_batches_tracked") else v.clone()) for k, v in P.items()})
    x, eps, zp = data["x"], data["noise"][0, 0], data["noise"][0, 1]
    st.step(x.to(DEV), eps.to(DEV), zp.to(DEV))
    torch.cuda.synchronize()
    got = st.logs()
    opts = {n: O.OptState(kind="rmsprop", lr=1e-4) for n in ("encoder", "decoder", "discriminator")}
    ref = O.stage1_step(P, opts, x, eps, zp, cfg_o)
    assert ref["fw"]["log_variances"].max().item() > 23.0, "the case must lie beyond fp16's sigma range"
    assert st.zs[0].item() < 1.0
    for k, v in got.items():
        if isinstance(v, float):
            assert np.isfinite(v), (k, v)
    for k in ("kl", "nle", "mse", "bce_orig", "bce_pred", "bce_samp"):
        rel = abs(got[k] - ref["logs"][k]) / max(abs(ref["logs"][k]), 1e-12)
        assert rel < 2e-2, (k, got[k], ref["logs"][k], rel)
    sd = st.state_dict()
    for k, v in sd.items():
        if v.is_floating_point():
            assert torch.isfinite(v).all(), k


def test_batchnorm_backward_saturates_finite_overflow_and_keeps_nan():
    """`sat16` of csrc/norm.hip: a BatchNorm-backward result beyond fp16's range is stored as +-65504, not inf (one inf
    there turned a whole step into NaN) -- and a NaN cotangent stays a NaN (nothing is hidden)."""
    from fmri_hip.ops import BatchNorm
    from test_kernels_gpu import _G
    M, C = 64, 64
    torch.manual_seed(9)
    x = torch.randn(M, C)
    x[:, 5] = 1.0 + 1e-4 * torch.randn(M)          # a nearly constant feature: rstd ~ 1 / sqrt(1e-8 + 1e-5) = 316
    gam = torch.ones(C)
    gam[5] = 50.0
    g = _G({"bn.weight": gam, "bn.bias": torch.zeros(C)})
    g.bufs = {"bn.running_mean": torch.zeros(C, device=DEV), "bn.running_var": torch.ones(C, device=DEV),
              "bn.num_batches_tracked": torch.zeros((), dtype=torch.int64, device=DEV)}
    bn = BatchNorm(g, "bn.", C)
    xs = x.half().to(DEV)
    _, sv = bn.forward(xs, relu=False, updates=0)
    dy = (torch.randn(M, C) * 100.0).half().to(DEV)          # gamma * rstd * g ~ 50 * 316 * 100 >> 65504
    dy[3, 9] = float("nan")
    dx, _ = bn.backward(xs, dy, sv, False, None)
    col = dx[:, 5].float()
    assert torch.isfinite(col).all() and col.abs().max().item() == 65504.0
    assert torch.isnan(dx[:, 9].float()).all()               # the NaN went through the column's batch sums
    ok = torch.ones(C, dtype=torch.bool)
    ok[5] = ok[9] = False
    assert torch.isfinite(dx[:, ok].float()).all()
