"""Drop-in API surface of ``models.vae_gan`` / ``configs.models_config``.

CPU part: module tree, constructor signatures and state_dict keys/shapes/order equal the reference's
(key lists come from the golden fixtures written by the real reference).  GPU part: the reference's literal
Stage-I loop body (three ``backward(retain_graph=True)`` calls interleaved with optimizer steps,
train/train_vgan_stage1.py:330-432) runs unchanged on the HIP engine and reproduces the oracle.
"""
import os

import numpy as np
import pytest
import torch


def _cfg64():
    import configs.models_config as mc
    mc.use_px64()
    return mc


def test_state_dict_keys_match_reference(golden_dir):
    _cfg64()
    import models.vae_gan as vg
    g = np.load(os.path.join(golden_dir, "stage1_b4.npz"))
    ref_keys = [str(k) for k in g["step0/state_keys"]]
    m = vg.VaeGan(device="cpu", z_size=128)
    assert list(m.state_dict().keys()) == ref_keys
    from oracle import vaegan_oracle as O
    shapes = {k: tuple(s) for k, s, _ in O.vaegan_spec(O.ArchCfg.px64())}
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == shapes[k], k
    # stage-II wiring: shared sub-modules, teacher_net.* keys (train_vgan_stage2.py:211-232)
    g2 = np.load(os.path.join(golden_dir, "stage2_b4.npz"))
    teacher = vg.VaeGan(device="cpu", z_size=128)
    cog = vg.CognitiveEncoder(input_size=4096, z_size=128)
    m2 = vg.VaeGanCognitive(device="cpu", encoder=cog, decoder=teacher.decoder, discriminator=teacher.discriminator,
                            teacher_net=teacher, stage=2, z_size=128)
    assert list(m2.state_dict().keys()) == [str(k) for k in g2["step0/state_keys"]]
    assert m2.decoder is teacher.decoder and m2.discriminator is teacher.discriminator
    # WAE container
    g3 = np.load(os.path.join(golden_dir, "wae1_b4.npz"))
    w = vg.WaeGan(device="cpu", z_size=128)
    assert list(w.state_dict().keys()) == [str(k) for k in g3["step0/state_keys"]]
    assert vg.VisualEncoder is vg.Encoder and vg.CognitiveVaeGan is vg.VaeGanCognitive
    assert vg.Encoder(z_size=128).size == 256           # attribute callers use (train_vgan_stage2.py:212)


def test_as_shipped_config_is_the_paper_setting():
    import importlib
    import configs.models_config as mc
    importlib.reload(mc)
    assert (mc.image_size, mc.fc_input, mc.latent_dim, mc.stride_gan) == (100, 13, 512, 2)
    assert mc.decoder_channels == [256, 128, 64, 3] and mc.output_pad_dec == [False, True, True]
    mc.use_px64()
    assert (mc.image_size, mc.fc_input, mc.latent_dim, mc.stride_gan, mc.fc_output_gan) == (64, 8, 128, 1, 512)


def test_cpu_forward_fails_loudly():
    _cfg64()
    import models.vae_gan as vg
    enc = vg.Encoder(z_size=128)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        enc(torch.zeros(2, 3, 64, 64))


@pytest.mark.gpu
def test_literal_reference_loop_body_runs_on_engine():
    from oracle import vaegan_oracle as O
    _cfg64()
    import models.vae_gan as vg
    dev = "cuda:0"
    cfg = O.ArchCfg.px64()
    B = 4
    data = O.synth_batch(B, cfg, seed=1234, steps=1)
    sd0 = O.fill_state(O.vaegan_spec(cfg), 0, True)
    model = vg.VaeGan(device=dev, z_size=128).to(dev)
    model.load_state_dict(sd0)
    model.train()
    lr = 1e-4
    mk = lambda p: torch.optim.RMSprop(params=p, lr=lr, alpha=0.9, eps=1e-8, weight_decay=0, momentum=0, centered=False)
    opt_e, opt_d, opt_s = mk(model.encoder.parameters()), mk(model.decoder.parameters()), mk(
        model.discriminator.parameters())
    x = data["x"].to(dev)
    eps, z_p = data["noise"][0, 0].to(dev), data["noise"][0, 1].to(dev).requires_grad_(True)
    # forward with explicit noise (sub-modules in the order of VaeGan.forward)
    mus, lv = model.encoder(x)
    x_tilde = model.decoder(eps * torch.exp(0.5 * lv) + mus)
    x_p = model.decoder(z_p)
    disc_layer = model.discriminator(x, x_tilde, x_p, "REC")
    disc_class = model.discriminator(x, x_tilde, x_p, "GAN")
    nle, kld, mse, bo, bp, bs = vg.VaeGan.loss(x, x_tilde, disc_layer[:B], disc_layer[B:-B], disc_layer[-B:],
                                               disc_class[:B], disc_class[B:-B], disc_class[-B:], mus, lv)
    lam = 1e-6
    loss_encoder = torch.sum(kld) + torch.sum(mse)
    loss_discriminator = torch.sum(bo) + torch.sum(bp) + torch.sum(bs)
    loss_decoder = torch.sum(lam * mse) - (1.0 - lam) * loss_discriminator
    # literal order of train_vgan_stage1.py:408-432 (raises on torch >= 1.5 with the reference's own modules)
    model.zero_grad()
    loss_encoder.backward(retain_graph=True)
    g_enc = {k: p.grad.clone() for k, p in model.encoder.named_parameters()}
    opt_e.step()
    model.zero_grad()
    loss_decoder.backward(retain_graph=True)
    g_dec = {k: p.grad.clone() for k, p in model.decoder.named_parameters()}
    opt_d.step()
    model.discriminator.zero_grad()
    loss_discriminator.backward()
    g_dis = {k: p.grad.clone() for k, p in model.discriminator.named_parameters()}
    opt_s.step()
    # oracle
    P = O.fill_state(O.vaegan_spec(cfg), 0, True)
    opts = {n: O.OptState(kind="rmsprop", lr=lr) for n in ("encoder", "decoder", "discriminator")}
    ref = O.stage1_step(P, opts, data["x"], data["noise"][0, 0], data["noise"][0, 1], cfg, keep_grads=True)
    got = dict(loss_encoder=loss_encoder.item(), loss_decoder=loss_decoder.item(),
               loss_discriminator=loss_discriminator.item(), nle=nle.sum().item(), kl=kld.sum().item(),
               mse=mse.sum().item())
    for k, v in got.items():
        assert abs(v - ref["logs"][k]) < 1e-3 * abs(ref["logs"][k]), (k, v, ref["logs"][k])

    def rel(a, b):
        a, b = a.float().cpu().reshape(-1), b.float().reshape(-1)
        return ((a - b).norm() / (b.norm() + 1e-30)).item()
    import gradcheck
    got_g = {pre + k: v for pre, gd in (("encoder.", g_enc), ("decoder.", g_dec), ("discriminator.", g_dis))
             for k, v in gd.items()}
    P16 = O.fill_state(O.vaegan_spec(cfg), 0, True)
    o16 = {n: O.OptState(kind="rmsprop", lr=lr) for n in ("encoder", "decoder", "discriminator")}
    with gradcheck.storage16(O):
        ref16 = O.stage1_step(P16, o16, data["x"], data["noise"][0, 0], data["noise"][0, 1], cfg, keep_grads=True)
    gradcheck.check(got_g, ref["grads"], ref16["grads"], "api stage1", tol16=None)
    # parameters after the three optimizer steps and BN bookkeeping
    sd = model.state_dict()
    for k, v in P.items():
        if "num_batches" in k:
            assert int(sd[k]) == int(v), k
        elif v.dtype == torch.float32 and "running" in k:
            assert rel(sd[k], v) < 2e-2, k
    upd = rel(sd["discriminator.fc.0.weight"] - sd0["discriminator.fc.0.weight"].to(dev),
              P["discriminator.fc.0.weight"] - sd0["discriminator.fc.0.weight"])
    print("rel err of the discriminator.fc.0.weight update", upd)
    assert upd < 0.3
    # eval-mode forward (inference path, models/vae_gan.py:288-297): runs, right shape, finite
    model.eval()
    with torch.no_grad():
        out = model(x)
    assert out.shape == (B, 3, 64, 64) and torch.isfinite(out).all()
    gen = model(None, 5)
    assert gen.shape == (5, 3, 64, 64)


@pytest.mark.gpu
def test_eval_forward_matches_reference_golden(golden_dir):
    """Inference path of the drop-in modules (BN with running statistics) against the reference's eval forward."""
    from oracle import vaegan_oracle as O
    from test_oracle_golden import eval_state
    _cfg64()
    import models.vae_gan as vg
    cfg = O.ArchCfg.px64()
    g = np.load(os.path.join(golden_dir, "eval_b4.npz"))
    B, seed = int(g["meta/B"]), int(g["meta/seed"])
    dev = "cuda:0"
    model = vg.VaeGan(device=dev, z_size=128).to(dev)
    model.load_state_dict(eval_state(cfg, seed))
    model.eval()
    data = O.synth_batch(B, cfg, seed=1234, steps=1)
    with torch.no_grad():
        mus, lv = model.encoder(data["x"].to(dev))
        x_tilde = model.decoder(data["noise"][0, 0].to(dev) * torch.exp(0.5 * lv) + mus)
        x_p = model.decoder(data["noise"][0, 1].to(dev))
        assert model(data["x"].to(dev)).shape == x_tilde.shape
    for k, v in dict(mus=mus, log_variances=lv, x_tilde=x_tilde, x_p=x_p).items():
        ref = g[f"fw/{k}"]
        got = O.tensor_summary(v.float().cpu())
        assert abs(got[0] - ref[0]) < 2e-3 * abs(ref[0]), (k, got[0], ref[0])          # L2 norm
        assert np.abs(got[2:] - ref[2:]).max() < 5e-3 * max(np.abs(ref[2:]).max(), 1e-3), k   # head / tail elements
    # running statistics and counters are untouched by eval-mode forwards
    sd = model.state_dict()
    ref_sd = eval_state(cfg, seed)
    for k in ("encoder.conv.0.bn.running_mean", "decoder.fc.1.running_var", "decoder.conv.2.bn.num_batches_tracked"):
        assert torch.equal(sd[k].cpu().reshape(-1).float(), ref_sd[k].reshape(-1).float()), k


@pytest.mark.gpu
def test_literal_stage2_loop_body_runs_on_engine():
    """The Stage-II loop body of train/train_vgan_stage2.py:330-407 on the drop-in modules: VaeGanCognitive wiring with
    the teacher distillation (models/vae_gan.py:359-395, noise passed explicitly in the reference's draw order),
    ``loss_encoder.backward(retain_graph=True)``, ``p.grad.data.clamp_(-1, 1)``, ``optimizer_encoder.step()``,
    ``model.zero_grad()``, then the discriminator the same way -- against the oracle's Stage-II step."""
    from oracle import vaegan_oracle as O
    import gradcheck
    _cfg64()
    import models.vae_gan as vg
    dev = "cuda:0"
    cfg = O.ArchCfg.px64()
    B, V, seed = 4, 4096, 1
    data = O.synth_batch(B, cfg, n_voxels=V, seed=1234, steps=1)
    tsd = O.fill_state(O.vaegan_spec(cfg), seed, True)
    csd = O.fill_state(O.cognitive_encoder_spec(cfg, V), seed + 100, True)
    teacher = vg.VaeGan(device=dev, z_size=128).to(dev)
    teacher.load_state_dict(tsd)
    cog = vg.CognitiveEncoder(input_size=V, z_size=128).to(dev)
    cog.load_state_dict({k[len("encoder."):]: v for k, v in csd.items()})
    model = vg.VaeGanCognitive(device=dev, encoder=cog, decoder=teacher.decoder, discriminator=teacher.discriminator,
                               teacher_net=teacher, stage=2, z_size=128).to(dev)
    model.train()
    lr = 1e-4
    mk = lambda p: torch.optim.RMSprop(params=p, lr=lr, alpha=0.9, eps=1e-8, weight_decay=0, momentum=0, centered=False)
    opt_e, opt_s = mk(model.encoder.parameters()), mk(model.discriminator.parameters())
    fmri, image = data["fmri"].to(dev), data["x"].to(dev)
    nz = data["noise"][0]
    eps, z_p, eps_t = nz[0].to(dev), nz[1].to(dev).requires_grad_(True), nz[2].to(dev)
    mus, lv = model.encoder(fmri)
    x_tilde = model.decoder(eps * torch.exp(0.5 * lv) + mus)
    for param in model.teacher_net.encoder.parameters():
        param.requires_grad = False
    mu_t, lv_t = model.teacher_net.encoder(image)
    gt_x = model.decoder(eps_t * torch.exp(0.5 * lv_t) + mu_t)
    x_p = model.decoder(z_p)
    disc_layer = model.discriminator(gt_x, x_tilde, x_p, "REC")
    disc_class = model.discriminator(gt_x, x_tilde, x_p, "GAN")
    nle, kld, mse, bo, bp, bs = vg.VaeGanCognitive.loss(gt_x, x_tilde, disc_layer[:B], disc_layer[B:-B], disc_layer[-B:],
                                                        disc_class[:B], disc_class[B:-B], disc_class[-B:], mus, lv)
    loss_encoder = torch.sum(kld) + torch.sum(mse)
    loss_discriminator = torch.sum(bo) + torch.sum(bp) + torch.sum(bs)
    model.zero_grad()
    loss_encoder.backward(retain_graph=True)
    [p.grad.data.clamp_(-1, 1) for p in model.encoder.parameters()]
    g_enc = {"encoder." + k: p.grad.clone() for k, p in model.encoder.named_parameters()}
    opt_e.step()
    model.zero_grad()
    loss_discriminator.backward()
    [p.grad.data.clamp_(-1, 1) for p in model.discriminator.parameters()]
    g_dis = {"discriminator." + k: p.grad.clone() for k, p in model.discriminator.named_parameters()}
    opt_s.step()
    # oracle (its optimizer steps update the tensors in place: keep the initial values)
    w_init = {k: v.clone() for k, v in {**tsd, **csd}.items()}
    P = dict(csd)
    P.update({k: v for k, v in tsd.items() if k.startswith(("decoder.", "discriminator."))})
    for k, v in tsd.items():
        P["teacher_net." + k] = P[k] if k.startswith(("decoder.", "discriminator.")) else v
    opts = {n: O.OptState(kind="rmsprop", lr=lr) for n in ("encoder", "decoder", "discriminator")}
    ref = O.stage2_step(P, opts, data["fmri"], data["x"], nz, cfg, V, keep_grads=True)
    got = dict(loss_encoder=loss_encoder.item(), loss_discriminator=loss_discriminator.item(), nle=nle.sum().item(),
               kl=kld.sum().item(), mse=mse.sum().item(), bce_orig=bo.sum().item(), bce_pred=bp.sum().item(),
               bce_samp=bs.sum().item())
    for k, v in got.items():
        assert abs(v - ref["logs"][k]) < 1e-3 * abs(ref["logs"][k]), (k, v, ref["logs"][k])
    # gradients BEFORE the clamp are what the oracle returns; the clamp only touches elements beyond +-1
    grads = {**g_enc, **g_dis}
    refg = {k: v.clamp(-1, 1) for k, v in ref["grads"].items() if v is not None}
    gradcheck.check(grads, refg, refg, "api stage2", tol16=None)
    # the two optimizer steps moved the same parameters the same way: RMSprop's first update is +-3.16 lr per element
    # (sign-like), so compare the UPDATE vectors -- a few % of the signs differ under fp16 noise at B = 4
    sd = model.state_dict()
    for k, v in P.items():
        if k.startswith(("encoder.", "discriminator.")) and v.dtype == torch.float32 and "running" not in k \
                and v.numel() > 1000:
            w0 = w_init[k].reshape(-1)
            ua, ub = sd[k].float().cpu().reshape(-1) - w0, v.reshape(-1) - w0
            cos = (ua @ ub / (ua.norm() * ub.norm() + 1e-30)).item()
            assert cos > 0.9 and abs(ua.norm().item() / ub.norm().item() - 1) < 0.05, (k, cos)


@pytest.mark.gpu
def test_literal_wae_stage1_loop_body_runs_on_engine():
    """The WAE Stage-I loop body of train/train_wae_stage1.py:259-311 on the drop-in modules, including the
    ``requires_grad`` toggling between the two phases (free_params / frozen_params, :33-40) and the pairs of
    ``backward(retain_graph=True)`` calls -- against the oracle's WAE Stage-I step."""
    from oracle import vaegan_oracle as O
    _cfg64()
    import models.vae_gan as vg
    dev = "cuda:0"
    cfg = O.ArchCfg.px64()
    B, seed = 4, 5
    data = O.synth_batch(B, cfg, seed=1234, steps=1)
    sd0 = O.fill_state(O.encoder_spec(cfg) + O.decoder_spec(cfg) + O.wae_discriminator_spec(cfg), seed, False)
    model = vg.WaeGan(device=dev, z_size=128).to(dev)
    model.load_state_dict(sd0)

    def free_params(module):
        for p in module.parameters():
            p.requires_grad = True

    def frozen_params(module):
        for p in module.parameters():
            p.requires_grad = False
    opt_enc = torch.optim.Adam(model.encoder.parameters(), lr=1e-4, betas=(0.5, 0.999))
    opt_dec = torch.optim.Adam(model.decoder.parameters(), lr=1e-4, betas=(0.5, 0.999))
    opt_dis = torch.optim.Adam(model.discriminator.parameters(), lr=0.5e-4, betas=(0.5, 0.999))
    x = data["x"].to(dev)
    model.train()
    model.encoder.zero_grad()
    model.decoder.zero_grad()
    model.discriminator.zero_grad()
    # ---------- discriminator phase
    frozen_params(model.decoder)
    frozen_params(model.encoder)
    free_params(model.discriminator)
    z_real, var = model.encoder(x)
    z_fake = (data["noise"][0, 2] * 0.5).to(dev)
    d_real = model.discriminator(z_real)
    d_fake = model.discriminator(z_fake)
    loss_discriminator_fake = -10 * torch.sum(torch.log(d_fake + 1e-3))
    loss_discriminator_real = -10 * torch.sum(torch.log(1 - d_real + 1e-3))
    loss_discriminator_fake.backward(retain_graph=True)
    loss_discriminator_real.backward(retain_graph=True)
    opt_dis.step()
    # ---------- generator phase
    free_params(model.encoder)
    free_params(model.decoder)
    frozen_params(model.discriminator)
    z_real, var = model.encoder(x)
    x_recon = model.decoder(z_real)
    d_real = model.discriminator(z_real)
    loss_reconstruction = torch.sum(torch.sum(0.5 * (x_recon - x) ** 2, 1))
    loss_penalty = -10 * torch.sum(torch.log(d_real + 1e-3))
    loss_reconstruction.backward(retain_graph=True)
    loss_penalty.backward()
    opt_enc.step()
    opt_dec.step()
    # oracle
    P = O.fill_state(O.encoder_spec(cfg) + O.decoder_spec(cfg) + O.wae_discriminator_spec(cfg), seed, False)
    opts = {"encoder": O.OptState(kind="adam", lr=1e-4), "decoder": O.OptState(kind="adam", lr=1e-4),
            "discriminator": O.OptState(kind="adam", lr=0.5e-4)}
    ref = O.wae_stage1_step(P, opts, data["x"], data["noise"][0, 2], cfg)
    got = dict(loss_reconstruction=loss_reconstruction.item(), loss_penalty=loss_penalty.item(),
               loss_discriminator_fake=loss_discriminator_fake.item(),
               loss_discriminator_real=loss_discriminator_real.item())
    for k, v in got.items():
        tol = 5e-3 if k == "loss_penalty" else 1e-3        # the penalty is scored after the discriminator's Adam step
        assert abs(v - ref["logs"][k]) < tol * abs(ref["logs"][k]), (k, v, ref["logs"][k])
    # Adam's first update is lr * sign(g): compare the update vectors (elements whose gradient changed sign under fp16
    # noise move the other way)
    sd = model.state_dict()
    for k, v in P.items():
        if v.dtype == torch.float32 and "running" not in k and v.numel() > 1000:
            w0 = sd0[k].reshape(-1)
            ua, ub = sd[k].float().cpu().reshape(-1) - w0, v.reshape(-1) - w0
            if ub.norm().item() == 0.0:          # l_var: the WAE encoder's variance head receives no gradient
                assert ua.norm().item() == 0.0, k
                continue
            cos = (ua @ ub / (ua.norm() * ub.norm() + 1e-30)).item()
            assert cos > 0.9 and abs(ua.norm().item() / ub.norm().item() - 1) < 0.05, (k, cos)


def test_init_parameters_bounds_and_fans():
    """VaeGan.init_parameters (reference models/vae_gan.py:252-264): every Conv2d / ConvTranspose2d / Linear weight is
    U(-s, s) with s = 1 / sqrt(prod(shape[1:])) / sqrt(3) -- for a ConvTranspose2d that fan is (C_out, k, k), not C_in --
    biases are 0, BatchNorm parameters are left at their defaults."""
    import math
    _cfg64()
    import models.vae_gan as vg
    torch.manual_seed(11)
    m = vg.VaeGan(device="cpu", z_size=128)
    seen = 0
    for name, mod in m.named_modules():
        if isinstance(mod, (torch.nn.Conv2d, torch.nn.ConvTranspose2d, torch.nn.Linear)):
            w = mod.weight.detach()
            s = 1.0 / math.sqrt(float(np.prod(w.shape[1:]))) / math.sqrt(3.0)
            assert float(w.abs().max()) <= s * (1 + 1e-6), name
            if w.numel() >= 1000:
                assert float(w.abs().max()) > 0.98 * s, name                       # the bound is attained
                assert abs(float(w.std()) - s / math.sqrt(3.0)) < 0.05 * s, name    # uniform, not normal
                assert abs(float(w.mean())) < 0.05 * s, name
            if mod.bias is not None:
                assert float(mod.bias.detach().abs().max()) == 0.0, name
            seen += 1
        elif isinstance(mod, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
            assert torch.equal(mod.weight.detach(), torch.ones_like(mod.weight)), name
            assert torch.equal(mod.bias.detach(), torch.zeros_like(mod.bias)), name
    assert seen == 3 + 3 + 4 + 1 + 4 + 2      # enc: 3 conv, fc, 2 heads; dec: fc, 3 deconv, conv; dis: 4 conv, 2 fc
    # the transposed convolutions' fan: decoder.conv.1 maps 256 -> 128 channels, weight (256, 128, 5, 5)
    w = m.decoder.conv[1].conv.weight.detach()
    assert tuple(w.shape) == (256, 128, 5, 5)
    assert float(w.abs().max()) > 0.98 / math.sqrt(128 * 25) / math.sqrt(3.0)


@pytest.mark.gpu
def test_encoder_and_decoder_blocks_forward_on_their_own():
    """EncoderBlock.forward(ten, out) / DecoderBlock.forward(ten) (reference models/vae_gan.py:23-35, :56-60) on the
    engine against torch on the CPU, train and eval mode, running statistics included."""
    import torch.nn.functional as F
    _cfg64()
    import models.vae_gan as vg
    dev = "cuda:0"
    torch.manual_seed(3)
    for kind in ("enc", "dec"):
        blk = (vg.EncoderBlock(64, 128) if kind == "enc" else vg.DecoderBlock(128, 64, out=True))
        with torch.no_grad():
            blk.conv.weight.copy_((torch.randn_like(blk.conv.weight) * 0.05).half().float())
            blk.bn.weight.uniform_(0.5, 1.5)
            blk.bn.bias.uniform_(-0.2, 0.2)
        ref = (vg.EncoderBlock if kind == "enc" else vg.DecoderBlock)
        x = torch.randn(3, 64 if kind == "enc" else 128, 12, 12).half().float()
        w, g, b = blk.conv.weight.detach().clone(), blk.bn.weight.detach().clone(), blk.bn.bias.detach().clone()
        rm, rv = torch.zeros_like(g), torch.ones_like(g)
        raw = F.conv2d(x, w, None, 2, 2) if kind == "enc" else F.conv_transpose2d(x, w, None, 2, 2, output_padding=1)
        want = F.relu(F.batch_norm(raw, rm, rv, g, b, True, 0.9, 1e-5))
        blk = blk.to(dev)
        blk.train()
        got = blk(x.to(dev), True) if kind == "enc" else blk(x.to(dev))
        act = got[0] if kind == "enc" else got
        assert (act.cpu() - want).abs().max().item() < 2e-2 * want.abs().max().item()
        if kind == "enc":
            assert (got[1].cpu() - raw).abs().max().item() < 2e-3 * raw.abs().max().item()
        assert torch.allclose(blk.bn.running_mean.cpu(), rm, atol=2e-3) and int(blk.bn.num_batches_tracked) == 1
        blk.eval()
        want_e = F.relu(F.batch_norm(raw, rm, rv, g, b, False, 0.9, 1e-5))
        got_e = blk(x.to(dev))
        assert (got_e.cpu() - want_e).abs().max().item() < 2e-2 * want_e.abs().max().item()
