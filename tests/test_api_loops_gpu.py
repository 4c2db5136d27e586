"""Drop-in API surface, the part round 2 left untested (VERDICT missing #4, #5):

  * ``VaeGanCognitive(mode='wae')`` train forward, ``WaeGanCognitive`` eval forward + constructor side effects,
    ``DCGan`` train / eval forward -- on the engine against fingerprints the REAL reference produced
    (tests/golden/surface_b4.npz, written by make_golden.py::case_surface);
  * the literal loop bodies of train_vgan_stage3.py:324-411, train_wae_stage2.py:276-328, train_wae_stage3.py:297-347
    and wae_vgan_stage1.py:290-441 on the drop-in modules (``requires_grad`` toggling, ``backward(retain_graph=True)``
    pairs, ``p.grad.data.clamp_``, torch optimizers) against the CPU oracle's restatement of the same step.

Bars: losses of the first step 1e-3 relative (north_star); the latent penalty, scored after the latent
discriminator's first sign-like update, 5e-3; update vectors by direction / length (see tests/test_api.py).
"""
import os

import numpy as np
import pytest
import torch

import gradcheck

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cfg64():
    import configs.models_config as mc
    mc.use_px64()
    return mc


def _rel(a, b):
    return abs(a - b) / max(abs(b), 1e-12)


def _summ_close(ref, t, what, ntol=2e-3, etol=1e-2):
    """L2 norm to ``ntol``; the 16 sampled elements to ``etol`` of the largest of them (fp16 storage: the raw conv-3
    features, 4 convolutions deep, carry ~1 % of their RMS per element; images and latents a few 1e-3)."""
    from oracle import vaegan_oracle as O
    got = O.tensor_summary(t.detach().float().cpu())
    if "disc_layer" in what:
        etol = max(etol, 2e-2)
    assert abs(got[0] - ref[0]) < ntol * max(abs(ref[0]), 1e-12), (what, "norm", got[0], ref[0])
    assert np.abs(got[2:] - ref[2:]).max() < etol * max(np.abs(ref[2:]).max(), 1e-3), (what, got[2:], ref[2:])


def _updates_agree(sd, P, w_init, prefixes, what, cos_min=0.9, len_tol=0.05):
    """First optimizer steps are sign-like (RMSprop: 3.16 lr sign(g); Adam: lr sign(g)): compare UPDATE vectors."""
    seen = 0
    for k, v in P.items():
        if not k.startswith(prefixes) or v.dtype != torch.float32 or "running" in k or v.numel() <= 1000:
            continue
        w0 = w_init[k].reshape(-1)
        ua, ub = sd[k].float().cpu().reshape(-1) - w0, v.reshape(-1) - w0
        if ub.norm().item() == 0.0:
            assert ua.norm().item() == 0.0, (what, k, "moved although the oracle's did not")
            continue
        cos = (ua @ ub / (ua.norm() * ub.norm() + 1e-30)).item()
        assert cos > cos_min and abs(ua.norm().item() / ub.norm().item() - 1) < len_tol, (what, k, cos)
        seen += 1
    assert seen > 0, what


def _freeze(m, flag):
    for p in m.parameters():
        p.requires_grad = not flag


# ----------------------------------------------------------------------------------------------------------------
def test_wrapper_forwards_match_reference_golden(golden_dir):
    from oracle import vaegan_oracle as O
    from test_oracle_golden import surface_states
    _cfg64()
    import models.vae_gan as vg
    cfg = O.ArchCfg.px64()
    g = np.load(os.path.join(golden_dir, "surface_b4.npz"))
    B, V, seed = int(g["meta/B"]), int(g["meta/V"]), int(g["meta/seed"])
    data = O.synth_batch(B, cfg, n_voxels=V, seed=1234, steps=1)
    x, fmri = data["x"], data["fmri"]
    tsd = O.fill_state(O.vaegan_spec(cfg), seed, True)
    csd = {k[len("encoder."):]: v for k, v in O.fill_state(O.cognitive_encoder_spec(cfg, V), seed + 100, True).items()}

    # ---- VaeGanCognitive(mode='wae'), models/vae_gan.py:379-395 (z_p is the only draw: host randn, seed 21)
    teacher = vg.VaeGan(device=DEV, z_size=128).to(DEV)
    teacher.load_state_dict(tsd)
    cog = vg.CognitiveEncoder(input_size=V, z_size=128).to(DEV)
    cog.load_state_dict(csd)
    model = vg.VaeGanCognitive(device=DEV, encoder=cog, decoder=teacher.decoder, discriminator=teacher.discriminator,
                               teacher_net=teacher, stage=2, z_size=128, mode="wae").to(DEV)
    model.train()
    torch.manual_seed(int(g["meta/torch_seed"]))
    outs = model({"fmri": fmri, "image": x})
    assert len(outs) == 6
    for k, v in zip(("gt_x", "x_tilde", "disc_class", "disc_layer", "mus", "log_variances"), outs):
        _summ_close(g[f"cogwae/{k}"], v, f"cogwae {k}")
    assert outs[3].shape == (3 * B, 16384) and outs[2].shape == (3 * B, 1)
    # BatchNorm bookkeeping of that forward: counters exact (decoder 3 calls, discriminator convs 2, teacher encoder 1),
    # running statistics close
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    keys = [str(k) for k in g["cogwae/state_keys"]]
    assert list(sd.keys()) == keys
    for i, k in enumerate(keys):
        ref = g["cogwae/state_sum"][i]
        if "num_batches" in k:
            assert float(sd[k]) == ref[1], (k, float(sd[k]), ref[1])
        elif "running" in k:
            assert _rel(sd[k].double().norm().item(), ref[0]) < 5e-3, k
    assert int(sd["decoder.conv.0.bn.num_batches_tracked"]) == 3
    assert int(sd["discriminator.conv.1.bn.num_batches_tracked"]) == 2
    assert int(sd["teacher_net.encoder.conv.0.bn.num_batches_tracked"]) == 1

    # ---- WaeGanCognitive, models/vae_gan.py:532-571: keys, frozen decoder, eval forward = decoder(encoder(fmri).mu)
    _, wae, _ = surface_states(cfg, V, seed)
    wteacher = vg.WaeGan(device=DEV, z_size=128).to(DEV)
    wteacher.decoder.load_state_dict({k[len("decoder."):]: v for k, v in wae.items() if k.startswith("decoder.")})
    wcog = vg.CognitiveEncoder(input_size=V, z_size=128).to(DEV)
    wcog.load_state_dict(csd)
    wmodel = vg.WaeGanCognitive(device=DEV, encoder=wcog, decoder=wteacher.decoder, z_size=128)
    assert list(wmodel.state_dict().keys()) == [str(k) for k in g["waecog/state_keys"]]
    assert all(not p.requires_grad for p in wmodel.decoder.parameters())
    assert wmodel.decoder is wteacher.decoder
    wmodel.eval()
    with torch.no_grad():
        _summ_close(g["waecog/x_tilde"], wmodel(fmri), "waecog eval forward")
    wmodel.train()
    with pytest.raises(TypeError):
        wmodel(fmri)            # the reference's train-mode forward calls WaeDiscriminator with 3 arguments (dead code)

    # ---- DCGan, models/vae_gan.py:581-622
    t2 = vg.VaeGan(device=DEV, z_size=128).to(DEV)
    t2.load_state_dict(tsd)
    dc = vg.DCGan(device=DEV, decoder=t2.decoder, discriminator=t2.discriminator, z_size=128)
    assert list(dc.state_dict().keys()) == [str(k) for k in g["dcgan/state_keys"]]
    dc.train()
    torch.manual_seed(22)
    outs = dc(x)
    assert len(outs) == 4
    for k, v in zip(("gt_x", "x_tilde", "disc_class", "disc_layer"), outs):
        _summ_close(g[f"dcgan/{k}"], v, f"dcgan {k}")
    dc.eval()
    torch.manual_seed(23)
    with torch.no_grad():
        _summ_close(g["dcgan/eval_x_p"], dc(x), "dcgan eval")
        gen = dc(None, 5)
    assert gen.shape == (5, 3, 64, 64)
    _summ_close(g["dcgan/gen5"], gen, "dcgan gen5")


# ----------------------------------------------------------------------------------------------------------------
def test_literal_stage3_loop_body_runs_on_engine():
    """train/train_vgan_stage3.py:324-411 (mode 'vae-gan'): cognitive encoder frozen, decoder + discriminator trained
    under the equilibrium gate, gradients clamped to +-1."""
    from oracle import vaegan_oracle as O
    _cfg64()
    import models.vae_gan as vg
    cfg = O.ArchCfg.px64()
    B, V, seed = 4, 4096, 2
    hp = O.GanHyper()
    data = O.synth_batch(B, cfg, n_voxels=V, seed=1234, steps=1)
    tsd = O.fill_state(O.vaegan_spec(cfg), seed, True)
    csd = O.fill_state(O.cognitive_encoder_spec(cfg, V), seed + 100, True)
    teacher = vg.VaeGan(device=DEV, z_size=128).to(DEV)
    teacher.load_state_dict(tsd)
    cog = vg.CognitiveEncoder(input_size=V, z_size=128).to(DEV)
    cog.load_state_dict({k[len("encoder."):]: v for k, v in csd.items()})
    model = vg.VaeGanCognitive(device=DEV, encoder=cog, decoder=teacher.decoder, discriminator=teacher.discriminator,
                               teacher_net=None, stage=3, z_size=128).to(DEV)
    mk = lambda p: torch.optim.RMSprop(params=p, lr=hp.lr, alpha=0.9, eps=1e-8, weight_decay=0, momentum=0,
                                       centered=False)
    opt_d, opt_s = mk(model.decoder.parameters()), mk(model.discriminator.parameters())
    fmri, image = data["fmri"].to(DEV), data["x"].to(DEV)
    nz = data["noise"][0]
    eps, z_p = nz[0].to(DEV), nz[1].to(DEV).requires_grad_(True)
    # ---- loop body
    model.train()
    for param in model.encoder.parameters():
        param.requires_grad = False
    for param in model.decoder.parameters():
        param.requires_grad = True
    for param in model.discriminator.parameters():
        param.requires_grad = True
    mus, lv = model.encoder(fmri)
    x_tilde = model.decoder(eps * torch.exp(0.5 * lv) + mus)
    x_gt = image
    x_p = model.decoder(z_p)
    disc_layer = model.discriminator(x_gt, x_tilde, x_p, "REC")
    disc_class = model.discriminator(x_gt, x_tilde, x_p, "GAN")
    nle, kld, mse, bo, bp, bs = vg.VaeGanCognitive.loss(x_gt, x_tilde, disc_layer[:B], disc_layer[B:-B], disc_layer[-B:],
                                                        disc_class[:B], disc_class[B:-B], disc_class[-B:], mus, lv)
    train_dis = True
    train_dec = True
    loss_encoder = torch.sum(kld) + torch.sum(mse)
    loss_discriminator = torch.sum(bo) + torch.sum(bp) + torch.sum(bs)
    loss_decoder = torch.sum(hp.lambda_mse * mse) - (1.0 - hp.lambda_mse) * loss_discriminator
    if torch.mean(bo).item() < hp.equilibrium - hp.margin or torch.mean(bp).item() < hp.equilibrium - hp.margin:
        train_dis = False
    if torch.mean(bo).item() > hp.equilibrium + hp.margin or torch.mean(bp).item() > hp.equilibrium + hp.margin:
        train_dec = False
    if train_dec is False and train_dis is False:
        train_dis = True
        train_dec = True
    model.zero_grad()
    grads = {}
    if train_dec:
        loss_decoder.backward(retain_graph=True)
        grads.update({"decoder." + k: p.grad.clone() for k, p in model.decoder.named_parameters()})
        [p.grad.data.clamp_(-1, 1) for p in model.decoder.parameters()]
        opt_d.step()
        model.discriminator.zero_grad()
    if train_dis:
        loss_discriminator.backward()
        grads.update({"discriminator." + k: p.grad.clone() for k, p in model.discriminator.named_parameters()})
        [p.grad.data.clamp_(-1, 1) for p in model.discriminator.parameters()]
        opt_s.step()
    assert all(p.grad is None or float(p.grad.abs().max()) == 0.0 for p in model.encoder.parameters())
    # ---- oracle
    w_init = {k: v.clone() for k, v in {**tsd, **csd}.items()}
    P = dict(csd)
    P.update({k: v for k, v in tsd.items() if k.startswith(("decoder.", "discriminator."))})
    opts = {n: O.OptState(kind="rmsprop", lr=hp.lr) for n in ("encoder", "decoder", "discriminator")}
    ref = O.stage3_step(P, opts, data["fmri"], data["x"], nz, cfg, V, keep_grads=True)
    assert train_dis == ref["logs"]["train_dis"] and train_dec == ref["logs"]["train_dec"]
    got = dict(loss_encoder=loss_encoder.item(), loss_discriminator=loss_discriminator.item(),
               loss_decoder=loss_decoder.item(), nle=nle.sum().item(), kl=kld.sum().item(), mse=mse.sum().item(),
               bce_orig=bo.sum().item(), bce_pred=bp.sum().item(), bce_samp=bs.sum().item())
    for k, v in got.items():
        print("stage3", k, v, ref["logs"][k], _rel(v, ref["logs"][k]))
        assert _rel(v, ref["logs"][k]) < 1e-3, (k, v, ref["logs"][k])
    refg = {k: v for k, v in ref["grads"].items() if v is not None and k in grads}
    assert refg
    # second reference: the oracle under its 16-bit storage model (ReLU masks closer to the engine's, tests/gradcheck.py)
    P16 = dict(O.fill_state(O.cognitive_encoder_spec(cfg, V), seed + 100, True))
    P16.update({k: v for k, v in O.fill_state(O.vaegan_spec(cfg), seed, True).items()
                if k.startswith(("decoder.", "discriminator."))})
    o16 = {n: O.OptState(kind="rmsprop", lr=hp.lr) for n in ("encoder", "decoder", "discriminator")}
    with gradcheck.storage16(O):
        ref16 = O.stage3_step(P16, o16, data["fmri"], data["x"], nz, cfg, V, keep_grads=True)
    gradcheck.check(grads, refg, {k: ref16["grads"][k] for k in refg}, "api stage3", tol16=None)
    sd = model.state_dict()
    pre = tuple(p for p, on in (("decoder.", train_dec), ("discriminator.", train_dis)) if on)
    _updates_agree(sd, P, w_init, pre, "api stage3")
    for k in ("encoder.fc1.0.weight", "encoder.l_mu.weight"):          # frozen: bit-identical
        assert torch.equal(sd[k].cpu(), w_init[k]), k


# ----------------------------------------------------------------------------------------------------------------
def _wae_cognitive_models(vg, O, cfg, V, seed):
    """Wiring of train_wae_stage2.py:195-202 / train_wae_stage3.py:208-221 on the recipe weights of
    tests/test_wae_gpu.py::_wae_state (same as make_golden.py::build_wae_cognitive)."""
    tsd = O.fill_state(O.encoder_spec(cfg) + O.decoder_spec(cfg) + O.wae_discriminator_spec(cfg), seed, True)
    teacher = vg.WaeGan(device=DEV, z_size=128).to(DEV)
    teacher.load_state_dict(tsd)
    cog = vg.CognitiveEncoder(input_size=V, z_size=128).to(DEV)
    cog.load_state_dict({k[len("encoder."):]: v for k, v in
                         O.fill_state(O.cognitive_encoder_spec(cfg, V), seed + 100, True).items()})
    model = vg.WaeGanCognitive(device=DEV, encoder=cog, decoder=teacher.decoder, z_size=128)
    model.discriminator.load_state_dict({k[len("discriminator."):]: v for k, v in
                                         O.fill_state(O.wae_discriminator_spec(cfg), seed + 200, True).items()})
    return teacher, model


@pytest.mark.parametrize("stage", [2, 3])
def test_literal_wae_stage23_loop_body_runs_on_engine(stage):
    """train/train_wae_stage2.py:276-328 (cognitive encoder + latent discriminator trained, decoder frozen) and
    train/train_wae_stage3.py:297-347 (decoder + latent discriminator trained, encoder frozen)."""
    from oracle import vaegan_oracle as O
    from test_wae_gpu import _wae_state
    _cfg64()
    import models.vae_gan as vg
    import torch.nn as nn
    cfg = O.ArchCfg.px64()
    B, V, seed = 4, 4096, 6 if stage == 2 else 7
    data = O.synth_batch(B, cfg, n_voxels=V, seed=1234, steps=1)
    teacher, model = _wae_cognitive_models(vg, O, cfg, V, seed)
    opt_e = torch.optim.Adam(model.encoder.parameters(), lr=0.001, betas=(0.5, 0.999))
    opt_d = torch.optim.Adam(model.decoder.parameters(), lr=0.001, betas=(0.5, 0.999))
    opt_s = torch.optim.Adam(model.discriminator.parameters(), lr=0.0005, betas=(0.5, 0.999))
    x_image, x_fmri = data["x"].to(DEV), data["fmri"].to(DEV)
    if stage == 2:
        _freeze(teacher.decoder, True)
    else:
        _freeze(teacher.encoder, True)
        _freeze(model.encoder, True)
    # ---- loop body
    model.train()
    teacher.train()
    if stage == 2:
        _freeze(model.decoder, True)
        model.encoder.zero_grad()
        model.discriminator.zero_grad()
        z, _ = teacher.encoder(x_image)
        teacher.decoder(z)                            # x_gt: unused by the script, moves BN statistics
        _freeze(model.encoder, True)
    else:
        _freeze(model.encoder, True)
        model.decoder.zero_grad()
        model.discriminator.zero_grad()
        _freeze(model.decoder, True)
    _freeze(model.discriminator, False)
    z_fake, var = model.encoder(x_fmri)
    z_real, var = teacher.encoder(x_image)
    d_real = model.discriminator(z_real)
    d_fake = model.discriminator(z_fake)
    l_fake = -10 * torch.sum(torch.log(d_fake + 1e-3))
    l_real = -10 * torch.sum(torch.log(1 - d_real + 1e-3))
    l_fake.backward(retain_graph=True)
    l_real.backward(retain_graph=True)
    opt_s.step()
    if stage == 2:
        _freeze(model.encoder, False)
    else:
        _freeze(model.decoder, False)
    _freeze(model.discriminator, True)
    z_real, var = model.encoder(x_fmri)
    x_recon = model.decoder(z_real)
    d_real = model.discriminator(z_real)
    l_rec = nn.MSELoss()(x_recon, x_image)
    l_pen = -10 * torch.mean(torch.log(d_real + 1e-3))
    l_rec.backward(retain_graph=True)
    if stage == 2:
        l_pen.backward()
        opt_e.step()
    else:
        opt_d.step()
    # ---- oracle
    P = _wae_state(O, cfg, stage, V, seed)
    w_init = {k: v.clone() for k, v in P.items()}
    opts = {"encoder": O.OptState(kind="adam", lr=1e-3), "decoder": O.OptState(kind="adam", lr=1e-3),
            "discriminator": O.OptState(kind="adam", lr=5e-4)}
    fn = O.wae_stage2_step if stage == 2 else O.wae_stage3_step
    ref = fn(P, opts, data["fmri"], data["x"], cfg, V)
    got = dict(loss_reconstruction=l_rec.item(), loss_penalty=l_pen.item(), loss_discriminator_fake=l_fake.item(),
               loss_discriminator_real=l_real.item())
    for k, v in got.items():
        tol = 5e-3 if k == "loss_penalty" else 1e-3
        print(f"wae{stage}", k, v, ref["logs"][k], _rel(v, ref["logs"][k]))
        assert _rel(v, ref["logs"][k]) < tol, (k, v, ref["logs"][k])
    sd = dict(model.state_dict())
    sd.update({"teacher_net.encoder." + k: v for k, v in teacher.encoder.state_dict().items()})
    _updates_agree(sd, P, w_init, ("encoder.",) if stage == 2 else ("decoder.",), f"api wae{stage}")
    # frozen sub-networks did not move; every BatchNorm counter follows the reference's call count
    frozen = "decoder.fc.0.weight" if stage == 2 else "encoder.fc1.0.weight"
    assert torch.equal(sd[frozen].cpu(), w_init[frozen]), frozen
    for k, v in P.items():
        if "num_batches" in k:
            assert int(sd[k]) == int(v), (k, int(sd[k]), int(v))


# ----------------------------------------------------------------------------------------------------------------
def test_literal_dual_stage1_loop_body_runs_on_engine():
    """train/wae_vgan_stage1.py:290-441 (mode 'vae-gan'): the Stage-I VAE/GAN forward and losses, the WAE
    latent-discriminator phase on the encoder means, the penalty back-propagated into the encoder from a third encoder
    pass, then the three gated updates with the penalty gradient still in the encoder's ``.grad``."""
    from oracle import vaegan_oracle as O
    _cfg64()
    import models.vae_gan as vg
    cfg = O.ArchCfg.px64()
    B, seed, lam = 4, 8, 1.0
    hp = O.GanHyper()
    data = O.synth_batch(B, cfg, seed=1234, steps=1)
    sd0 = O.fill_state(O.vaegan_spec(cfg), seed, True)
    wd0 = O.fill_state(O.wae_discriminator_spec(cfg, pre="wae_discriminator."), seed + 200, True)
    model = vg.VaeGan(device=DEV, z_size=128).to(DEV)
    model.load_state_dict(sd0)
    model_wae = vg.WaeGan(device=DEV, z_size=128).to(DEV)
    model_wae.discriminator.load_state_dict({k[len("wae_discriminator."):]: v for k, v in wd0.items()})
    mk = lambda p: torch.optim.RMSprop(params=p, lr=hp.lr, alpha=0.9, eps=1e-8, weight_decay=0, momentum=0,
                                       centered=False)
    opt_e, opt_d, opt_s = mk(model.encoder.parameters()), mk(model.decoder.parameters()), mk(
        model.discriminator.parameters())
    opt_w = mk(model_wae.discriminator.parameters())
    x = data["x"].to(DEV)
    nz = data["noise"][0]
    eps, z_p = nz[0].to(DEV), nz[1].to(DEV).requires_grad_(True)
    # ---- loop body
    model.train()
    mus, lv = model.encoder(x)
    x_tilde = model.decoder(eps * torch.exp(0.5 * lv) + mus)
    x_p = model.decoder(z_p)
    disc_layer = model.discriminator(x, x_tilde, x_p, "REC")
    disc_class = model.discriminator(x, x_tilde, x_p, "GAN")
    nle, kld, mse, bo, bp, bs = vg.VaeGan.loss(x, x_tilde, disc_layer[:B], disc_layer[B:-B], disc_layer[-B:],
                                               disc_class[:B], disc_class[B:-B], disc_class[-B:], mus, lv)
    train_enc, train_dis, train_dec = True, True, True
    loss_encoder = torch.sum(kld) + torch.sum(mse)
    loss_discriminator = torch.sum(bo) + torch.sum(bp) + torch.sum(bs)
    loss_decoder = torch.sum(hp.lambda_mse * mse) - (1.0 - hp.lambda_mse) * loss_discriminator
    if torch.mean(bo).item() < hp.equilibrium - hp.margin or torch.mean(bp).item() < hp.equilibrium - hp.margin:
        train_dis = False
    if torch.mean(bo).item() > hp.equilibrium + hp.margin or torch.mean(bp).item() > hp.equilibrium + hp.margin:
        train_dec = False
    if train_dec is False and train_dis is False:
        train_dis = True
        train_dec = True
    model.encoder.zero_grad()
    model.decoder.zero_grad()
    model.discriminator.zero_grad()
    # WAE discriminator phase (:378-397)
    _freeze(model.decoder, True)
    _freeze(model.encoder, True)
    _freeze(model_wae.discriminator, False)
    z_real, var = model.encoder(x)
    z_fake = (nz[2] * 0.5).to(DEV)
    d_real = model_wae.discriminator(z_real)
    d_fake = model_wae.discriminator(z_fake)
    l_fake = -lam * torch.sum(torch.log(d_fake + 1e-3))
    l_real = -lam * torch.sum(torch.log(1 - d_real + 1e-3))
    l_fake.backward(retain_graph=True)
    l_real.backward(retain_graph=True)
    opt_w.step()
    # generator phase (:401-417)
    _freeze(model.encoder, False)
    _freeze(model.decoder, False)
    _freeze(model_wae.discriminator, True)
    z_real, var = model.encoder(x)
    x_recon = model.decoder(z_real)
    d_real = model_wae.discriminator(z_real)
    l_pen = -lam * torch.sum(torch.log(d_real + 1e-3))
    l_pen.backward()
    opt_d.step()                                       # :417 -- no decoder gradient exists yet: a no-op on step 1
    # VAE/GAN updates (:419-441)
    if train_enc:
        loss_encoder.backward(retain_graph=True)
        g_enc = {"encoder." + k: p.grad.clone() for k, p in model.encoder.named_parameters()}
        opt_e.step()
        model.zero_grad()
    g_dec, g_dis = {}, {}
    if train_dec:
        loss_decoder.backward(retain_graph=True)
        g_dec = {"decoder." + k: p.grad.clone() for k, p in model.decoder.named_parameters()}
        opt_d.step()
        model.discriminator.zero_grad()
    if train_dis:
        loss_discriminator.backward()
        g_dis = {"discriminator." + k: p.grad.clone() for k, p in model.discriminator.named_parameters()}
        opt_s.step()
    # ---- oracle
    P = O.fill_state(O.vaegan_spec(cfg), seed, True)
    P.update(O.fill_state(O.wae_discriminator_spec(cfg, pre="wae_discriminator."), seed + 200, True))
    w_init = {k: v.clone() for k, v in P.items()}
    opts = {n: O.OptState(kind="rmsprop", lr=hp.lr) for n in ("encoder", "decoder", "discriminator",
                                                                 "wae_discriminator")}
    ref = O.dual_stage1_step(P, opts, data["x"], nz, cfg, lam=lam, keep_grads=True)
    assert train_dis == ref["logs"]["train_dis"] and train_dec == ref["logs"]["train_dec"]
    got = dict(loss_encoder=loss_encoder.item(), loss_decoder=loss_decoder.item(),
               loss_discriminator=loss_discriminator.item(), nle=nle.sum().item(), kl=kld.sum().item(),
               mse=mse.sum().item(), bce_orig=bo.sum().item(), bce_pred=bp.sum().item(), bce_samp=bs.sum().item(),
               loss_penalty=l_pen.item(), loss_discriminator_fake=l_fake.item(), loss_discriminator_real=l_real.item())
    for k, v in got.items():
        tol = 5e-3 if k == "loss_penalty" else 1e-3
        print("dual", k, v, ref["logs"][k], _rel(v, ref["logs"][k]))
        assert _rel(v, ref["logs"][k]) < tol, (k, v, ref["logs"][k])
    # the encoder gradient carries loss_encoder AND the penalty (:421 accumulates onto :413)
    grads = {**g_enc, **g_dec, **g_dis}
    refg = {k: v for k, v in ref["grads"].items() if v is not None and k in grads}
    P16 = O.fill_state(O.vaegan_spec(cfg), seed, True)
    P16.update(O.fill_state(O.wae_discriminator_spec(cfg, pre="wae_discriminator."), seed + 200, True))
    o16 = {n: O.OptState(kind="rmsprop", lr=hp.lr) for n in ("encoder", "decoder", "discriminator",
                                                                "wae_discriminator")}
    with gradcheck.storage16(O):
        ref16 = O.dual_stage1_step(P16, o16, data["x"], nz, cfg, lam=lam, keep_grads=True)
    gradcheck.check(grads, refg, {k: ref16["grads"][k] for k in refg}, "api dual", tol16=None)
    sd = dict(model.state_dict())
    sd.update({"wae_discriminator." + k: v for k, v in model_wae.discriminator.state_dict().items()})
    pre = ("encoder.",) + tuple(p for p, on in (("decoder.", train_dec), ("discriminator.", train_dis)) if on)
    _updates_agree(sd, P, w_init, pre, "api dual")
    for k, v in P.items():
        if "num_batches" in k:
            assert int(sd[k]) == int(v), (k, int(sd[k]), int(v))


@pytest.mark.parametrize("level", [1, 2])
def test_discriminator_recon_level_matches_oracle(level):
    """``Discriminator(recon_level=1 | 2)`` (models/vae_gan.py:139-173): the 'REC' call returns the RAW convolution output
    of block ``recon_level`` and stops there -- the blocks above it see no forward, no running-statistics update and no
    gradient -- while a 'GAN' call runs the whole stack.  Features, their gradients w.r.t. the predicted images and the
    parameters, the class probabilities and the BatchNorm counters against the CPU oracle."""
    from oracle import vaegan_oracle as O
    _cfg64()
    import models.vae_gan as M
    cfg_o = O.ArchCfg.px64()
    B = 4
    rs = np.random.RandomState(77 + level)
    xs = [torch.from_numpy(rs.uniform(-1, 1, (B, 3, 64, 64)).astype(np.float32)) for _ in range(3)]
    P = O.fill_state(O.discriminator_spec(cfg_o, ""), 21, True)
    dis = M.Discriminator(channel_in=3, recon_level=level).to(DEV)
    dis.load_state_dict({k: v.clone() for k, v in P.items()})
    dis.train()
    xo, xp, xq = (t.to(DEV) for t in xs)
    xp.requires_grad_(True)
    feat = dis(xo, xp, xq, "REC")
    w = torch.from_numpy(rs.standard_normal(tuple(feat.shape)).astype(np.float32))
    (feat * w.to(DEV)).sum().backward()
    # oracle: same call under autograd
    Po = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and "running" not in k else v.clone())
          for k, v in P.items()}
    xpo = xs[1].clone().requires_grad_(True)
    ref = O.discriminator_fwd(Po, "", xs[0], xpo, xs[2], "REC", cfg_o, True, recon_level=level)
    (ref * w).sum().backward()
    assert tuple(feat.shape) == tuple(ref.shape)
    err = ((feat.detach().cpu() - ref.detach()).norm() / ref.detach().norm()).item()
    assert err < 5e-3, err
    gx = ((xp.grad.cpu() - xpo.grad).norm() / xpo.grad.norm()).item()
    assert gx < 3e-2, gx
    got = dict(dis.named_parameters())
    for k, v in Po.items():
        if not (torch.is_tensor(v) and v.requires_grad):
            continue
        blk = int(k.split(".")[1]) if k.startswith("conv.") else 99
        if blk > level or v.grad is None:
            # never reached by a REC call (blocks above the level; the level block's own BatchNorm, whose output the
            # call throws away)
            assert v.grad is None or float(v.grad.abs().max()) == 0.0
            assert got[k].grad is None or float(got[k].grad.abs().max()) == 0.0, k
            continue
        g, r = got[k].grad.cpu().double().reshape(-1), v.grad.double().reshape(-1)
        cos = float(g @ r / (g.norm() * r.norm() + 1e-30))
        assert cos > 0.98 and abs(float(g.norm() / r.norm()) - 1) < 0.05, (k, cos, float(g.norm() / r.norm()))
    sd = dis.state_dict()
    for i in (1, 2, 3):
        assert int(sd[f"conv.{i}.bn.num_batches_tracked"]) == (1 if i <= level else 0), i
        assert int(Po[f"conv.{i}.bn.num_batches_tracked"]) == (1 if i <= level else 0), i
    # a GAN call on the same tensors runs every block (and is not served from the REC call's conv stack)
    prob = dis(xo, xp.detach(), xq, "GAN")
    refp = O.discriminator_fwd(Po, "", xs[0], xs[1], xs[2], "GAN", cfg_o, True)
    assert float((prob.detach().cpu() - refp.detach()).abs().max()) < 3e-3
    sd = dis.state_dict()
    for i in (1, 2, 3):
        assert int(sd[f"conv.{i}.bn.num_batches_tracked"]) == (2 if i <= level else 1), i


def test_discriminator_recon_level_outside_the_reference_range_is_refused():
    """Level 0 raises a TypeError in the reference (its block 0 is an nn.Sequential called with two arguments), levels
    above 3 make its forward return None: the engine refuses both at construction."""
    _cfg64()
    import models.vae_gan as M
    for level in (0, 4):
        with pytest.raises(ValueError):
            M.Discriminator(channel_in=3, recon_level=level).to(DEV)._engine()


def test_discriminator_recon_levels_match_reference_golden(golden_dir):
    """The same calls on the engine against the numbers the REAL reference produced (tests/golden/recon_b4.npz,
    make_golden.py::case_recon): 'REC' features of recon_level 1 and 2, the gradient w.r.t. the predicted images, the
    parameter gradients' norms, the BatchNorm counters after the REC call and after a GAN call on the same tensors."""
    from oracle import vaegan_oracle as O
    _cfg64()
    import models.vae_gan as M
    cfg_o = O.ArchCfg.px64()
    g = np.load(os.path.join(golden_dir, "recon_b4.npz"))
    B, seed = int(g["meta/B"]), int(g["meta/seed"])
    rs = np.random.RandomState(1000 + seed)
    xs = [torch.from_numpy(rs.uniform(-1, 1, (B, 3, 64, 64)).astype(np.float32)) for _ in range(3)]
    for level in (1, 2):
        tag = f"level{level}"
        dis = M.Discriminator(channel_in=3, recon_level=level).to(DEV)
        dis.load_state_dict({k: v.clone() for k, v in O.fill_state(O.discriminator_spec(cfg_o, ""), seed, True).items()})
        dis.train()
        xo, xp, xq = (t.to(DEV) for t in xs)
        xp.requires_grad_(True)
        feat = dis(xo, xp, xq, "REC")
        assert list(feat.shape) == [int(v) for v in g[f"{tag}/shape"]]
        w = torch.from_numpy(np.random.RandomState(2000 + level).standard_normal(tuple(feat.shape)).astype(np.float32))
        (feat * w.to(DEV)).sum().backward()
        _summ_close(g[f"{tag}/feat"], feat, f"{tag} feat disc_layer")
        got = O.tensor_summary(xp.grad.detach().float().cpu())
        ref = g[f"{tag}/dxp"]
        assert abs(got[0] - ref[0]) < 3e-2 * abs(ref[0]), (tag, "dxp norm", got[0], ref[0])
        params = dict(dis.named_parameters())
        for k, r in zip([str(k) for k in g[f"{tag}/grad_keys"]], g[f"{tag}/grad_sum"]):
            gk = params[k].grad
            if np.isnan(r[0]):
                assert gk is None or float(gk.abs().max()) == 0.0, k
            elif params[k].numel() >= 64:
                n = float(gk.double().norm())
                assert abs(n - r[0]) < 6e-2 * abs(r[0]), (tag, k, n, r[0])      # ReLU-mask noise of 16-bit activations
        sd = dis.state_dict()
        assert [int(sd[f"conv.{i}.bn.num_batches_tracked"]) for i in (1, 2, 3)] == [int(v) for v in g[f"{tag}/nbt_rec"]]
        prob = dis(xo, xp.detach(), xq, "GAN")
        _summ_close(g[f"{tag}/prob"], prob, f"{tag} prob", ntol=2e-3, etol=1e-2)
        sd = dis.state_dict()
        assert [int(sd[f"conv.{i}.bn.num_batches_tracked"]) for i in (1, 2, 3)] == [int(v) for v in g[f"{tag}/nbt_gan"]]
