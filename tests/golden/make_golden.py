#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz by RUNNING THE REAL REFERENCE.

Run in the build container only (needs /root/reference; that tree never travels):

    python tests/golden/make_golden.py            # writes tests/golden/<case>.npz

What it does
  * imports ``models.vae_gan`` / ``configs.models_config`` from /root/reference (torchvision is stubbed:
    only ``ResNetEncoder`` uses it), switches the config module to the wanted resolution *before* import,
  * loads the deterministic parameter recipe (oracle.fill_state) into the reference nn.Modules,
  * checks once per case that calling the sub-modules in the documented order with explicit noise equals
    the reference's own ``model(x)`` under a fixed torch seed (RNG draw order),
  * runs the inline step bodies of the training scripts (restated here because the scripts cannot be
    imported: tensorboard/torchvision/nibabel are missing and their optimizer ordering needs torch<=1.4):
    literal ``loss.backward()`` x3 at the pre-update weights, then ``torch.optim`` steps,
  * stores compact fingerprints: logged losses, gate flags, per-tensor [norm,sum,first8,last8] of every
    gradient, every post-step state_dict entry and every forward output.

Fixtures are data only (numbers); no reference source is stored.
"""
import importlib
import os
import sys
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")

from oracle import vaegan_oracle as O  # noqa: E402  (recipe + fingerprints only)

REF = "/root/reference"


def load_reference(cfg: O.ArchCfg):
    """(Re-)import the reference model module under architecture ``cfg``."""
    if REF not in sys.path:
        sys.path.insert(0, REF)
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tv.models = tvm
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = tvm
    for m in ("models.vae_gan", "models", "configs.models_config", "configs"):
        sys.modules.pop(m, None)
    mc = importlib.import_module("configs.models_config")
    mc.image_size = cfg.image_size
    mc.fc_input = cfg.fc_input
    mc.fc_output = cfg.fc_output
    mc.fc_input_gan = cfg.fc_input_gan
    mc.fc_output_gan = cfg.fc_output_gan
    mc.stride_gan = cfg.stride_gan
    mc.latent_dim = cfg.latent_dim
    mc.output_pad_dec = list(cfg.output_pad_dec)
    mc.decoder_channels = list(cfg.decoder_channels)
    mc.encoder_channels = list(cfg.encoder_channels)
    mc.discrim_channels = list(cfg.discrim_channels)
    return importlib.import_module("models.vae_gan")


def summarize_state(sd):
    keys = list(sd.keys())
    return keys, np.stack([O.tensor_summary(sd[k] if sd[k].is_floating_point() else sd[k].float()) for k in keys])


def grads_of(params_named):
    return {k: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for k, p in params_named}


def pack(out, prefix, dct):
    for k, v in dct.items():
        out[f"{prefix}/{k}"] = np.asarray(v)


def literal_three_backward(model, loss_enc, loss_dec, loss_dis, nets):
    """train_vgan_stage1.py:408-432 with optimizer steps deferred."""
    g = {}
    model.zero_grad()
    if "encoder" in nets:
        loss_enc.backward(retain_graph=True)
        g.update(grads_of([("encoder." + k, p) for k, p in model.encoder.named_parameters()]))
        model.zero_grad()
    if "decoder" in nets:
        loss_dec.backward(retain_graph=True)
        g.update(grads_of([("decoder." + k, p) for k, p in model.decoder.named_parameters()]))
        model.discriminator.zero_grad()
    if "discriminator" in nets:
        loss_dis.backward()
        g.update(grads_of([("discriminator." + k, p) for k, p in model.discriminator.named_parameters()]))
    return g


def apply_grads(module, prefix, g, opt, clamp=None):
    for k, p in module.named_parameters():
        p.grad = g[prefix + k].clone()
        if clamp is not None:
            p.grad.data.clamp_(-clamp, clamp)
    opt.step()


def gate(bo, bp, hp):
    train_dis, train_dec = True, True
    if bo < hp.equilibrium - hp.margin or bp < hp.equilibrium - hp.margin:
        train_dis = False
    if bo > hp.equilibrium + hp.margin or bp > hp.equilibrium + hp.margin:
        train_dec = False
    if train_dec is False and train_dis is False:
        train_dis = True
        train_dec = True
    return train_dis, train_dec


def rms(params, lr):
    return torch.optim.RMSprop(params=params, lr=lr, alpha=0.9, eps=1e-8, weight_decay=0, momentum=0, centered=False)


def record_step(out, tag, logs, fw, grads):
    pack(out, f"{tag}/logs", {k: np.float64(v) for k, v in logs.items()})
    pack(out, f"{tag}/fw", {k: O.tensor_summary(v) for k, v in fw.items()})
    out[f"{tag}/grad_keys"] = np.array(list(grads.keys()))
    out[f"{tag}/grad_sum"] = np.stack([O.tensor_summary(v) for v in grads.values()])


# ------------------------------------------------------------------------------------------------
def case_stage1(name, cfg, B, seed, perturb, steps=2, mode="vae-gan", beta=1.0):
    vg = load_reference(cfg)
    hp = O.GanHyper()
    model = vg.VaeGan(device="cpu", z_size=cfg.latent_dim)
    sd0 = O.fill_state(O.vaegan_spec(cfg), seed, perturb)
    assert list(model.state_dict().keys()) == list(sd0.keys())
    model.load_state_dict(sd0)
    model.train()
    data = O.synth_batch(B, cfg, seed=1234, steps=steps)
    x = data["x"]

    # RNG-order check against the reference's own forward
    chk = vg.VaeGan(device="cpu", z_size=cfg.latent_dim)
    chk.load_state_dict(sd0)
    chk.train()
    torch.manual_seed(7)
    ref_out = chk(x)
    torch.manual_seed(7)
    eps_t = torch.empty(B, cfg.latent_dim).normal_()
    zp_t = torch.randn(B, cfg.latent_dim)
    chk2 = vg.VaeGan(device="cpu", z_size=cfg.latent_dim)
    chk2.load_state_dict(sd0)
    chk2.train()
    mus, lv = chk2.encoder(x)
    xt = chk2.decoder(eps_t * torch.exp(0.5 * lv) + mus)
    xp = chk2.decoder(zp_t)
    dl = chk2.discriminator(x, xt, xp, "REC")
    dc = chk2.discriminator(x, xt, xp, "GAN")
    for a, b in zip(ref_out, (xt, dc, dl, mus, lv)):
        assert torch.equal(a, b), "sub-module order does not reproduce model(x)"

    opt_e = rms(model.encoder.parameters(), hp.lr)
    opt_d = rms(model.decoder.parameters(), hp.lr)
    opt_s = rms(model.discriminator.parameters(), hp.lr)
    out = {"meta/case": np.array("stage1"), "meta/B": B, "meta/seed": seed, "meta/perturb": perturb,
           "meta/steps": steps, "meta/image_size": cfg.image_size, "meta/mode": np.array(mode), "meta/beta": beta}
    for s in range(steps):
        eps, z_p = data["noise"][s, 0], data["noise"][s, 1]
        mus, lv = model.encoder(x)
        z = eps * torch.exp(0.5 * lv) + mus
        x_tilde = model.decoder(z)
        x_p = model.decoder(z_p)
        disc_layer = model.discriminator(x, x_tilde, x_p, "REC")
        disc_class = model.discriminator(x, x_tilde, x_p, "GAN")
        nle, kld, mse, bo, bp, bs = vg.VaeGan.loss(x, x_tilde, disc_layer[:B], disc_layer[B:-B], disc_layer[-B:],
                                                   disc_class[:B], disc_class[B:-B], disc_class[-B:], mus, lv)
        # loss compositions of train_vgan_stage1.py:359-388, flags and gate :353-357, :396-404
        train_enc, train_dis, train_dec = True, True, True
        if mode == "beta-vae":
            loss_enc = torch.sum(kld) * beta * (1 / B) + torch.sum(mse)
            loss_dis = torch.sum(bo) + torch.sum(bp) + torch.sum(bs)
            loss_dec = torch.sum(hp.lambda_mse * mse) - (1.0 - hp.lambda_mse) * loss_dis
        elif mode == "vae-gan":
            loss_enc = torch.sum(kld) + torch.sum(mse)
            loss_dis = torch.sum(bo) + torch.sum(bp) + torch.sum(bs)
            loss_dec = torch.sum(hp.lambda_mse * mse) - (1.0 - hp.lambda_mse) * loss_dis
        elif mode == "dcgan":
            train_enc = False
            for param in model.encoder.parameters():
                param.requires_grad = False
            loss_enc = torch.sum(kld) + torch.sum(nle)
            loss_dis = torch.sum(bo) + torch.sum(bs)
            loss_dec = torch.sum(hp.lambda_mse * nle) - (1.0 - hp.lambda_mse) * loss_dis
        elif mode == "vae":
            loss_enc = torch.sum(kld) + torch.sum(nle)
            loss_dis = torch.sum(bo) + torch.sum(bs)
            loss_dec = torch.sum(hp.lambda_mse * nle)
            train_dis = False
        if torch.mean(bo).item() < hp.equilibrium - hp.margin or torch.mean(bp).item() < hp.equilibrium - hp.margin:
            train_dis = False
        if torch.mean(bo).item() > hp.equilibrium + hp.margin or torch.mean(bp).item() > hp.equilibrium + hp.margin:
            train_dec = False
        if train_dec is False and train_dis is False:
            train_dis = True
            train_dec = True
        nets = (("encoder",) if train_enc else ()) + ("decoder", "discriminator")
        g = literal_three_backward(model, loss_enc, loss_dec, loss_dis, nets)
        if train_enc:
            apply_grads(model.encoder, "encoder.", g, opt_e)
        if train_dec:
            apply_grads(model.decoder, "decoder.", g, opt_d)
        if train_dis:
            apply_grads(model.discriminator, "discriminator.", g, opt_s)
        model.zero_grad()
        logs = dict(loss_encoder=loss_enc.item(), loss_discriminator=loss_dis.item(), loss_decoder=loss_dec.item(),
                    nle=torch.sum(nle).item(), kl=torch.sum(kld).item(), mse=torch.sum(mse).item(),
                    bce_orig=torch.sum(bo).item(), bce_pred=torch.sum(bp).item(), bce_samp=torch.sum(bs).item(),
                    train_dis=float(train_dis), train_dec=float(train_dec))
        fw = dict(x_tilde=x_tilde, x_p=x_p, disc_class=disc_class, disc_layer=disc_layer, mus=mus, log_variances=lv)
        record_step(out, f"step{s}", logs, fw, g)
        keys, summ = summarize_state(model.state_dict())
        out[f"step{s}/state_keys"] = np.array(keys)
        out[f"step{s}/state_sum"] = summ
        print(name, "step", s, {k: round(v, 5) for k, v in logs.items()})
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


def build_cognitive(vg, cfg, V, seed, perturb, stage, mode="vae-gan"):
    """Model wiring of train_vgan_stage2.py:211-232 / train_vgan_stage3.py:222-245 on recipe weights (``mode`` 'vae' at
    Stage II: no teacher net, train_vgan_stage2.py:234-238)."""
    teacher = vg.VaeGan(device="cpu", z_size=cfg.latent_dim)
    teacher.load_state_dict(O.fill_state(O.vaegan_spec(cfg), seed, perturb))
    cog = vg.CognitiveEncoder(input_size=V, z_size=cfg.latent_dim)
    cog.load_state_dict({k[len("encoder."):]: v for k, v in
                         O.fill_state(O.cognitive_encoder_spec(cfg, V), seed + 100, perturb).items()})
    if stage == 2:
        for p in teacher.decoder.parameters():
            p.requires_grad = False
        model = vg.VaeGanCognitive(device="cpu", encoder=cog, decoder=teacher.decoder,
                                   discriminator=teacher.discriminator, teacher_net=None if mode == "vae" else teacher,
                                   stage=2, z_size=cfg.latent_dim)
    else:
        # stage 3 builds fresh Decoder/Discriminator and loads stage-II weights; equivalent: reuse tensors
        for p in cog.parameters():
            p.requires_grad = False
        model = vg.VaeGanCognitive(device="cpu", encoder=cog, decoder=teacher.decoder,
                                   discriminator=teacher.discriminator, teacher_net=None, stage=3,
                                   z_size=cfg.latent_dim)
    model.train()
    return model


def case_cognitive(name, cfg, B, V, seed, perturb, stage, steps=2, mode="vae-gan"):
    """``mode``: 'vae-gan' or 'vae' (train_vgan_stage2.py:355-366, train_vgan_stage3.py:361-374)."""
    vg = load_reference(cfg)
    hp = O.GanHyper()
    teach = stage == 2 and mode != "vae"
    _build = build_cognitive
    build_cognitive_m = lambda *a: _build(*a, mode)
    model = build_cognitive_m(vg, cfg, V, seed, perturb, stage)
    data = O.synth_batch(B, cfg, n_voxels=V, seed=1234, steps=steps)
    x, fmri = data["x"], data["fmri"]

    # RNG-order check against the reference's own forward (eps_cog, [eps_teacher], z_p)
    chk = build_cognitive_m(vg, cfg, V, seed, perturb, stage)
    torch.manual_seed(11)
    ref_out = chk({"fmri": fmri, "image": x})
    torch.manual_seed(11)
    e1 = torch.empty(B, cfg.latent_dim).normal_()
    e2 = torch.empty(B, cfg.latent_dim).normal_() if teach else None
    zp = torch.randn(B, cfg.latent_dim)
    chk2 = build_cognitive_m(vg, cfg, V, seed, perturb, stage)
    mus, lv = chk2.encoder(fmri)
    xt = chk2.decoder(e1 * torch.exp(0.5 * lv) + mus)
    gt = x
    if teach:
        mt, lt = chk2.teacher_net.encoder(x)
        gt = chk2.decoder(e2 * torch.exp(0.5 * lt) + mt)
    xp = chk2.decoder(zp)
    dl = chk2.discriminator(gt, xt, xp, "REC")
    dc = chk2.discriminator(gt, xt, xp, "GAN")
    for a, b in zip(ref_out, (gt, xt, dc, dl, mus, lv)):
        assert torch.equal(a, b), "sub-module order does not reproduce model(sample)"

    opt_e = rms(model.encoder.parameters(), hp.lr)
    opt_d = rms(model.decoder.parameters(), hp.lr)
    opt_s = rms(model.discriminator.parameters(), hp.lr)
    out = {"meta/case": np.array(f"stage{stage}"), "meta/B": B, "meta/V": V, "meta/seed": seed,
           "meta/perturb": perturb, "meta/steps": steps, "meta/image_size": cfg.image_size, "meta/mode": np.array(mode)}
    for s in range(steps):
        nz = data["noise"][s]
        if stage == 2:
            for p in model.decoder.parameters():
                p.requires_grad = False
        else:
            for p in model.encoder.parameters():
                p.requires_grad = False
            for p in model.decoder.parameters():
                p.requires_grad = True
            for p in model.discriminator.parameters():
                p.requires_grad = True
        mus, lv = model.encoder(fmri)
        x_tilde = model.decoder(nz[0] * torch.exp(0.5 * lv) + mus)
        gt_x = x
        if teach:
            for p in model.teacher_net.encoder.parameters():
                p.requires_grad = False
            mt, lt = model.teacher_net.encoder(x)
            gt_x = model.decoder(nz[2] * torch.exp(0.5 * lt) + mt)
        x_p = model.decoder(nz[1])
        disc_layer = model.discriminator(gt_x, x_tilde, x_p, "REC")
        disc_class = model.discriminator(gt_x, x_tilde, x_p, "GAN")
        nle, kld, mse, bo, bp, bs = vg.VaeGanCognitive.loss(gt_x, x_tilde, disc_layer[:B], disc_layer[B:-B],
                                                            disc_layer[-B:], disc_class[:B], disc_class[B:-B],
                                                            disc_class[-B:], mus, lv)
        if mode == "vae":
            loss_enc = torch.sum(kld) + torch.sum(nle)
            loss_dis = torch.sum(bo) + torch.sum(bs)
            loss_dec = torch.sum(hp.lambda_mse * nle)
        else:
            loss_enc = torch.sum(kld) + torch.sum(mse)
            loss_dis = torch.sum(bo) + torch.sum(bp) + torch.sum(bs)
            loss_dec = torch.sum(hp.lambda_mse * mse) - (1.0 - hp.lambda_mse) * loss_dis
        if stage == 2:
            # train_vgan_stage2.py:366 sets train_dis = False for 'vae', :375-376 then overwrite both flags
            train_dis, train_dec = True, False
            g = literal_three_backward(model, loss_enc, loss_dec, loss_dis, ("encoder", "discriminator"))
            apply_grads(model.encoder, "encoder.", g, opt_e, clamp=1.0)
            apply_grads(model.discriminator, "discriminator.", g, opt_s, clamp=1.0)
        else:
            # train_vgan_stage3.py:356-357 (both True), :374 ('vae': train_dis = False), gate :382-389
            train_dis, train_dec = mode != "vae", True
            bom, bpm = torch.mean(bo).item(), torch.mean(bp).item()
            if bom < hp.equilibrium - hp.margin or bpm < hp.equilibrium - hp.margin:
                train_dis = False
            if bom > hp.equilibrium + hp.margin or bpm > hp.equilibrium + hp.margin:
                train_dec = False
            if train_dec is False and train_dis is False:
                train_dis, train_dec = True, True
            g = literal_three_backward(model, loss_enc, loss_dec, loss_dis, ("decoder", "discriminator"))
            if train_dec:
                apply_grads(model.decoder, "decoder.", g, opt_d, clamp=1.0)
            if train_dis:
                apply_grads(model.discriminator, "discriminator.", g, opt_s, clamp=1.0)
        model.zero_grad()
        logs = dict(loss_encoder=loss_enc.item(), loss_discriminator=loss_dis.item(), loss_decoder=loss_dec.item(),
                    nle=torch.sum(nle).item(), kl=torch.sum(kld).item(), mse=torch.sum(mse).item(),
                    bce_orig=torch.sum(bo).item(), bce_pred=torch.sum(bp).item(), bce_samp=torch.sum(bs).item(),
                    train_dis=float(train_dis), train_dec=float(train_dec))
        fw = dict(gt_x=gt_x, x_tilde=x_tilde, x_p=x_p, disc_class=disc_class, disc_layer=disc_layer, mus=mus,
                  log_variances=lv)
        record_step(out, f"step{s}", logs, fw, g)
        keys, summ = summarize_state(model.state_dict())
        out[f"step{s}/state_keys"] = np.array(keys)
        out[f"step{s}/state_sum"] = summ
        print(name, "step", s, {k: round(v, 5) for k, v in logs.items()})
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


def case_wae1(name, cfg, B, seed, steps=2):
    """train/train_wae_stage1.py:259-311 on the reference WaeGan container."""
    vg = load_reference(cfg)
    model = vg.WaeGan(device="cpu", z_size=cfg.latent_dim)
    spec = O.encoder_spec(cfg) + O.decoder_spec(cfg) + O.wae_discriminator_spec(cfg)
    sd0 = O.fill_state(spec, seed, False)
    assert list(model.state_dict().keys()) == list(sd0.keys())
    model.load_state_dict(sd0)
    lr = 1e-4
    opt_e = torch.optim.Adam(model.encoder.parameters(), lr=lr, betas=(0.5, 0.999))
    opt_d = torch.optim.Adam(model.decoder.parameters(), lr=lr, betas=(0.5, 0.999))
    opt_s = torch.optim.Adam(model.discriminator.parameters(), lr=0.5 * lr, betas=(0.5, 0.999))
    data = O.synth_batch(B, cfg, seed=1234, steps=steps)
    x = data["x"]

    def freeze(m, flag):
        for p in m.parameters():
            p.requires_grad = not flag

    out = {"meta/case": np.array("wae1"), "meta/B": B, "meta/seed": seed, "meta/steps": steps,
           "meta/image_size": cfg.image_size}
    for s in range(steps):
        model.train()
        model.encoder.zero_grad()
        model.decoder.zero_grad()
        model.discriminator.zero_grad()
        freeze(model.decoder, True)
        freeze(model.encoder, True)
        freeze(model.discriminator, False)
        z_real, var = model.encoder(x)
        z_fake = data["noise"][s, 2] * 0.5
        d_real = model.discriminator(z_real)
        d_fake = model.discriminator(z_fake)
        l_fake = -10 * torch.sum(torch.log(d_fake + 1e-3))
        l_real = -10 * torch.sum(torch.log(1 - d_real + 1e-3))
        l_fake.backward(retain_graph=True)
        l_real.backward(retain_graph=True)
        g = grads_of([("discriminator." + k, p) for k, p in model.discriminator.named_parameters()])
        opt_s.step()
        freeze(model.encoder, False)
        freeze(model.decoder, False)
        freeze(model.discriminator, True)
        z_real, var = model.encoder(x)
        x_recon = model.decoder(z_real)
        d_real = model.discriminator(z_real)
        l_rec = torch.sum(torch.sum(0.5 * (x_recon - x) ** 2, 1))
        l_pen = -10 * torch.sum(torch.log(d_real + 1e-3))
        l_rec.backward(retain_graph=True)
        l_pen.backward()
        g.update(grads_of([("encoder." + k, p) for k, p in model.encoder.named_parameters()]))
        g.update(grads_of([("decoder." + k, p) for k, p in model.decoder.named_parameters()]))
        opt_e.step()
        opt_d.step()
        logs = dict(loss_reconstruction=l_rec.item(), loss_penalty=l_pen.item(),
                    loss_discriminator_fake=l_fake.item(), loss_discriminator_real=l_real.item())
        record_step(out, f"step{s}", logs, dict(x_recon=x_recon, z_real=z_real), g)
        keys, summ = summarize_state(model.state_dict())
        out[f"step{s}/state_keys"] = np.array(keys)
        out[f"step{s}/state_sum"] = summ
        print(name, "step", s, {k: round(v, 5) for k, v in logs.items()})
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


def build_wae_cognitive(vg, cfg, V, seed):
    """Model wiring of train/train_wae_stage2.py:195-202 (Stage III wires the same objects, :208-221) on recipe
    weights: `teacher` = Stage-I WaeGan (image encoder + decoder), `model` = WaeGanCognitive sharing its decoder."""
    Z = cfg.latent_dim
    teacher = vg.WaeGan(device="cpu", z_size=Z)
    teacher.load_state_dict(O.fill_state(O.encoder_spec(cfg) + O.decoder_spec(cfg) + O.wae_discriminator_spec(cfg),
                                         seed, True))
    cog = vg.CognitiveEncoder(input_size=V, z_size=Z)
    cog.load_state_dict({k[len("encoder."):]: v for k, v in
                         O.fill_state(O.cognitive_encoder_spec(cfg, V), seed + 100, True).items()})
    model = vg.WaeGanCognitive(device="cpu", encoder=cog, decoder=teacher.decoder, z_size=Z)
    model.discriminator.load_state_dict({k[len("discriminator."):]: v for k, v in
                                         O.fill_state(O.wae_discriminator_spec(cfg), seed + 200, True).items()})
    return teacher, model


def case_wae23(name, cfg, B, V, seed, stage, steps=2):
    """train/train_wae_stage2.py:276-328 (stage 2) / train/train_wae_stage3.py:297-347 (stage 3)."""
    vg = load_reference(cfg)
    teacher, model = build_wae_cognitive(vg, cfg, V, seed)
    import torch.nn as nn
    opt_e = torch.optim.Adam(model.encoder.parameters(), lr=0.001, betas=(0.5, 0.999))
    opt_d = torch.optim.Adam(model.decoder.parameters(), lr=0.001, betas=(0.5, 0.999))
    opt_s = torch.optim.Adam(model.discriminator.parameters(), lr=0.0005, betas=(0.5, 0.999))
    data = O.synth_batch(B, cfg, n_voxels=V, seed=1234, steps=steps)
    x_image, x_fmri = data["x"], data["fmri"]

    def freeze(m, flag):
        for p in m.parameters():
            p.requires_grad = not flag

    if stage == 2:
        freeze(teacher.decoder, True)
    else:
        freeze(teacher.encoder, True)
        freeze(model.encoder, True)
    out = {"meta/case": np.array(f"wae{stage}"), "meta/B": B, "meta/V": V, "meta/seed": seed, "meta/steps": steps,
           "meta/image_size": cfg.image_size}
    for s in range(steps):
        model.train()
        g = {}
        if stage == 2:
            freeze(model.decoder, True)
            model.encoder.zero_grad()
            model.discriminator.zero_grad()
            z, _ = teacher.encoder(x_image)
            teacher.decoder(z)                       # x_gt: unused by the script, moves BN statistics
            freeze(model.encoder, True)
        else:
            freeze(model.encoder, True)
            model.decoder.zero_grad()
            model.discriminator.zero_grad()
            freeze(model.decoder, True)
        freeze(model.discriminator, False)
        z_fake, var = model.encoder(x_fmri)
        z_real, var = teacher.encoder(x_image)
        d_real = model.discriminator(z_real)
        d_fake = model.discriminator(z_fake)
        l_fake = -10 * torch.sum(torch.log(d_fake + 1e-3))
        l_real = -10 * torch.sum(torch.log(1 - d_real + 1e-3))
        l_fake.backward(retain_graph=True)
        l_real.backward(retain_graph=True)
        g.update(grads_of([("discriminator." + k, p) for k, p in model.discriminator.named_parameters()]))
        opt_s.step()
        if stage == 2:
            freeze(model.encoder, False)
        else:
            freeze(model.decoder, False)
        freeze(model.discriminator, True)
        z_real, var = model.encoder(x_fmri)
        x_recon = model.decoder(z_real)
        d_real = model.discriminator(z_real)
        l_rec = nn.MSELoss()(x_recon, x_image)
        l_pen = -10 * torch.mean(torch.log(d_real + 1e-3))
        l_rec.backward(retain_graph=True)
        if stage == 2:
            l_pen.backward()
            g.update(grads_of([("encoder." + k, p) for k, p in model.encoder.named_parameters()]))
            opt_e.step()
        else:
            g.update(grads_of([("decoder." + k, p) for k, p in model.decoder.named_parameters()]))
            opt_d.step()
        logs = dict(loss_reconstruction=l_rec.item(), loss_penalty=l_pen.item(),
                    loss_discriminator_fake=l_fake.item(), loss_discriminator_real=l_real.item())
        record_step(out, f"step{s}", logs, dict(x_recon=x_recon, z_real=z_real), g)
        sd = dict(model.state_dict())
        sd.update({"teacher_net.encoder." + k: v for k, v in teacher.encoder.state_dict().items()})
        keys, summ = summarize_state(sd)
        out[f"step{s}/state_keys"] = np.array(keys)
        out[f"step{s}/state_sum"] = summ
        print(name, "step", s, {k: round(v, 5) for k, v in logs.items()})
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


def case_dual1(name, cfg, B, seed, perturb, steps=3, lam=1.0, mode="vae-gan", beta=1.0):
    """train/wae_vgan_stage1.py:284-441, mode 'vae-gan': Stage-I VAE/GAN step + WAE latent-discriminator phase +
    latent penalty into the encoder.  Optimizer steps are applied after all gradients are taken;
    `zero_grad()` follows the pinned torch 1.4 (zeroes instead of dropping .grad), which makes the
    `optimizer_decoder.step()` of :417 an extra decay of the decoder's RMSprop state from the 2nd iteration on."""
    vg = load_reference(cfg)
    hp = O.GanHyper()
    Z = cfg.latent_dim
    model = vg.VaeGan(device="cpu", z_size=Z)
    model.load_state_dict(O.fill_state(O.vaegan_spec(cfg), seed, perturb))
    model_wae = vg.WaeGan(device="cpu", z_size=Z)
    model_wae.discriminator.load_state_dict(
        {k[len("wae_discriminator."):]: v for k, v in
         O.fill_state(O.wae_discriminator_spec(cfg, pre="wae_discriminator."), seed + 200, perturb).items()})
    model.train()
    data = O.synth_batch(B, cfg, seed=1234, steps=steps)
    x = data["x"]
    opt_e = rms(model.encoder.parameters(), hp.lr)
    opt_d = rms(model.decoder.parameters(), hp.lr)
    opt_s = rms(model.discriminator.parameters(), hp.lr)
    opt_w = rms(model_wae.discriminator.parameters(), hp.lr)

    def freeze(m, flag):
        for p in m.parameters():
            p.requires_grad = not flag

    out = {"meta/case": np.array("dual1"), "meta/B": B, "meta/seed": seed, "meta/perturb": perturb,
           "meta/steps": steps, "meta/image_size": cfg.image_size, "meta/lam": lam, "meta/mode": np.array(mode),
           "meta/beta": beta}
    for s in range(steps):
        nz = data["noise"][s]
        model.train()
        mus, lv = model.encoder(x)
        x_tilde = model.decoder(nz[0] * torch.exp(0.5 * lv) + mus)
        x_p = model.decoder(nz[1])
        disc_layer = model.discriminator(x, x_tilde, x_p, "REC")
        disc_class = model.discriminator(x, x_tilde, x_p, "GAN")
        nle, kld, mse, bo, bp, bs = vg.VaeGan.loss(x, x_tilde, disc_layer[:B], disc_layer[B:-B], disc_layer[-B:],
                                                   disc_class[:B], disc_class[B:-B], disc_class[-B:], mus, lv)
        # loss compositions and flags of wae_vgan_stage1.py:311-364
        train_enc, train_dis, train_dec = True, True, True
        if mode == "beta-vae":
            loss_enc = torch.sum(kld) * beta * (1 / B) + torch.sum(mse)
            loss_dis = torch.sum(bo) + torch.sum(bp) + torch.sum(bs)
            loss_dec = torch.sum(hp.lambda_mse * mse) - (1.0 - hp.lambda_mse) * loss_dis
        elif mode == "vae-gan":
            loss_enc = torch.sum(kld) + torch.sum(mse)
            loss_dis = torch.sum(bo) + torch.sum(bp) + torch.sum(bs)
            loss_dec = torch.sum(hp.lambda_mse * mse) - (1.0 - hp.lambda_mse) * loss_dis
        elif mode == "dcgan":
            train_enc = False
            loss_enc = torch.sum(kld) + torch.sum(nle)
            loss_dis = torch.sum(bo) + torch.sum(bs)
            loss_dec = torch.sum(hp.lambda_mse * nle) - (1.0 - hp.lambda_mse) * loss_dis
        elif mode == "vae":
            loss_enc = torch.sum(kld) + torch.sum(nle)
            loss_dis = torch.sum(bo) + torch.sum(bs)
            loss_dec = torch.sum(hp.lambda_mse * nle)
            train_dis = False
        bom, bpm = torch.mean(bo).item(), torch.mean(bp).item()
        if bom < hp.equilibrium - hp.margin or bpm < hp.equilibrium - hp.margin:
            train_dis = False
        if bom > hp.equilibrium + hp.margin or bpm > hp.equilibrium + hp.margin:
            train_dec = False
        if train_dec is False and train_dis is False:
            train_dis, train_dec = True, True
        model.zero_grad()
        # ---- WAE discriminator phase (:378-397)
        freeze(model.decoder, True)
        freeze(model.encoder, True)
        freeze(model_wae.discriminator, False)
        z_real, var = model.encoder(x)
        z_fake = nz[2] * 0.5
        d_real = model_wae.discriminator(z_real)
        d_fake = model_wae.discriminator(z_fake)
        l_fake = -lam * torch.sum(torch.log(d_fake + 1e-3))
        l_real = -lam * torch.sum(torch.log(1 - d_real + 1e-3))
        model_wae.discriminator.zero_grad()
        l_fake.backward(retain_graph=True)
        l_real.backward(retain_graph=True)
        g = grads_of([("wae_discriminator." + k, p) for k, p in model_wae.discriminator.named_parameters()])
        opt_w.step()
        # ---- generator phase (:401-417)
        freeze(model.encoder, False)
        freeze(model.decoder, False)
        freeze(model_wae.discriminator, True)
        z_real, var = model.encoder(x)
        x_recon = model.decoder(z_real)
        d_real = model_wae.discriminator(z_real)
        l_pen = -lam * torch.sum(torch.log(d_real + 1e-3))
        g_pen = torch.autograd.grad(l_pen, list(model.encoder.parameters()), allow_unused=True)
        # ---- VAE/GAN updates (:419-441), gradients at the pre-update weights
        g3 = literal_three_backward(model, loss_enc, loss_dec, loss_dis,
                                    (("encoder",) if train_enc else ()) + ("decoder", "discriminator"))
        if train_enc:
            for (k, p), gp in zip(model.encoder.named_parameters(), g_pen):
                if gp is not None:
                    g3["encoder." + k] = g3["encoder." + k] + gp  # :421 accumulates onto the penalty gradient
        g.update(g3)
        if s > 0:   # :417 under torch 1.4: decoder .grad are zero tensors there -> RMSprop decays square_avg, params
            #         unchanged (run here, after the backward passes, because the step bumps tensor versions)
            for p in model.decoder.parameters():
                p.grad = torch.zeros_like(p)
            opt_d.step()
        if train_enc:                                  # ('dcgan': the encoder is never stepped, :419)
            apply_grads(model.encoder, "encoder.", g, opt_e)
        if train_dec:
            apply_grads(model.decoder, "decoder.", g, opt_d)
        if train_dis:
            apply_grads(model.discriminator, "discriminator.", g, opt_s)
        model.zero_grad()
        logs = dict(loss_encoder=loss_enc.item(), loss_discriminator=loss_dis.item(), loss_decoder=loss_dec.item(),
                    nle=torch.sum(nle).item(), kl=torch.sum(kld).item(), mse=torch.sum(mse).item(),
                    bce_orig=torch.sum(bo).item(), bce_pred=torch.sum(bp).item(), bce_samp=torch.sum(bs).item(),
                    train_dis=float(train_dis), train_dec=float(train_dec), loss_penalty=l_pen.item(),
                    loss_discriminator_fake=l_fake.item(), loss_discriminator_real=l_real.item())
        fw = dict(x_tilde=x_tilde, x_p=x_p, disc_class=disc_class, disc_layer=disc_layer, mus=mus, log_variances=lv)
        record_step(out, f"step{s}", logs, fw, g)
        sd = dict(model.state_dict())
        sd.update({"wae_discriminator." + k: v for k, v in model_wae.discriminator.state_dict().items()})
        keys, summ = summarize_state(sd)
        out[f"step{s}/state_keys"] = np.array(keys)
        out[f"step{s}/state_sum"] = summ
        print(name, "step", s, {k: round(v, 5) for k, v in logs.items()})
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


def case_metrics(name):
    """PearsonCorrelation / StructuralSimilarity of the reference's train/train_utils.py on seeded image batches."""
    if REF not in sys.path:
        sys.path.insert(0, REF)
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tvu = types.ModuleType("torchvision.utils")
    tvu.make_grid = tvu.save_image = lambda *a, **k: None
    tv.models, tv.utils = tvm, tvu
    sys.modules.update({"torchvision": tv, "torchvision.models": tvm, "torchvision.utils": tvu})
    for m in ("train.train_utils", "train"):
        sys.modules.pop(m, None)
    tu = importlib.import_module("train.train_utils")
    pcc, ssim = tu.PearsonCorrelation(), tu.StructuralSimilarity()
    out = {"meta/case": np.array("metrics")}
    cases = [("b8_64", 8, 64, 64, 21), ("b3_100", 3, 100, 100, 22), ("b2_50x37", 2, 50, 37, 23)]
    out["meta/cases"] = np.array([c[0] for c in cases])
    for tag, n, h, w, seed in cases:
        rs = np.random.RandomState(seed)
        a = torch.from_numpy(rs.uniform(-1, 1, (n, 3, h, w)).astype(np.float32))
        b = (0.6 * a + 0.4 * torch.from_numpy(rs.uniform(-1, 1, (n, 3, h, w)).astype(np.float32)))
        s, c = ssim(a, b, full=True)
        out[f"{tag}/shape"] = np.array([n, 3, h, w, seed])
        out[f"{tag}/pcc"] = np.float64(pcc(a, b).item())
        out[f"{tag}/ssim"] = np.float64(s.item())
        out[f"{tag}/contrast"] = np.float64(c.item())
        out[f"{tag}/ssim_default"] = np.float64(ssim(a, b).item())
        print(name, tag, out[f"{tag}/pcc"], out[f"{tag}/ssim"], out[f"{tag}/contrast"])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


def eval_state(cfg, seed):
    """Recipe weights with non-trivial BN running statistics (as after some training)."""
    sd = O.fill_state(O.vaegan_spec(cfg), seed, True)
    rs = np.random.RandomState(seed + 1000)
    for k in sd:
        if k.endswith("running_mean"):
            sd[k] = torch.from_numpy(rs.uniform(-0.2, 0.2, tuple(sd[k].shape)).astype(np.float32))
        elif k.endswith("running_var"):
            sd[k] = torch.from_numpy(rs.uniform(0.5, 1.5, tuple(sd[k].shape)).astype(np.float32))
    return sd


def case_eval(name, cfg, B, seed):
    """Eval-mode forward of the reference VaeGan (models/vae_gan.py:288-297): BN with running statistics."""
    vg = load_reference(cfg)
    model = vg.VaeGan(device="cpu", z_size=cfg.latent_dim)
    model.load_state_dict(eval_state(cfg, seed))
    model.eval()
    data = O.synth_batch(B, cfg, seed=1234, steps=1)
    x, eps = data["x"], data["noise"][0, 0]
    with torch.no_grad():
        mus, lv = model.encoder(x)
        x_tilde = model.decoder(eps * torch.exp(0.5 * lv) + mus)
        x_p = model.decoder(data["noise"][0, 1])
        torch.manual_seed(3)
        ref = model(x)
        torch.manual_seed(3)
        e2 = torch.empty(B, cfg.latent_dim).normal_()
        assert torch.equal(ref, model.decoder(e2 * torch.exp(0.5 * lv) + mus))
    out = {"meta/case": np.array("eval"), "meta/B": B, "meta/seed": seed, "meta/image_size": cfg.image_size}
    pack(out, "fw", {k: O.tensor_summary(v) for k, v in dict(mus=mus, log_variances=lv, x_tilde=x_tilde, x_p=x_p).items()})
    print(name, float(x_tilde.norm()))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


def case_surface(name, cfg, B, V, seed):
    """Forward fingerprints of the wrapper classes no training script of the reference drives through ``forward``
    on the hot path: ``VaeGanCognitive(mode='wae')`` (models/vae_gan.py:379-387), ``WaeGanCognitive`` eval
    (:564-571), ``DCGan`` train / eval (:602-622).  The only random draw of these paths is ``z_p = torch.randn`` on
    the HOST (then ``.to(device)``), so the drop-in modules reproduce them under the same ``torch.manual_seed``."""
    vg = load_reference(cfg)
    data = O.synth_batch(B, cfg, n_voxels=V, seed=1234, steps=1)
    x, fmri = data["x"], data["fmri"]
    out = {"meta/case": np.array("surface"), "meta/B": B, "meta/V": V, "meta/seed": seed,
           "meta/image_size": cfg.image_size, "meta/torch_seed": 21}
    # --- VaeGanCognitive(mode='wae'), train mode, with the Stage-I teacher
    model = build_cognitive(vg, cfg, V, seed, True, 2)
    model.mode = "wae"
    torch.manual_seed(21)
    gt_x, x_tilde, disc_class, disc_layer, mus, lv = model({"fmri": fmri, "image": x})
    pack(out, "cogwae", {k: O.tensor_summary(v.detach()) for k, v in dict(
        gt_x=gt_x, x_tilde=x_tilde, disc_class=disc_class, disc_layer=disc_layer, mus=mus, log_variances=lv).items()})
    keys, summ = summarize_state(model.state_dict())       # running statistics after the forward's BN updates
    out["cogwae/state_keys"], out["cogwae/state_sum"] = np.array(keys), summ
    # --- WaeGanCognitive: keys + eval forward = decoder(encoder(fmri).mu)
    _, wmodel = build_wae_cognitive(vg, cfg, V, seed)
    out["waecog/state_keys"] = np.array(list(wmodel.state_dict().keys()))
    wmodel.eval()
    with torch.no_grad():
        xw = wmodel(fmri)
    out["waecog/x_tilde"] = O.tensor_summary(xw)
    assert all(not p.requires_grad for p in wmodel.decoder.parameters())        # the constructor freezes the decoder
    # --- DCGan: train forward (4-tuple) and eval forward
    teacher = vg.VaeGan(device="cpu", z_size=cfg.latent_dim)
    teacher.load_state_dict(O.fill_state(O.vaegan_spec(cfg), seed, True))
    dc = vg.DCGan(device="cpu", decoder=teacher.decoder, discriminator=teacher.discriminator, z_size=cfg.latent_dim)
    out["dcgan/state_keys"] = np.array(list(dc.state_dict().keys()))
    dc.train()
    torch.manual_seed(22)
    g, xt, dcl, dly = dc(x)
    pack(out, "dcgan", {k: O.tensor_summary(v.detach()) for k, v in dict(gt_x=g, x_tilde=xt, disc_class=dcl,
                                                                         disc_layer=dly).items()})
    dc.eval()
    torch.manual_seed(23)
    with torch.no_grad():
        out["dcgan/eval_x_p"] = O.tensor_summary(dc(x))
        out["dcgan/gen5"] = O.tensor_summary(dc(None, 5))
    print(name, float(x_tilde.norm()), float(xw.norm()), float(xt.norm()))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


def case_recon(name, cfg, B, seed):
    """``Discriminator(recon_level = 1 | 2)`` of the reference (models/vae_gan.py:139-173): the 'REC' call's output (raw
    convolution output of block ``recon_level``), its gradient w.r.t. the predicted images and the parameters under a
    seeded cotangent, the BatchNorm counters after the REC call and after a following 'GAN' call; plus the facts that
    level 0 raises and level 4 returns None."""
    vg = load_reference(cfg)
    rs = np.random.RandomState(1000 + seed)
    xs = [torch.from_numpy(rs.uniform(-1, 1, (B, 3, cfg.image_size, cfg.image_size)).astype(np.float32))
          .to(torch.get_default_dtype()) for _ in range(3)]
    out = {"meta/case": np.array("recon"), "meta/B": B, "meta/seed": seed, "meta/image_size": cfg.image_size}
    for level in (1, 2):
        dis = vg.Discriminator(channel_in=3, recon_level=level)
        dis.load_state_dict(O.fill_state(O.discriminator_spec(cfg, ""), seed, True))
        dis.train()
        xp = xs[1].clone().requires_grad_(True)
        feat = dis(xs[0], xp, xs[2], "REC")
        w = torch.from_numpy(np.random.RandomState(2000 + level).standard_normal(tuple(feat.shape)).astype(np.float32)) \
            .to(torch.get_default_dtype())
        (feat * w).sum().backward()
        tag = f"level{level}"
        out[f"{tag}/shape"] = np.array(feat.shape)
        out[f"{tag}/feat"] = O.tensor_summary(feat.detach())
        out[f"{tag}/dxp"] = O.tensor_summary(xp.grad)
        gk, gs = [], []
        for k, p_ in dis.named_parameters():
            gk.append(k)
            gs.append(O.tensor_summary(p_.grad) if p_.grad is not None else np.full(18, np.nan))
        out[f"{tag}/grad_keys"], out[f"{tag}/grad_sum"] = np.array(gk), np.stack(gs)
        sd = dis.state_dict()
        out[f"{tag}/nbt_rec"] = np.array([int(sd[f"conv.{i}.bn.num_batches_tracked"]) for i in (1, 2, 3)])
        with torch.no_grad():
            prob = dis(xs[0], xs[1], xs[2], "GAN")
        out[f"{tag}/prob"] = O.tensor_summary(prob)
        sd = dis.state_dict()
        out[f"{tag}/nbt_gan"] = np.array([int(sd[f"conv.{i}.bn.num_batches_tracked"]) for i in (1, 2, 3)])
    # the levels the reference cannot run
    d0 = vg.Discriminator(channel_in=3, recon_level=0)
    try:
        d0(xs[0], xs[1], xs[2], "REC")
        out["level0/raises"] = np.array("")
    except Exception as e:                                     # noqa: BLE001  (recording what the reference does)
        out["level0/raises"] = np.array(type(e).__name__)
    d4 = vg.Discriminator(channel_in=3, recon_level=4)
    with torch.no_grad():
        out["level4/is_none"] = np.array(d4(xs[0], xs[1], xs[2], "REC") is None)
    print(name, [int(v) for v in out["level1/shape"]], [int(v) for v in out["level2/shape"]], str(out["level0/raises"]),
          bool(out["level4/is_none"]))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


def case_resize(name):
    """CenterCrop + Resize of train_vgan_stage1.py:162-165 as torchvision 0.5.0 performs them, i.e. through PIL
    (torchvision itself is not installed here: its two functions are the three lines restated below)."""
    from PIL import Image
    from oracle.resize_oracle import RESIZE_CASES, resize_inputs          # seeded inputs only (no arithmetic of the oracle)
    out = {"meta/case": np.array("resize"), "meta/cases": np.array(RESIZE_CASES, np.int32), "meta/pillow": np.array(Image.__version__)}
    for idx, (img, (h, w, c, crop, size)) in enumerate(zip(resize_inputs(), RESIZE_CASES)):
        pil = Image.fromarray(img[:, :, 0] if c == 1 else img)
        iw, ih = pil.size
        i = int(round((ih - crop) / 2.))                      # torchvision 0.5.0 functional.center_crop
        j = int(round((iw - crop) / 2.))
        pil = pil.crop((j, i, j + crop, i + crop))
        pil = pil.resize((size, size), Image.BILINEAR)        # torchvision 0.5.0 functional.resize, (h, w) size
        res = np.asarray(pil)
        out[f"out/{idx}"] = res if res.ndim == 3 else res[:, :, None]
    print(name, [tuple(v.shape) for k, v in out.items() if k.startswith("out/")])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


def load_reference_data_loader():
    """Import the reference's ``data_preprocessing.data_loader`` (for its ``rand_shift`` and ``GreyToColor``).  Its
    top-level imports of packages that are not installed here and that those two never touch (nibabel, skimage) are
    satisfied with empty stub modules, like torchvision for models.vae_gan."""
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for name in ("nibabel", "skimage", "skimage.transform"):
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except ImportError:
                sys.modules[name] = types.ModuleType(name)
    if isinstance(sys.modules.get("skimage"), types.ModuleType) and not hasattr(sys.modules["skimage"], "transform"):
        sys.modules["skimage"].transform = sys.modules["skimage.transform"]
    for m in ("configs.data_config", "configs"):
        sys.modules.pop(m, None)
    return importlib.import_module("data_preprocessing.data_loader")


def case_ingest(name):
    """Tail of the image pipeline on uint8 images (train_vgan_stage1.py:162-170, data_loader.py:186-217):
    RandomHorizontalFlip -> RandomShift -> ToTensor -> GreyToColor -> Normalize.  The shift and the grey -> colour
    step are the REFERENCE'S OWN ``rand_shift`` (seeded numpy draw, scipy.ndimage.shift nearest / order 0) and
    ``GreyToColor`` (data_preprocessing/data_loader.py:203-217, :374-400), imported from /root/reference; ToTensor /
    Normalize / the flip are torchvision-0.5 semantics restated with torch ops (torchvision is not installed here)."""
    dl = load_reference_data_loader()
    rs = np.random.RandomState(31)
    out = {"meta/case": np.array("ingest")}
    for tag, n, h, w, c in (("rgb", 3, 20, 24, 3), ("grey", 2, 16, 16, 1)):
        img = rs.randint(0, 256, (n, h, w, c)).astype(np.uint8)
        flip = rs.randint(0, 2, n).astype(np.int32)
        shifts = np.zeros((n, 2), np.int32)
        mean, std = (0.5, 0.5, 0.5), (0.5, 0.5, 0.5)
        res = []
        for i in range(n):
            a = img[i]
            if flip[i]:
                a = np.ascontiguousarray(a[:, ::-1, :])
            seed = 1000 + 17 * i + (0 if tag == "rgb" else 500)
            np.random.seed(seed)
            shifts[i] = np.random.randint(-5, 6, size=2)              # the draw rand_shift makes (data_loader.py:215)
            np.random.seed(seed)
            a = dl.rand_shift(a, 5)                                    # REFERENCE code
            t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).float().div(255)      # ToTensor
            if t.shape[0] != 3:
                t = dl.GreyToColor(h)(t[0]).clone()                    # REFERENCE code (square grey images)
            t = (t - torch.tensor(mean).view(3, 1, 1)) / torch.tensor(std).view(3, 1, 1)           # Normalize
            res.append(t.numpy())
        out[f"{tag}/img"] = img
        out[f"{tag}/flip"] = flip
        out[f"{tag}/shift"] = shifts
        out[f"{tag}/out"] = np.stack(res).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "written")


if __name__ == "__main__":
    torch.set_num_threads(8)
    which = [a for a in sys.argv[1:] if not a.startswith("--")] or ["all"]
    # --f64: the same cases with the reference modules, the recipe weights and the inputs in DOUBLE precision, written as
    # <case>_f64.npz.  Two fp32 runs of the same arithmetic differ by ~1e-3 in some gradient norms with the host's thread
    # count alone (reduction order of the CPU convolutions); two float64 runs agree to ~1e-12, so these fixtures pin the
    # oracle's RESTATEMENT of the reference exactly, on any host (tests/test_oracle_golden.py).
    F64 = "--f64" in sys.argv[1:]
    if F64:
        torch.set_default_dtype(torch.float64)
        _fill, _synth = O.fill_state, O.synth_batch
        dbl = lambda d: {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in d.items()}
        O.fill_state = lambda *a, **k: dbl(_fill(*a, **k))
        O.synth_batch = lambda *a, **k: dbl(_synth(*a, **k))
        _savez = np.savez_compressed

        def savez_f64(path, **out):
            assert path.endswith(".npz")
            return _savez(path[:-4] + "_f64.npz", **out)
        np.savez_compressed = savez_f64

    def want(n):
        return "all" in which or n in which

    if want("stage1_b4"):
        case_stage1("stage1_b4", O.ArchCfg.px64(), B=4, seed=0, perturb=True)
    if want("stage1_b32"):
        case_stage1("stage1_b32", O.ArchCfg.px64(), B=32, seed=0, perturb=False, steps=1)
    if want("stage1_px100_b2"):
        case_stage1("stage1_px100_b2", O.ArchCfg.px100(), B=2, seed=3, perturb=True, steps=1)
    if want("stage1_betavae_b4"):
        case_stage1("stage1_betavae_b4", O.ArchCfg.px64(), B=4, seed=0, perturb=True, mode="beta-vae", beta=4.0)
    if want("stage1_dcgan_b4"):
        case_stage1("stage1_dcgan_b4", O.ArchCfg.px64(), B=4, seed=0, perturb=True, mode="dcgan")
    if want("stage1_vae_b4"):
        case_stage1("stage1_vae_b4", O.ArchCfg.px64(), B=4, seed=0, perturb=True, mode="vae")
    if want("stage3_px128_b2"):
        case_cognitive("stage3_px128_b2", O.ArchCfg.px128(), B=2, V=3620, seed=5, perturb=True, stage=3, steps=1)
    if want("wae3_px128_b2"):
        case_wae23("wae3_px128_b2", O.ArchCfg.px128(), B=2, V=3620, seed=7, stage=3, steps=1)
    if want("stage2_vae_b4"):
        case_cognitive("stage2_vae_b4", O.ArchCfg.px64(), B=4, V=4096, seed=1, perturb=True, stage=2, mode="vae")
    if want("stage3_vae_b4"):
        case_cognitive("stage3_vae_b4", O.ArchCfg.px64(), B=4, V=4096, seed=2, perturb=True, stage=3, mode="vae")
    if want("stage2_b4"):
        case_cognitive("stage2_b4", O.ArchCfg.px64(), B=4, V=4096, seed=1, perturb=True, stage=2)
    if want("stage3_b4"):
        case_cognitive("stage3_b4", O.ArchCfg.px64(), B=4, V=4096, seed=2, perturb=True, stage=3)
    if want("wae1_b4"):
        case_wae1("wae1_b4", O.ArchCfg.px64(), B=4, seed=5)
    if want("wae2_b4"):
        case_wae23("wae2_b4", O.ArchCfg.px64(), B=4, V=4096, seed=6, stage=2)
    if want("wae3_b4"):
        case_wae23("wae3_b4", O.ArchCfg.px64(), B=4, V=4096, seed=7, stage=3)
    if want("dual1_b4"):
        case_dual1("dual1_b4", O.ArchCfg.px64(), B=4, seed=8, perturb=True)
    for m in ("beta-vae", "dcgan", "vae"):
        n = "dual1_" + m.replace("-", "") + "_b4"
        if want(n):
            case_dual1(n, O.ArchCfg.px64(), B=4, seed=8, perturb=True, steps=2, mode=m, beta=4.0)
    if want("metrics"):
        case_metrics("metrics")
    if want("eval_b4"):
        case_eval("eval_b4", O.ArchCfg.px64(), B=4, seed=9)
    if want("ingest"):
        case_ingest("ingest")
    if want("resize"):
        case_resize("resize")
    if want("surface_b4"):
        case_surface("surface_b4", O.ArchCfg.px64(), B=4, V=4096, seed=12)
    if want("recon_b4"):
        case_recon("recon_b4", O.ArchCfg.px64(), B=4, seed=21)
