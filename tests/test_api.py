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
