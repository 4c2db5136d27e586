"""Fused WAE training steps: our restatement of the inline loop bodies of the reference's WAE scripts.

    WaeStep(stage=1)   train/train_wae_stage1.py:259-311   image -> image, encoder + decoder + latent-D (Adam)
    WaeStep(stage=2)   train/train_wae_stage2.py:276-328   fMRI -> image, cognitive encoder + latent-D trained
    WaeStep(stage=3)   train/train_wae_stage3.py:297-347   fMRI -> image, decoder + latent-D trained
    DualStage1Step     train/wae_vgan_stage1.py:284-441    Stage-I VAE/GAN step + latent-D phase + latent penalty

Every script runs two phases per batch with an optimizer step in between:
  D phase  latent discriminator on "real" vs "fake" latents (both detached), two log-losses, one update;
  G phase  the generator side is run AGAIN (same weights -> same activations, so the engine runs it once and
           lets the train-mode BatchNorm layers take the matching number of running-stat updates), the UPDATED
           discriminator scores the latents, and reconstruction + penalty are back-propagated.
Nothing synchronises with the host inside a step; fp16 cotangents are kept in range with static scales and the
device-side unit-RMS re-normalisation of the encoder cotangent (same scheme as steps.py).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import lib
from .nets import CognitiveEncoderNet, DecoderNet, EncoderNet, WaeDiscriminatorNet
from .ops import images_to_nhwc, nhwc_to_images, pad8, require_gpu, rows_to_f16
from .params import ArchConfig
from .steps import (S_ESQ, S_NA, S_NB, S_NE, GanHyper, Scales, Stage1Step, _attach_reducers, _Dist, _GanStepBase,
                    _Optim)

_P = lib.ptr

# slots of the scalar block used by the WAE steps (sums over the global batch; [0, 4) are all-reduced)
W_REC, W_PEN, W_DFAKE, W_DREAL = 0, 1, 2, 3
W_LOG_KEYS = ("loss_reconstruction", "loss_penalty", "loss_discriminator_fake", "loss_discriminator_real")


@dataclass
class WaeHyper:
    """Stage I: train/train_wae_stage1.py:221-224 (lr from configs/wae_config.py); Stage II/III: hard-coded
    train/train_wae_stage2.py:237-239."""
    lr_enc: float = 1e-4
    lr_dec: float = 1e-4
    lr_dis: float = 0.5e-4
    lam: float = 10.0
    betas: tuple = (0.5, 0.999)

    @staticmethod
    def stage23() -> "WaeHyper":
        return WaeHyper(lr_enc=1e-3, lr_dec=1e-3, lr_dis=5e-4)


class _LatentDiscPhase:
    """The latent-discriminator pieces shared by WaeStep and DualStage1Step."""

    def _dis_phase(self, wd: WaeDiscriminatorNet, opt: _Optim, z_real16, z_fake16, lam: float, scal, dd: _Dist):
        """D phase on detached latents: l_fake = -lam*sum log(d_fake+1e-3), l_real = -lam*sum log(1-d_real+1e-3)
        (e.g. train_wae_stage1.py:278-288); one forward over [real ; fake], gradients, optimizer step."""
        B = z_real16.shape[0]
        dev = z_real16.device
        zz = torch.cat([z_real16, z_fake16], 0)
        logit32, ctx = wd.forward(zz)
        dlogit = torch.empty(2 * B, 8, dtype=torch.float16, device=dev)
        lib.call("fmri_wae_logloss", _P(logit32[:B]), 1, B, 1, lam, _P(scal[W_DREAL:W_DREAL + 1]), None,
                 _P(dlogit[:B]), 8, 1.0)
        lib.call("fmri_wae_logloss", _P(logit32[B:]), 1, B, 0, lam, _P(scal[W_DFAKE:W_DFAKE + 1]), None,
                 _P(dlogit[B:]), 8, 1.0)
        wd.group.zero_grad()
        wd.backward(ctx, dlogit, 1.0, True, False)
        dd.all_reduce(wd.group.grad)
        opt.step()

    def _penalty(self, wd: WaeDiscriminatorNet, z16, w: float, gscale: float, scal, need_dz: bool):
        """G-phase penalty -w*sum log(d(z)+1e-3) with the updated discriminator; returns d penalty / d z (fp32,
        true scale) or None."""
        B = z16.shape[0]
        logit32, ctx = wd.forward(z16)
        dlogit = torch.empty(B, 8, dtype=torch.float16, device=z16.device) if need_dz else None
        lib.call("fmri_wae_logloss", _P(logit32), 1, B, 0, w, _P(scal[W_PEN:W_PEN + 1]), None, _P(dlogit), 8, gscale)
        if not need_dz:
            return None
        return wd.backward(ctx, dlogit, gscale, False, True)


class WaeStep(_LatentDiscPhase):
    """WAE/GAN Stage I / II / III step."""

    def __init__(self, cfg: ArchConfig, device, stage: int = 1, n_voxels: int = 0, hp: Optional[WaeHyper] = None,
                 scales: Optional[Scales] = None, distributed: bool = False, sync_bn: bool = True):
        assert stage in (1, 2, 3)
        self.cfg, self.stage, self.n_voxels = cfg, stage, n_voxels
        self.sc = Scales() if scales is None else scales
        self.hp = hp if hp is not None else (WaeHyper() if stage == 1 else WaeHyper.stage23())
        self.device = torch.device(device)
        self.img_enc = EncoderNet(cfg, device)                       # Stage I: trained; II/III: Stage-I teacher
        self.cog = CognitiveEncoderNet(cfg, n_voxels, device) if stage > 1 else None
        self.dec = DecoderNet(cfg, device, self.img_enc.size)
        self.dec.fc_bn.enable_lazy_running()
        self._pre_replay = [self.dec.fc_bn._running_in]
        self.wd = WaeDiscriminatorNet(cfg, device)
        self.scal = torch.zeros(32, dtype=torch.float32, device=device)
        self.dd = _Dist(distributed, sync_bn)
        _attach_reducers([n for n in (self.img_enc, self.cog, self.dec) if n is not None], self.dd)
        hp_ = self.hp
        self.enc = self.img_enc if stage == 1 else self.cog          # the network `model.encoder` refers to
        self.opt_enc = _Optim(self.enc.group, "adam", hp_.lr_enc, betas=hp_.betas)
        self.opt_dec = _Optim(self.dec.group, "adam", hp_.lr_dec, betas=hp_.betas)
        self.opt_dis = _Optim(self.wd.group, "adam", hp_.lr_dis, betas=hp_.betas)
        self.fw = {}

    # ---- parameters (the golden-fixture recipes of tests/golden/make_golden.py) --------------------------
    def load_recipe(self, seed: int, perturb: Optional[bool] = None):
        if self.stage == 1:
            rs = np.random.RandomState(seed)
            for n in (self.img_enc, self.dec, self.wd):
                n.group.load_recipe(rs, bool(perturb))
        else:
            rs = np.random.RandomState(seed)
            for n in (self.img_enc, self.dec):
                n.group.load_recipe(rs, True)
            self.cog.group.load_recipe(np.random.RandomState(seed + 100), True)
            self.wd.group.load_recipe(np.random.RandomState(seed + 200), True)

    def state_dict(self):
        sd = {}
        sd.update(self.enc.group.state_dict("encoder."))
        sd.update(self.dec.group.state_dict("decoder."))
        sd.update(self.wd.group.state_dict("discriminator."))
        if self.stage > 1:
            sd.update(self.img_enc.group.state_dict("teacher_net.encoder."))
        return sd

    # ---- the step --------------------------------------------------------------------------------------------
    def step(self, image: torch.Tensor, z_fake_noise: Optional[torch.Tensor] = None,
             fmri: Optional[torch.Tensor] = None):
        """Stage I: step(x, z_fake_noise) with z_fake = 0.5 * noise (train_wae_stage1.py:276).
        Stage II/III: step(image, fmri=fmri)."""
        require_gpu(image)
        cfg, hp, sc, st = self.cfg, self.hp, self.sc, self.stage
        B, _, H, W = image.shape
        Z, zp = cfg.latent_dim, pad8(cfg.latent_dim)
        dev = image.device
        Bg = B * self.dd.world
        self.scal.zero_()
        x16 = images_to_nhwc(image)

        def latent16(head32, rows=B):
            z16 = torch.empty(rows, zp, dtype=torch.float16, device=dev)
            lib.call("fmri_latent_fwd", _P(head32), None, rows, Z, zp, _P(z16), None, None, 0)     # z = mu
            return z16

        # ---- generator-side forwards (run once; BN running stats take the script's number of updates) ------
        if st == 1:
            head32, ectx = self.img_enc.forward(x16, updates=2)                       # :275 and :296
            z16 = latent16(head32)
            z_real16, z_fake16 = z16, rows_to_f16(z_fake_noise, 0.5)
            y, dctx = self.dec.forward(z16, 1)                                        # :297
        else:
            head_t, _ = self.img_enc.forward(x16, updates=2 if st == 2 else 1)        # stage 2: :284,:293; 3: :312
            z_t16 = latent16(head_t)
            head32, ectx = self.cog.forward(rows_to_f16(fmri), updates=2)             # :292,:314 / :311,:333
            z16 = latent16(head32)
            z_real16, z_fake16 = z_t16, z16
            if st == 2:
                # decoder call order: x_gt = dec(z_teacher) (:285, unused, moves BN statistics), then x_recon
                yy, dctx = self.dec.forward(torch.cat([z_t16, z16], 0), 2, stat_order=(0, 1))
                y = yy[B:]
                g_rec = 1
            else:
                y, dctx = self.dec.forward(z16, 1)
        if st != 2:
            g_rec = 0

        # ---- D phase ---------------------------------------------------------------------------------------------
        self._dis_phase(self.wd, self.opt_dis, z_real16, z_fake16, hp.lam, self.scal, self.dd)

        # ---- G phase ---------------------------------------------------------------------------------------------
        npix = B * H * W
        if st == 1:
            rec_w, rec_scale = 1.0, 1.0                                  # sum 0.5 (x~ - x)^2, d/dx~ = x~ - x
            pen_w, pen_scale = hp.lam, 1.0                               # -lam * sum log
        else:
            n_el = float(Bg * 3 * H * W)
            rec_w, rec_scale = 2.0 / n_el, n_el / 2.0                   # MSELoss(mean): (2/N) * 0.5 sum (.)^2
            pen_w, pen_scale = hp.lam / Bg, float(Bg)                    # -lam * mean log
        dxt = torch.empty(B, H, W, 8, dtype=torch.float16, device=dev)
        lib.call("fmri_pixel_sq", _P(x16), _P(y), npix, 3, 8, _P(self.scal[W_REC:W_REC + 1]), _P(dxt), 1.0)
        self.scal[W_REC:W_REC + 1].mul_(rec_w)
        train_enc = st != 3
        dz_pen = self._penalty(self.wd, z16, pen_w, pen_scale, self.scal, need_dz=train_enc)
        self.dd.all_reduce(self.scal[:4])

        train_dec = st != 2
        if train_dec:
            self.dec.group.zero_grad()
        entries = [dict(g=g_rec, scale=rec_scale, train=train_dec, need_dz=train_enc)]
        dz_rec = self.dec.backward(dctx, dxt, entries)
        if train_dec:
            self.dd.all_reduce_async(self.dec.group.grad)           # runs under the encoder's backward pass
        if train_enc:
            dz = dz_rec[0] + dz_pen[:, :Z]
            dhead32 = torch.zeros(B, 2 * Z, dtype=torch.float32, device=dev)        # l_var gets no gradient
            dhead32[:, :Z] = dz
            dhead16 = self._renorm(dhead32, sc.enc, Bg)
            self.enc.group.zero_grad()
            self.enc.backward(ectx, dhead16, sc.enc)
            self.dd.all_reduce_async(self.enc.group.grad)
        self.dd.wait_all()
        if train_enc:
            self.opt_enc.step(gdev=self.scal[S_NE:S_NE + 1])
        if train_dec:
            self.opt_dec.step()
        self.fw = dict(B=B, y=y, head32=head32, Z=Z)
        return self.scal

    # HIP-graph recording of the whole step (Adam's step count and learning rates live on the device): the WAE steps are
    # ~300 launches of a few microseconds each -- eagerly issued they are bound by the host, replayed by the GPU
    capture = _GanStepBase.capture
    _capture = _GanStepBase._capture
    _versioned = _GanStepBase._versioned

    def _renorm(self, x32: torch.Tensor, scale: float, rows_global: int):
        n = x32.numel()
        ne = self.scal[S_NE:S_NE + 1]
        esq = self.__dict__.get("esq64")
        if esq is None:
            esq = self.esq64 = torch.zeros(1, dtype=torch.float64, device=x32.device)
        lib.call("fmri_sumsq_f64", _P(x32), n, _P(esq), 1)
        self.dd.all_reduce(esq)
        out = torch.empty(x32.shape, dtype=torch.float16, device=x32.device)
        lib.call("fmri_renorm_f64", _P(x32), _P(out), n, float(scale), _P(esq), float(rows_global) * (n // x32.shape[0]),
                 None, _P(ne))
        return out

    # ---- views for tests / API -----------------------------------------------------------------------------
    def logs(self):
        v = self.scal.tolist()
        return {k: v[i] for i, k in enumerate(W_LOG_KEYS)}

    def outputs(self):
        fw = self.fw
        return dict(x_recon=nhwc_to_images(fw["y"], 3), z_real=fw["head32"][:, :fw["Z"]].clone())

    def named_grads(self):
        """True-scale gradients of the last step (syncs; tests only)."""
        ne = self.scal[S_NE].item()
        out = {}
        for k, v in self.wd.group.grads.items():
            out["discriminator." + k] = v.clone()
        if self.stage != 3:
            for k, v in self.enc.group.grads.items():
                out["encoder." + k] = v / ne
        if self.stage != 2:
            for k, v in self.dec.group.grads.items():
                out["decoder." + k] = v.clone()
        return out


class DualStage1Step(Stage1Step, _LatentDiscPhase):
    """Dual WAE + VAE/GAN Stage-I step (train/wae_vgan_stage1.py:284-441; ``mode``: its four loss compositions,
    :311-364, as in Stage1Step -- 'vae-gan' default, 'beta-vae', 'dcgan' (the encoder is never stepped, :419), 'vae'):
    the Stage-I VAE/GAN step plus a WAE latent discriminator (RMSprop) trained on the encoder means, whose penalty
    gradient is added to the encoder's VAE/GAN gradient.  The encoder runs three times per batch in the script (one pass here, three
    running-stat updates) and the decoder three times (z, z_p, mu -- the last only moves BN statistics).

    ``torch14_zero_grad=True`` reproduces the pinned torch 1.4: the script's `optimizer_decoder.step()` at :417
    runs on zeroed gradients from the second iteration on, which only decays the decoder's RMSprop state."""

    def __init__(self, cfg: ArchConfig, device, hp: Optional[GanHyper] = None, scales: Optional[Scales] = None,
                 lam: float = 1.0, distributed: bool = False, sync_bn: bool = True, torch14_zero_grad: bool = True,
                 mode: str = "vae-gan"):
        super().__init__(cfg, device, hp, scales, distributed, sync_bn, mode=mode)
        hp = self.hp
        self.lam = lam
        self.torch14 = torch14_zero_grad
        self.wd = WaeDiscriminatorNet(cfg, device)
        self.opt_wd = _Optim(self.wd.group, "rmsprop", hp.lr, hp.alpha, hp.eps)
        self.wscal = torch.zeros(8, dtype=torch.float32, device=device)
        self.enc_updates = 3
        self.extra_mu_decoder_pass = True
        self._it = 0

    def load_recipe(self, seed: int, perturb: bool = False):
        super().load_recipe(seed, perturb)
        self.wd.group.load_recipe(np.random.RandomState(seed + 200), perturb)

    def state_dict(self):
        sd = super().state_dict()
        sd.update(self.wd.group.state_dict("wae_discriminator."))
        return sd

    def step(self, x, eps, z_p, z_fake_noise):
        fw = self.forward(x, eps, z_p)
        self.gate(fw["B"] * self.dd.world)
        B, Z = fw["B"], self.cfg.latent_dim
        zp = pad8(Z)
        self.wscal.zero_()
        mu16 = torch.empty(B, zp, dtype=torch.float16, device=x.device)
        lib.call("fmri_latent_fwd", _P(fw["head32"]), None, B, Z, zp, _P(mu16), None, None, 0)
        self._dis_phase(self.wd, self.opt_wd, mu16, rows_to_f16(z_fake_noise, 0.5), self.lam, self.wscal, self.dd)
        dz_pen = self._penalty(self.wd, mu16, self.lam, 1.0, self.wscal, need_dz=True)
        self.dd.all_reduce(self.wscal[:4])
        if self.torch14 and self._it > 0:
            self.opt_dec.s1.mul_(self.hp.alpha)                       # :417 on zeroed grads: state decay only
        self._it += 1
        self.backward(extra_dmu=dz_pen)
        self.apply()
        return self.scal

    def logs(self):
        out = super().logs()
        v = self.wscal.tolist()
        out.update(loss_penalty=v[W_PEN], loss_discriminator_fake=v[W_DFAKE], loss_discriminator_real=v[W_DREAL])
        return out

    def named_grads(self):
        out = super().named_grads()
        for k, v in self.wd.group.grads.items():
            out["wae_discriminator." + k] = v.clone()
        return out
