"""TEST INFRASTRUCTURE ONLY -- functional PyTorch-CPU fp32 restatement of the reference hot path.

This is the parity oracle for the HIP engine.  It restates, in a functional style (a flat
``{state_dict key: tensor}`` parameter dict instead of nn.Modules), the arithmetic of

  * the sub-networks of ``models/vae_gan.py``   (Encoder :63-96, Decoder :99-132,
    Discriminator :135-187, CognitiveEncoder :190-232, WaeDiscriminator :499-529),
  * ``VaeGan.forward`` / ``VaeGan.loss``         (models/vae_gan.py:271-287, :302-320),
  * ``VaeGanCognitive.forward``                  (models/vae_gan.py:352-395),
  * the inline training-step bodies of ``train/train_vgan_stage1.py:330-432``,
    ``train/train_vgan_stage2.py:331-407``, ``train/train_vgan_stage3.py:336-411``,
    ``train/train_wae_stage1.py:259-311`` and ``train/wae_vgan_stage1.py:277-454``.

Pinning: ``tests/test_oracle_golden.py`` checks every function here against the golden vectors in
``tests/golden/*.npz`` that ``tests/golden/make_golden.py`` produced by importing and running the
real reference (``/root/reference``) in the build container.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
State = Dict[str, Tensor]


# ----------------------------------------------------------------------------------------------
# architecture description  (configs/models_config.py:3-31)
# ----------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class ArchCfg:
    image_size: int = 64
    fc_input: int = 8
    fc_output: int = 1024
    fc_input_gan: int = 8
    fc_output_gan: int = 512
    stride_gan: int = 1
    latent_dim: int = 128
    output_pad_dec: Tuple[bool, bool, bool] = (True, True, True)
    encoder_channels: Tuple[int, ...] = (64, 128, 256)
    decoder_channels: Tuple[int, ...] = (256, 128, 32, 3)
    discrim_channels: Tuple[int, ...] = (32, 128, 256, 256, 512)
    kernel_size: int = 5
    stride: int = 2
    padding: int = 2

    @staticmethod
    def px64() -> "ArchCfg":
        """64x64 settings: the commented block configs/models_config.py:23-31 (+ :9)."""
        return ArchCfg()

    @staticmethod
    def px100() -> "ArchCfg":
        """as-shipped 'paper settings', configs/models_config.py:12-21."""
        return ArchCfg(image_size=100, fc_input=13, fc_output=1024, fc_input_gan=7, fc_output_gan=256,
                       stride_gan=2, latent_dim=512, output_pad_dec=(False, True, True),
                       decoder_channels=(256, 128, 64, 3))

    @staticmethod
    def px128() -> "ArchCfg":
        """128x128 variant used by BASELINE config 5: fc_input=16, stride_gan=2."""
        return ArchCfg(image_size=128, fc_input=16, fc_input_gan=8, fc_output_gan=512, stride_gan=2)


# ----------------------------------------------------------------------------------------------
# state-dict specifications (key order == reference ``state_dict()`` order)
# ----------------------------------------------------------------------------------------------
def _bn_spec(pre: str, c: int):
    return [(pre + "weight", (c,), "gamma"), (pre + "bias", (c,), "beta"),
            (pre + "running_mean", (c,), "rm"), (pre + "running_var", (c,), "rv"),
            (pre + "num_batches_tracked", (), "nbt")]


def encoder_spec(cfg: ArchCfg, pre: str = "encoder.", channel_in: int = 3):
    k = cfg.kernel_size
    out, cin = [], channel_in
    for i, c in enumerate(cfg.encoder_channels[:3]):
        out.append((f"{pre}conv.{i}.conv.weight", (c, cin, k, k), "w"))
        out += _bn_spec(f"{pre}conv.{i}.bn.", c)
        cin = c
    out.append((f"{pre}fc.0.weight", (cfg.fc_output, cfg.fc_input * cfg.fc_input * cin), "w"))
    out += _bn_spec(f"{pre}fc.1.", cfg.fc_output)
    for h in ("l_mu", "l_var"):
        out.append((f"{pre}{h}.weight", (cfg.latent_dim, cfg.fc_output), "w"))
        out.append((f"{pre}{h}.bias", (cfg.latent_dim,), "b"))
    return out


def decoder_spec(cfg: ArchCfg, pre: str = "decoder.", size: int = None):
    k = cfg.kernel_size
    size = cfg.encoder_channels[2] if size is None else size
    feat = cfg.fc_input * cfg.fc_input * size
    out = [(f"{pre}fc.0.weight", (feat, cfg.latent_dim), "w")]
    out += _bn_spec(f"{pre}fc.1.", feat)
    chans = [(size, size), (size, cfg.decoder_channels[1]), (cfg.decoder_channels[1], cfg.decoder_channels[2])]
    for i, (ci, co) in enumerate(chans):
        out.append((f"{pre}conv.{i}.conv.weight", (ci, co, k, k), "w"))  # ConvTranspose2d: (Cin, Cout, k, k)
        out += _bn_spec(f"{pre}conv.{i}.bn.", co)
    out.append((f"{pre}conv.3.0.weight", (cfg.decoder_channels[3], cfg.decoder_channels[2], 5, 5), "w"))
    out.append((f"{pre}conv.3.0.bias", (cfg.decoder_channels[3],), "b"))
    return out


def discriminator_spec(cfg: ArchCfg, pre: str = "discriminator."):
    k = cfg.kernel_size
    d = cfg.discrim_channels
    out = [(f"{pre}conv.0.0.weight", (d[0], 3, 5, 5), "w"), (f"{pre}conv.0.0.bias", (d[0],), "b")]
    cin = d[0]
    for i in (1, 2, 3):
        out.append((f"{pre}conv.{i}.conv.weight", (d[i], cin, k, k), "w"))
        out += _bn_spec(f"{pre}conv.{i}.bn.", d[i])
        cin = d[i]
    out.append((f"{pre}fc.0.weight", (cfg.fc_output_gan, cfg.fc_input_gan * cfg.fc_input_gan * cin), "w"))
    out += _bn_spec(f"{pre}fc.1.", cfg.fc_output_gan)
    out.append((f"{pre}fc.3.weight", (1, cfg.fc_output_gan), "w"))
    out.append((f"{pre}fc.3.bias", (1,), "b"))
    return out


def cognitive_encoder_spec(cfg: ArchCfg, n_voxels: int, pre: str = "encoder."):
    out = [(f"{pre}fc1.0.weight", (1024, n_voxels), "w")]
    out += _bn_spec(f"{pre}fc1.1.", 1024)
    for h in ("l_mu", "l_var"):
        out.append((f"{pre}{h}.weight", (cfg.latent_dim, 1024), "w"))
        out.append((f"{pre}{h}.bias", (cfg.latent_dim,), "b"))
    return out


def wae_discriminator_spec(cfg: ArchCfg, pre: str = "discriminator.", dim_h: int = 512):
    dims = [cfg.latent_dim, dim_h, dim_h, dim_h, dim_h, 1]
    out = []
    for j, idx in enumerate((0, 2, 4, 6, 8)):
        out.append((f"{pre}main.{idx}.weight", (dims[j + 1], dims[j]), "wn"))
        out.append((f"{pre}main.{idx}.bias", (dims[j + 1],), "b"))
    return out


def vaegan_spec(cfg: ArchCfg, pre: str = ""):
    return encoder_spec(cfg, pre + "encoder.") + decoder_spec(cfg, pre + "decoder.") + \
        discriminator_spec(cfg, pre + "discriminator.")


def fill_state(spec, seed: int, perturb: bool = False) -> State:
    """Deterministic parameter recipe shared by the golden generator, the oracle and the engine tests.

    Mirrors ``VaeGan.init_parameters`` (models/vae_gan.py:252-264): conv/deconv/linear weights
    ~ U(+-1/sqrt(prod(shape[1:]))/sqrt(3)), biases 0, BN gamma 1 / beta 0 -- but drawn from a numpy
    MT19937 stream (stable across torch versions).  ``perturb=True`` additionally jitters biases and
    BN affine parameters so that those code paths cannot hide behind 0/1 values.
    """
    rs = np.random.RandomState(seed)
    out: State = {}
    for key, shape, kind in spec:
        if kind == "w":
            s = 1.0 / math.sqrt(float(np.prod(shape[1:]))) / math.sqrt(3.0)
            v = rs.uniform(-s, s, shape).astype(np.float32)
        elif kind == "wn":  # WaeDiscriminator init, models/vae_gan.py:522-525
            v = rs.normal(0.0, 0.0099999, shape).astype(np.float32)
        elif kind == "b":
            v = rs.uniform(-0.05, 0.05, shape).astype(np.float32) if perturb else np.zeros(shape, np.float32)
        elif kind == "gamma":
            v = (1.0 + rs.uniform(-0.2, 0.2, shape)).astype(np.float32) if perturb else np.ones(shape, np.float32)
        elif kind == "beta":
            v = rs.uniform(-0.1, 0.1, shape).astype(np.float32) if perturb else np.zeros(shape, np.float32)
        elif kind == "rm":
            v = np.zeros(shape, np.float32)
        elif kind == "rv":
            v = np.ones(shape, np.float32)
        elif kind == "nbt":
            v = np.zeros(shape, np.int64)
        else:
            raise ValueError(kind)
        out[key] = torch.from_numpy(np.ascontiguousarray(v))
    return out


def param_keys(spec) -> List[str]:
    return [k for k, _, kind in spec if kind in ("w", "wn", "b", "gamma", "beta")]


def synth_batch(batch: int, cfg: ArchCfg, n_voxels: int = 0, seed: int = 1234, steps: int = 1):
    """Synthetic inputs: x~U[-1,1], fmri~N(0,1), noise~N(0,1), all numpy RandomState.

    ``noise`` has shape (steps, 4, B, z): per step [0]=eps, [1]=z_p, [2]=eps_teacher / z_fake, [3]=spare."""
    x = np.random.RandomState(seed).uniform(-1, 1, (batch, 3, cfg.image_size, cfg.image_size)).astype(np.float32)
    out = {"x": torch.from_numpy(x)}
    if n_voxels:
        out["fmri"] = torch.from_numpy(
            np.random.RandomState(seed + 1).standard_normal((batch, n_voxels)).astype(np.float32))
    nz = np.random.RandomState(seed + 2).standard_normal((steps, 4, batch, cfg.latent_dim)).astype(np.float32)
    out["noise"] = torch.from_numpy(nz)
    return out


# ----------------------------------------------------------------------------------------------
# 16-bit storage model (tests only)
# ----------------------------------------------------------------------------------------------
# STORAGE16 = True makes the forward passes below round every tensor the HIP engine STORES in fp16 -- images, GEMM
# weights, conv / linear outputs, BatchNorm+ReLU outputs, latent codes, generated images -- at the point where the
# engine stores it, with a straight-through gradient (the arithmetic stays fp32, like the engine's fp32 accumulators).
# The ReLU masks of such a run are the engine's masks, so its gradients can be compared tightly; the default (False)
# is the plain fp32 restatement that the golden vectors pin.
STORAGE16 = False
# tags of storage points kept in fp32 under STORAGE16 (tests/probe_storage16_levers.py: which 16-bit tensors carry the
# after-one-update distance from fp32): "rec_feat" (the raw conv features the 'REC' call returns), "images" (decoder
# outputs / the discriminator's input batch), "disc_raw" (every raw convolution output of the discriminator)
STORAGE16_KEEP32 = frozenset()


def _q(t: Tensor, tag: str = None) -> Tensor:
    if not STORAGE16 or (tag is not None and tag in STORAGE16_KEEP32):
        return t
    return t + (t.detach().half().float() - t.detach())


# RELU_MASKS (tests only): a list of 0/1 tensors consumed in call order by every ReLU below -- the masks of another
# run of the same network (the HIP engine's stored activations > 0).  With the masks pinned the gradients are a smooth
# function of the activations, so two implementations can be compared at rounding level instead of at the level of
# "which pre-activations within fp16 rounding of zero fell on which side".
RELU_MASKS = None


def _relu(x: Tensor) -> Tensor:
    if RELU_MASKS is None:
        return F.relu(x)
    m = RELU_MASKS.pop(0)
    assert m.shape == x.shape, (tuple(m.shape), tuple(x.shape))
    return x * m


def _w(P: State, key: str) -> Tensor:
    """GEMM weight as the engine multiplies it (fp16 copy of the fp32 master under STORAGE16)."""
    return _q(P[key])


# ----------------------------------------------------------------------------------------------
# sub-network forwards
# ----------------------------------------------------------------------------------------------
def _bn(P: State, pre: str, x: Tensor, train: bool) -> Tensor:
    """BatchNorm with momentum 0.9, eps 1e-5 (models/vae_gan.py:21,54,81,108,158)."""
    y = F.batch_norm(x, P[pre + "running_mean"], P[pre + "running_var"], P[pre + "weight"], P[pre + "bias"],
                     train, 0.9, 1e-5)
    if train:
        P[pre + "num_batches_tracked"] += 1
    return y


def encoder_fwd(P: State, pre: str, x: Tensor, cfg: ArchCfg, train: bool = True):
    """Encoder.forward, models/vae_gan.py:87-93."""
    h = _q(x)
    for i in range(3):
        h = _q(F.conv2d(h, _w(P, f"{pre}conv.{i}.conv.weight"), None, cfg.stride, cfg.padding))
        h = _q(_relu(_bn(P, f"{pre}conv.{i}.bn.", h, train)))
    h = h.reshape(h.shape[0], -1)
    h = _q(_relu(_bn(P, f"{pre}fc.1.", _q(F.linear(h, _w(P, f"{pre}fc.0.weight"))), train)))
    mu = F.linear(h, _w(P, f"{pre}l_mu.weight"), P[f"{pre}l_mu.bias"])
    logvar = F.linear(h, _w(P, f"{pre}l_var.weight"), P[f"{pre}l_var.bias"])
    return mu, logvar


def decoder_fwd(P: State, pre: str, z: Tensor, cfg: ArchCfg, train: bool = True):
    """Decoder.forward, models/vae_gan.py:125-129."""
    h = _q(_relu(_bn(P, f"{pre}fc.1.", _q(F.linear(_q(z), _w(P, f"{pre}fc.0.weight"))), train)))
    h = h.reshape(h.shape[0], -1, cfg.fc_input, cfg.fc_input)
    for i in range(3):
        h = _q(F.conv_transpose2d(h, _w(P, f"{pre}conv.{i}.conv.weight"), None, cfg.stride, cfg.padding,
                                  output_padding=1 if cfg.output_pad_dec[i] else 0))
        h = _q(_relu(_bn(P, f"{pre}conv.{i}.bn.", h, train)))
    h = F.conv2d(h, _w(P, f"{pre}conv.3.0.weight"), P[f"{pre}conv.3.0.bias"], 1, 2)
    return _q(torch.tanh(h), "images")


def discriminator_fwd(P: State, pre: str, x_orig: Tensor, x_pred: Tensor, x_samp: Tensor, mode: str,
                      cfg: ArchCfg, train: bool = True, recon_level: int = 3):
    """Discriminator.forward, models/vae_gan.py:163-183 (mode 'REC' or 'GAN')."""
    h = _q(torch.cat((x_orig, x_pred, x_samp), 0), "images")
    h = _q(_relu(F.conv2d(h, _w(P, f"{pre}conv.0.0.weight"), P[f"{pre}conv.0.0.bias"], cfg.stride_gan, 2)))
    for i in (1, 2, 3):
        raw = F.conv2d(h, _w(P, f"{pre}conv.{i}.conv.weight"), None, cfg.stride, cfg.padding)
        raw = _q(raw, "rec_feat" if i == recon_level and "rec_feat" in STORAGE16_KEEP32 else "disc_raw")
        if mode == "REC" and i == recon_level:
            # reference still runs bn+relu on this block before returning (vae_gan.py:25-30)
            _bn(P, f"{pre}conv.{i}.bn.", raw, train)
            return raw.reshape(raw.shape[0], -1)
        h = _q(_relu(_bn(P, f"{pre}conv.{i}.bn.", raw, train)))
    h = h.reshape(h.shape[0], -1)
    h = _q(_relu(_bn(P, f"{pre}fc.1.", _q(F.linear(h, _w(P, f"{pre}fc.0.weight"))), train)))
    h = F.linear(h, _w(P, f"{pre}fc.3.weight"), P[f"{pre}fc.3.bias"])
    return torch.sigmoid(h)


def cognitive_encoder_fwd(P: State, pre: str, fmri: Tensor, train: bool = True):
    """CognitiveEncoder.forward, models/vae_gan.py:224-229."""
    h = _q(_relu(_bn(P, f"{pre}fc1.1.", _q(F.linear(_q(fmri), _w(P, f"{pre}fc1.0.weight"))), train)))
    mu = F.linear(h, _w(P, f"{pre}l_mu.weight"), P[f"{pre}l_mu.bias"])
    logvar = F.linear(h, _w(P, f"{pre}l_var.weight"), P[f"{pre}l_var.bias"])
    return mu, logvar


def wae_discriminator_fwd(P: State, pre: str, z: Tensor):
    """WaeDiscriminator.forward, models/vae_gan.py:527-529."""
    h = _q(z)
    for idx in (0, 2, 4, 6):
        h = _q(_relu(F.linear(h, _w(P, f"{pre}main.{idx}.weight"), P[f"{pre}main.{idx}.bias"])))
    return torch.sigmoid(F.linear(h, _w(P, f"{pre}main.8.weight"), P[f"{pre}main.8.bias"]))


def reparameterize(mu: Tensor, logvar: Tensor, eps: Tensor) -> Tensor:
    """VaeGan.reparameterize (models/vae_gan.py:266-269) with the normal draw passed explicitly."""
    return eps * torch.exp(0.5 * logvar) + mu


# ----------------------------------------------------------------------------------------------
# wrapper forwards + loss
# ----------------------------------------------------------------------------------------------
def vaegan_forward(P: State, x: Tensor, eps: Tensor, z_p: Tensor, cfg: ArchCfg, pre: str = ""):
    """VaeGan.forward (train branch), models/vae_gan.py:276-287.  RNG order: eps, then z_p."""
    mu, logvar = encoder_fwd(P, pre + "encoder.", x, cfg)
    z = reparameterize(mu, logvar, eps)
    x_tilde = decoder_fwd(P, pre + "decoder.", z, cfg)
    x_p = decoder_fwd(P, pre + "decoder.", z_p, cfg)
    disc_layer = discriminator_fwd(P, pre + "discriminator.", x, x_tilde, x_p, "REC", cfg)
    disc_class = discriminator_fwd(P, pre + "discriminator.", x, x_tilde, x_p, "GAN", cfg)
    return dict(x_tilde=x_tilde, x_p=x_p, disc_class=disc_class, disc_layer=disc_layer, mus=mu, log_variances=logvar)


def vaegan_loss(x, x_tilde, dl_o, dl_p, dl_s, dc_o, dc_p, dc_s, mus, variances):
    """VaeGan.loss == VaeGanCognitive.loss, models/vae_gan.py:302-320 / :411-432."""
    nle = 0.5 * (x.reshape(len(x), -1) - x_tilde.reshape(len(x_tilde), -1)) ** 2
    kl = -0.5 * torch.sum(-variances.exp() - mus.pow(2) + variances + 1, 1)
    mse = torch.sum(0.5 * (dl_o - dl_p) ** 2, 1)
    bce_o = -torch.log(dc_o + 1e-3)
    bce_p = -torch.log(1 - dc_p + 1e-3)
    bce_s = -torch.log(1 - dc_s + 1e-3)
    return nle, kl, mse, bce_o, bce_p, bce_s


def cognitive_forward(P: State, fmri: Tensor, image: Tensor, noise: Tensor, cfg: ArchCfg, stage: int,
                      teacher: bool = True):
    """VaeGanCognitive.forward (train, mode='vae'), models/vae_gan.py:359-395.

    noise[0] = eps of the cognitive encoder, noise[1] = z_p, noise[2] = eps of the teacher encoder
    (only drawn at stage 2).  RNG order in the reference: eps_cog, eps_teacher, z_p.
    Keys: cognitive encoder under ``encoder.``, decoder ``decoder.``, discriminator ``discriminator.``,
    teacher encoder under ``teacher_net.encoder.``.
    """
    mu, logvar = cognitive_encoder_fwd(P, "encoder.", fmri)
    z = reparameterize(mu, logvar, noise[0])
    x_tilde = decoder_fwd(P, "decoder.", z, cfg)
    gt_x = image
    if teacher and stage == 2:
        mu_t, lv_t = encoder_fwd(P, "teacher_net.encoder.", image, cfg)
        gt_x = decoder_fwd(P, "decoder.", reparameterize(mu_t, lv_t, noise[2]), cfg)
    x_p = decoder_fwd(P, "decoder.", noise[1], cfg)
    disc_layer = discriminator_fwd(P, "discriminator.", gt_x, x_tilde, x_p, "REC", cfg)
    disc_class = discriminator_fwd(P, "discriminator.", gt_x, x_tilde, x_p, "GAN", cfg)
    return dict(gt_x=gt_x, x_tilde=x_tilde, x_p=x_p, disc_class=disc_class, disc_layer=disc_layer,
                mus=mu, log_variances=logvar)


def cognitive_forward_wae(P: State, fmri: Tensor, image: Tensor, z_p: Tensor, cfg: ArchCfg):
    """VaeGanCognitive.forward (train, mode='wae'), models/vae_gan.py:379-395: no reparameterisation -- the decoder
    takes the means; the teacher's image encoder (``teacher_net.encoder.``) runs in train mode too (its BatchNorm
    statistics are batch statistics, its running statistics are updated).  Only random draw: z_p."""
    mu, logvar = cognitive_encoder_fwd(P, "encoder.", fmri)
    x_tilde = decoder_fwd(P, "decoder.", mu, cfg)
    mu_t, _ = encoder_fwd(P, "teacher_net.encoder.", image, cfg)
    gt_x = decoder_fwd(P, "decoder.", mu_t, cfg)
    x_p = decoder_fwd(P, "decoder.", z_p, cfg)
    disc_layer = discriminator_fwd(P, "discriminator.", gt_x, x_tilde, x_p, "REC", cfg)
    disc_class = discriminator_fwd(P, "discriminator.", gt_x, x_tilde, x_p, "GAN", cfg)
    return dict(gt_x=gt_x, x_tilde=x_tilde, disc_class=disc_class, disc_layer=disc_layer, mus=mu, log_variances=logvar)


def wae_cognitive_eval(P: State, fmri: Tensor, cfg: ArchCfg):
    """WaeGanCognitive.forward in eval mode (models/vae_gan.py:568-571): decoder(encoder(fmri).mu), running statistics."""
    mu, _ = cognitive_encoder_fwd(P, "encoder.", fmri, train=False)
    return decoder_fwd(P, "decoder.", mu, cfg, train=False)


def dcgan_forward(P: State, x: Tensor, z_p: Tensor, cfg: ArchCfg, train: bool = True):
    """DCGan.forward (models/vae_gan.py:602-622).  Train: the generated batch takes BOTH the 'predicted' and the
    'sampled' slot of the discriminator; eval / sample=None: decoder(z_p)."""
    if not train:
        return dict(x_p=decoder_fwd(P, "decoder.", z_p, cfg, train=False))
    x_tilde = decoder_fwd(P, "decoder.", z_p, cfg)
    disc_layer = discriminator_fwd(P, "discriminator.", x, x_tilde, x_tilde, "REC", cfg)
    disc_class = discriminator_fwd(P, "discriminator.", x, x_tilde, x_tilde, "GAN", cfg)
    return dict(gt_x=x, x_tilde=x_tilde, disc_class=disc_class, disc_layer=disc_layer)


# ----------------------------------------------------------------------------------------------
# optimizers (functional restatement of torch.optim.RMSprop / Adam as the scripts configure them)
# ----------------------------------------------------------------------------------------------
@dataclass
class OptState:
    kind: str = "rmsprop"          # 'rmsprop' (train_vgan_stage1.py:275) or 'adam' (train_wae_stage1.py:221)
    lr: float = 1e-4
    alpha: float = 0.9
    eps: float = 1e-8
    betas: Tuple[float, float] = (0.5, 0.999)
    step: int = 0
    bufs: Dict[str, Tensor] = field(default_factory=dict)


@torch.no_grad()
def opt_step(P: State, keys: List[str], grads: List[Tensor], opt: OptState, clamp: float = None):
    opt.step += 1
    for k, g in zip(keys, grads):
        if g is None:
            continue
        if clamp is not None:
            g = g.clamp(-clamp, clamp)
        p = P[k]
        if opt.kind == "rmsprop":
            sq = opt.bufs.setdefault(k, torch.zeros_like(p))
            sq.mul_(opt.alpha).addcmul_(g, g, value=1 - opt.alpha)
            p.addcdiv_(g, sq.sqrt().add_(opt.eps), value=-opt.lr)
        else:  # adam, amsgrad off, weight_decay 0
            b1, b2 = opt.betas
            m = opt.bufs.setdefault(k + "/m", torch.zeros_like(p))
            v = opt.bufs.setdefault(k + "/v", torch.zeros_like(p))
            m.mul_(b1).add_(g, alpha=1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            bc1 = 1 - b1 ** opt.step
            bc2 = 1 - b2 ** opt.step
            denom = (v.sqrt() / math.sqrt(bc2)).add_(opt.eps)
            p.addcdiv_(m, denom, value=-opt.lr / bc1)


def _leafify(P: State, keys: List[str]):
    for k in keys:
        P[k] = P[k].detach().requires_grad_(True)


def _grads(loss: Tensor, P: State, keys: List[str], retain: bool, literal_all: List[str] = None):
    """Gradient of ``loss`` w.r.t. P[keys] at the current (pre-update) weights.

    ``literal_all``: also differentiate w.r.t. every other trainable key and throw those away --
    this reproduces the cost of the reference's full ``loss.backward()`` traversal
    (train_vgan_stage1.py:412,422,430) for the CPU-baseline timing; values are identical.
    """
    if literal_all is not None:
        allk = list(keys) + [k for k in literal_all if k not in keys]
        g = torch.autograd.grad(loss, [P[k] for k in allk], retain_graph=retain, allow_unused=True)
        return list(g[:len(keys)])
    return list(torch.autograd.grad(loss, [P[k] for k in keys], retain_graph=retain, allow_unused=True))


@dataclass
class GanHyper:
    """configs/gan_config.py:19-31 defaults consumed by the step bodies."""
    lr: float = 1e-4
    lambda_mse: float = 1e-6
    margin: float = 0.35
    equilibrium: float = 0.68


def equilibrium_gate(bce_o_mean: float, bce_p_mean: float, hp: GanHyper):
    """train_vgan_stage1.py:396-404."""
    train_dis, train_dec = True, True
    if bce_o_mean < hp.equilibrium - hp.margin or bce_p_mean < hp.equilibrium - hp.margin:
        train_dis = False
    if bce_o_mean > hp.equilibrium + hp.margin or bce_p_mean > hp.equilibrium + hp.margin:
        train_dec = False
    if (not train_dec) and (not train_dis):
        train_dis, train_dec = True, True
    return train_dis, train_dec


def _compose_losses(fw, x_real, B, hp: GanHyper, mode: str = "vae-gan", beta: float = 1.0):
    """VaeGan.loss + the loss compositions of train_vgan_stage1.py:359-388 (``mode``)."""
    dl, dc = fw["disc_layer"], fw["disc_class"]
    nle, kl, mse, bo, bp, bs = vaegan_loss(_q(x_real), fw["x_tilde"], dl[:B], dl[B:-B], dl[-B:], dc[:B], dc[B:-B],
                                           dc[-B:], fw["mus"], fw["log_variances"])
    if mode == "beta-vae":                                                               # :360-366
        loss_enc = torch.sum(kl) * beta * (1.0 / B) + torch.sum(mse)
        loss_dis = torch.sum(bo) + torch.sum(bp) + torch.sum(bs)
        loss_dec = torch.sum(hp.lambda_mse * mse) - (1.0 - hp.lambda_mse) * loss_dis
    elif mode == "vae-gan":                                                              # :369-372
        loss_enc = torch.sum(kl) + torch.sum(mse)
        loss_dis = torch.sum(bo) + torch.sum(bp) + torch.sum(bs)
        loss_dec = torch.sum(hp.lambda_mse * mse) - (1.0 - hp.lambda_mse) * loss_dis
    elif mode == "dcgan":                                                                # :375-381
        loss_enc = torch.sum(kl) + torch.sum(nle)
        loss_dis = torch.sum(bo) + torch.sum(bs)
        loss_dec = torch.sum(hp.lambda_mse * nle) - (1.0 - hp.lambda_mse) * loss_dis
    elif mode == "vae":                                                                  # :384-388
        loss_enc = torch.sum(kl) + torch.sum(nle)
        loss_dis = torch.sum(bo) + torch.sum(bs)
        loss_dec = torch.sum(hp.lambda_mse * nle)
    else:
        raise ValueError(mode)
    logs = dict(loss_encoder=loss_enc.item(), loss_discriminator=loss_dis.item(), loss_decoder=loss_dec.item(),
                nle=torch.sum(nle).item(), kl=torch.sum(kl).item(), mse=torch.sum(mse).item(),
                bce_orig=torch.sum(bo).item(), bce_pred=torch.sum(bp).item(), bce_samp=torch.sum(bs).item())
    return loss_enc, loss_dis, loss_dec, logs


def stage1_step(P: State, opts: Dict[str, OptState], x: Tensor, eps: Tensor, z_p: Tensor, cfg: ArchCfg,
                hp: GanHyper = GanHyper(), literal: bool = False, keep_grads: bool = False, mode: str = "vae-gan",
                beta: float = 1.0):
    """One Stage-I VAE/GAN step (train/train_vgan_stage1.py:330-432), ``mode`` in 'vae-gan' (default), 'beta-vae',
    'dcgan' (encoder not trained), 'vae' (discriminator not trained unless the gate re-arms both).

    One forward, three gradient sets at the pre-update weights, gate, three RMSprop steps.
    """
    spec = vaegan_spec(cfg)
    enc_k = param_keys(encoder_spec(cfg))
    dec_k = param_keys(decoder_spec(cfg))
    dis_k = param_keys(discriminator_spec(cfg))
    _leafify(P, enc_k + dec_k + dis_k)
    B = x.shape[0]
    fw = vaegan_forward(P, x, eps, z_p, cfg)
    loss_enc, loss_dis, loss_dec, logs = _compose_losses(fw, x, B, hp, mode, beta)
    train_enc = mode != "dcgan"                                                          # :376
    train_dis, train_dec = mode != "vae", True                                           # :353-354, :388
    if logs["bce_orig"] / B < hp.equilibrium - hp.margin or logs["bce_pred"] / B < hp.equilibrium - hp.margin:
        train_dis = False                                                                # :396-398
    if logs["bce_orig"] / B > hp.equilibrium + hp.margin or logs["bce_pred"] / B > hp.equilibrium + hp.margin:
        train_dec = False                                                                # :399-401
    if (not train_dec) and (not train_dis):
        train_dis, train_dec = True, True                                                # :402-404
    allk = enc_k + dec_k + dis_k if literal else None
    g_enc = _grads(loss_enc, P, enc_k, True, allk)
    g_dec = _grads(loss_dec, P, dec_k, True, allk)
    g_dis = _grads(loss_dis, P, dis_k, False, allk)
    if train_enc:
        opt_step(P, enc_k, g_enc, opts["encoder"])
    if train_dec:
        opt_step(P, dec_k, g_dec, opts["decoder"])
    if train_dis:
        opt_step(P, dis_k, g_dis, opts["discriminator"])
    for k in enc_k + dec_k + dis_k:
        P[k] = P[k].detach()
    logs.update(train_dis=train_dis, train_dec=train_dec)
    out = dict(logs=logs, fw={k: v.detach() for k, v in fw.items()})
    if keep_grads:
        out["grads"] = {**dict(zip(enc_k, g_enc)), **dict(zip(dec_k, g_dec)), **dict(zip(dis_k, g_dis))}
    return out


def stage2_step(P: State, opts: Dict[str, OptState], fmri: Tensor, image: Tensor, noise: Tensor, cfg: ArchCfg,
                n_voxels: int, hp: GanHyper = GanHyper(), keep_grads: bool = False, mode: str = "vae-gan"):
    """Stage-II step (train/train_vgan_stage2.py:321-407): decoder frozen, teacher distillation,
    encoder + discriminator trained, no gate, gradients clamped to +-1.

    ``mode='vae'`` (:234-238, :362-366): the model is built without a teacher net (the "real" slot of the discriminator
    is the ground-truth image), the encoder minimises KL + the PIXEL nle, the discriminator bce_orig + bce_sampled; the
    `train_dis = False` of :366 is overwritten by :375-376, so the discriminator is trained in this mode too."""
    if mode not in ("vae-gan", "vae"):
        raise ValueError(mode)
    enc_k = param_keys(cognitive_encoder_spec(cfg, n_voxels))
    dis_k = param_keys(discriminator_spec(cfg))
    _leafify(P, enc_k + dis_k)
    B = fmri.shape[0]
    fw = cognitive_forward(P, fmri, image, noise, cfg, stage=2, teacher=mode != "vae")
    loss_enc, loss_dis, loss_dec, logs = _compose_losses(fw, fw["gt_x"], B, hp, mode)
    g_enc = _grads(loss_enc, P, enc_k, True)
    g_dis = _grads(loss_dis, P, dis_k, False)
    opt_step(P, enc_k, g_enc, opts["encoder"], clamp=1.0)
    opt_step(P, dis_k, g_dis, opts["discriminator"], clamp=1.0)
    for k in enc_k + dis_k:
        P[k] = P[k].detach()
    logs.update(train_dis=True, train_dec=False)
    out = dict(logs=logs, fw={k: v.detach() for k, v in fw.items()})
    if keep_grads:
        out["grads"] = {**dict(zip(enc_k, g_enc)), **dict(zip(dis_k, g_dis))}
    return out


def stage3_step(P: State, opts: Dict[str, OptState], fmri: Tensor, image: Tensor, noise: Tensor, cfg: ArchCfg,
                n_voxels: int, hp: GanHyper = GanHyper(), keep_grads: bool = False, mode: str = "vae-gan"):
    """Stage-III step (train/train_vgan_stage3.py:324-411): cognitive encoder frozen, decoder +
    discriminator trained, equilibrium gate on, gradients clamped to +-1.

    ``mode='vae'`` (:370-374): decoder loss lambda * nle, discriminator loss bce_orig + bce_sampled with train_dis
    starting False -- the discriminator is updated only in a step whose gate re-arms both (:382-389)."""
    if mode not in ("vae-gan", "vae"):
        raise ValueError(mode)
    dec_k = param_keys(decoder_spec(cfg))
    dis_k = param_keys(discriminator_spec(cfg))
    _leafify(P, dec_k + dis_k)
    B = fmri.shape[0]
    fw = cognitive_forward(P, fmri, image, noise, cfg, stage=3, teacher=False)
    loss_enc, loss_dis, loss_dec, logs = _compose_losses(fw, image, B, hp, mode)
    train_dis, train_dec = mode != "vae", True                                           # :356-357, :374
    if logs["bce_orig"] / B < hp.equilibrium - hp.margin or logs["bce_pred"] / B < hp.equilibrium - hp.margin:
        train_dis = False
    if logs["bce_orig"] / B > hp.equilibrium + hp.margin or logs["bce_pred"] / B > hp.equilibrium + hp.margin:
        train_dec = False
    if (not train_dec) and (not train_dis):
        train_dis, train_dec = True, True
    g_dec = _grads(loss_dec, P, dec_k, True)
    g_dis = _grads(loss_dis, P, dis_k, False)
    if train_dec:
        opt_step(P, dec_k, g_dec, opts["decoder"], clamp=1.0)
    if train_dis:
        opt_step(P, dis_k, g_dis, opts["discriminator"], clamp=1.0)
    for k in dec_k + dis_k:
        P[k] = P[k].detach()
    logs.update(train_dis=train_dis, train_dec=train_dec)
    out = dict(logs=logs, fw={k: v.detach() for k, v in fw.items()})
    if keep_grads:
        out["grads"] = {**dict(zip(dec_k, g_dec)), **dict(zip(dis_k, g_dis))}
    return out


def wae_stage1_step(P: State, opts: Dict[str, OptState], x: Tensor, z_fake_noise: Tensor, cfg: ArchCfg,
                    keep_grads: bool = False):
    """WAE Stage-I step (train/train_wae_stage1.py:259-311).

    Phase D: latent discriminator on encoder means vs 0.5*N(0,1) (:275-288); phase G: encoder+decoder on
    pixel reconstruction + GAN penalty with the *updated* discriminator (:296-311).
    """
    enc_k = param_keys(encoder_spec(cfg))
    dec_k = param_keys(decoder_spec(cfg))
    dis_k = param_keys(wae_discriminator_spec(cfg))
    _leafify(P, dis_k)
    with torch.no_grad():
        z_real, _ = encoder_fwd(P, "encoder.", x, cfg)          # frozen encoder, still train-mode BN
    z_fake = z_fake_noise * 0.5                                 # :276
    d_real = wae_discriminator_fwd(P, "discriminator.", z_real)
    d_fake = wae_discriminator_fwd(P, "discriminator.", z_fake)
    l_fake = -10 * torch.sum(torch.log(d_fake + 1e-3))          # :281
    l_real = -10 * torch.sum(torch.log(1 - d_real + 1e-3))      # :282
    # two separate backward passes accumulate into .grad in the reference (:283-284)
    g_f, g_r = _grads(l_fake, P, dis_k, True), _grads(l_real, P, dis_k, False)
    g_dis = [a + b for a, b in zip(g_f, g_r)]
    dis_terms = {k: float(a.norm() + b.norm()) for k, a, b in zip(dis_k, g_f, g_r)}
    opt_step(P, dis_k, g_dis, opts["discriminator"])
    for k in dis_k:
        P[k] = P[k].detach()
    _leafify(P, enc_k + dec_k)
    z_real, _ = encoder_fwd(P, "encoder.", x, cfg)              # :296, second encoder pass
    x_recon = decoder_fwd(P, "decoder.", z_real, cfg)
    d_real2 = wae_discriminator_fwd(P, "discriminator.", z_real)
    l_rec = torch.sum(torch.sum(0.5 * (x_recon - x) ** 2, 1))   # :301
    l_pen = -10 * torch.sum(torch.log(d_real2 + 1e-3))          # :303
    g_rec = _grads(l_rec, P, enc_k + dec_k, True)               # :306
    g_pen = _grads(l_pen, P, enc_k + dec_k, False)              # :307 (decoder gets no penalty gradient)
    g = [a if b is None else a + b for a, b in zip(g_rec, g_pen)]
    opt_step(P, enc_k, g[:len(enc_k)], opts["encoder"])
    opt_step(P, dec_k, g[len(enc_k):], opts["decoder"])
    for k in enc_k + dec_k:
        P[k] = P[k].detach()
    logs = dict(loss_reconstruction=l_rec.item(), loss_penalty=l_pen.item(),
                loss_discriminator_fake=l_fake.item(), loss_discriminator_real=l_real.item())
    out = dict(logs=logs, fw=dict(x_recon=x_recon.detach(), z_real=z_real.detach()))
    if keep_grads:
        out["grads"] = {**dict(zip(dis_k, g_dis)), **dict(zip(enc_k + dec_k, g))}
        out["grad_terms"] = dis_terms
    return out


def _gsum(a: List[Tensor], b: List[Tensor]) -> List[Tensor]:
    """Element-wise sum of two gradient lists where None = "no gradient" (e.g. l_var under a mean-only latent)."""
    return [y if x is None else (x if y is None else x + y) for x, y in zip(a, b)]


def _wae_dis_phase(P: State, opt: OptState, z_real: Tensor, z_fake: Tensor, dis_k: List[str], lam: float,
                   pre: str = "discriminator."):
    """Latent-discriminator phase shared by every WAE script (e.g. train/train_wae_stage2.py:297-307): two
    separate backward passes accumulate into .grad, then one optimizer step."""
    _leafify(P, dis_k)
    d_real = wae_discriminator_fwd(P, pre, z_real)
    d_fake = wae_discriminator_fwd(P, pre, z_fake)
    l_fake = -lam * torch.sum(torch.log(d_fake + 1e-3))
    l_real = -lam * torch.sum(torch.log(1 - d_real + 1e-3))
    g_f, g_r = _grads(l_fake, P, dis_k, True), _grads(l_real, P, dis_k, False)
    g_dis = [a + b for a, b in zip(g_f, g_r)]
    # ||g_fake|| + ||g_real|| per tensor: the "fake" and "real" passes push the discriminator in opposite directions
    # and can cancel 20-50 x in the sum; tests bound errors against the magnitude of the terms, not of the remainder
    _wae_dis_phase.last_terms = {k: float(a.norm() + b.norm()) for k, a, b in zip(dis_k, g_f, g_r)}
    opt_step(P, dis_k, g_dis, opt)
    for k in dis_k:
        P[k] = P[k].detach()
    return l_fake, l_real, g_dis


def wae_stage2_step(P: State, opts: Dict[str, OptState], fmri: Tensor, image: Tensor, cfg: ArchCfg, n_voxels: int,
                    keep_grads: bool = False):
    """WAE Stage-II step (train/train_wae_stage2.py:276-328): cognitive encoder + latent discriminator trained,
    decoder frozen (but in train mode: BN batch statistics, running stats updated by both decoder calls), the
    Stage-I image encoder `teacher_net.encoder.` gives the "real" latents (train mode as well, called twice)."""
    enc_k = param_keys(cognitive_encoder_spec(cfg, n_voxels))
    dis_k = param_keys(wae_discriminator_spec(cfg))
    with torch.no_grad():
        z_t, _ = encoder_fwd(P, "teacher_net.encoder.", image, cfg)       # :284
        decoder_fwd(P, "decoder.", z_t, cfg)                               # :285 x_gt: unused, BN side effects only
        z_fake, _ = cognitive_encoder_fwd(P, "encoder.", fmri)             # :292
        z_real, _ = encoder_fwd(P, "teacher_net.encoder.", image, cfg)     # :293
    l_fake, l_real, g_dis = _wae_dis_phase(P, opts["discriminator"], z_real, z_fake, dis_k, 10.0)
    _leafify(P, enc_k)
    z, _ = cognitive_encoder_fwd(P, "encoder.", fmri)                      # :314
    x_recon = decoder_fwd(P, "decoder.", z, cfg)
    d = wae_discriminator_fwd(P, "discriminator.", z)
    l_rec = F.mse_loss(x_recon, image)                                     # :320
    l_pen = -10 * torch.mean(torch.log(d + 1e-3))                          # :321
    g = _gsum(_grads(l_rec, P, enc_k, True), _grads(l_pen, P, enc_k, False))
    opt_step(P, enc_k, g, opts["encoder"])
    for k in enc_k:
        P[k] = P[k].detach()
    logs = dict(loss_reconstruction=l_rec.item(), loss_penalty=l_pen.item(),
                loss_discriminator_fake=l_fake.item(), loss_discriminator_real=l_real.item())
    out = dict(logs=logs, fw=dict(x_recon=x_recon.detach(), z_real=z.detach()))
    if keep_grads:
        out["grads"] = {**dict(zip(dis_k, g_dis)), **dict(zip(enc_k, g))}
        out["grad_terms"] = dict(_wae_dis_phase.last_terms)
    return out


def wae_stage3_step(P: State, opts: Dict[str, OptState], fmri: Tensor, image: Tensor, cfg: ArchCfg, n_voxels: int,
                    keep_grads: bool = False):
    """WAE Stage-III step (train/train_wae_stage3.py:297-347): cognitive encoder frozen (train-mode BN, two calls),
    latent discriminator and decoder trained; the penalty is computed but never back-propagated (:344)."""
    dec_k = param_keys(decoder_spec(cfg))
    dis_k = param_keys(wae_discriminator_spec(cfg))
    with torch.no_grad():
        z_fake, _ = cognitive_encoder_fwd(P, "encoder.", fmri)             # :311
        z_real, _ = encoder_fwd(P, "teacher_net.encoder.", image, cfg)     # :312
    l_fake, l_real, g_dis = _wae_dis_phase(P, opts["discriminator"], z_real, z_fake, dis_k, 10.0)
    _leafify(P, dec_k)
    with torch.no_grad():
        z, _ = cognitive_encoder_fwd(P, "encoder.", fmri)                  # :333
        d = wae_discriminator_fwd(P, "discriminator.", z)
    x_recon = decoder_fwd(P, "decoder.", z, cfg)
    l_rec = F.mse_loss(x_recon, image)                                     # :339
    l_pen = -10 * torch.mean(torch.log(d + 1e-3))                          # :340 (logged only)
    g_dec = _grads(l_rec, P, dec_k, False)
    opt_step(P, dec_k, g_dec, opts["decoder"])
    for k in dec_k:
        P[k] = P[k].detach()
    logs = dict(loss_reconstruction=l_rec.item(), loss_penalty=l_pen.item(),
                loss_discriminator_fake=l_fake.item(), loss_discriminator_real=l_real.item())
    out = dict(logs=logs, fw=dict(x_recon=x_recon.detach(), z_real=z.detach()))
    if keep_grads:
        out["grads"] = {**dict(zip(dis_k, g_dis)), **dict(zip(dec_k, g_dec))}
        out["grad_terms"] = dict(_wae_dis_phase.last_terms)
    return out


def dual_stage1_step(P: State, opts: Dict[str, OptState], x: Tensor, noise: Tensor, cfg: ArchCfg,
                     hp: GanHyper = GanHyper(), lam: float = 1.0, keep_grads: bool = False, mode: str = "vae-gan",
                     beta: float = 1.0):
    """Dual WAE + VAE/GAN Stage-I step (train/wae_vgan_stage1.py:284-441), ``mode`` in 'vae-gan' (default), 'beta-vae',
    'dcgan' (encoder never stepped, :419), 'vae' (train_dis starts False) -- the compositions of :311-364.

    = the Stage-I VAE/GAN forward and losses (:290-364), then a WAE latent-discriminator phase on the encoder
    means (`wae_discriminator.`, RMSprop, :384-397), then the penalty -lam*sum log(d_real+1e-3) back-propagated
    into the encoder from a third encoder pass (:405-413; the decoder call :406 only moves BN statistics and the
    `optimizer_decoder.step()` of :417 sees no new gradient), and finally the three gated VAE/GAN updates with the
    penalty gradient still sitting in the encoder's .grad (:419-441).  noise = [eps, z_p, z_fake] (3, B, z).

    torch-1.4 detail (the pinned version): `zero_grad()` zeroes existing .grad tensors instead of dropping them,
    so from the second iteration on the RMSprop step of :417 runs with an all-zero decoder gradient -- parameters
    stay put but `square_avg` decays by alpha once more.  Restated here as that extra decay."""
    enc_k = param_keys(encoder_spec(cfg))
    dec_k = param_keys(decoder_spec(cfg))
    dis_k = param_keys(discriminator_spec(cfg))
    wd_k = param_keys(wae_discriminator_spec(cfg, pre="wae_discriminator."))
    _leafify(P, enc_k + dec_k + dis_k)
    B = x.shape[0]
    fw = vaegan_forward(P, x, noise[0], noise[1], cfg)
    loss_enc, loss_dis, loss_dec, logs = _compose_losses(fw, x, B, hp, mode, beta)
    train_enc = mode != "dcgan"                                            # :343
    train_dis, train_dec = mode != "vae", True                             # :311-312, :354
    if logs["bce_orig"] / B < hp.equilibrium - hp.margin or logs["bce_pred"] / B < hp.equilibrium - hp.margin:
        train_dis = False
    if logs["bce_orig"] / B > hp.equilibrium + hp.margin or logs["bce_pred"] / B > hp.equilibrium + hp.margin:
        train_dec = False
    if (not train_dec) and (not train_dis):
        train_dis, train_dec = True, True
    with torch.no_grad():
        z_real, _ = encoder_fwd(P, "encoder.", x, cfg)                     # :384
    z_fake = noise[2] * 0.5                                                # :385
    l_fake, l_real, g_wd = _wae_dis_phase(P, opts["wae_discriminator"], z_real, z_fake, wd_k, lam,
                                          pre="wae_discriminator.")
    z_real2, _ = encoder_fwd(P, "encoder.", x, cfg)                        # :405
    with torch.no_grad():
        decoder_fwd(P, "decoder.", z_real2, cfg)                           # :406 x_recon: unused, BN side effects
    d_real = wae_discriminator_fwd(P, "wae_discriminator.", z_real2)
    l_pen = -lam * torch.sum(torch.log(d_real + 1e-3))                     # :411
    g_pen = _grads(l_pen, P, enc_k, False)
    dopt = opts["decoder"]
    if dopt.step > 0:                                                      # :417 with zeroed grads (torch 1.4)
        with torch.no_grad():
            for k in dec_k:
                if k in dopt.bufs:
                    dopt.bufs[k].mul_(dopt.alpha)
    g_enc = _gsum(_grads(loss_enc, P, enc_k, True), g_pen) if train_enc else [None] * len(enc_k)   # :421
    g_dec = _grads(loss_dec, P, dec_k, True)
    g_dis = _grads(loss_dis, P, dis_k, False)
    if train_enc:
        opt_step(P, enc_k, g_enc, opts["encoder"])
    if train_dec:
        opt_step(P, dec_k, g_dec, opts["decoder"])
    if train_dis:
        opt_step(P, dis_k, g_dis, opts["discriminator"])
    for k in enc_k + dec_k + dis_k:
        P[k] = P[k].detach()
    logs.update(train_dis=train_dis, train_dec=train_dec, loss_penalty=l_pen.item(),
                loss_discriminator_fake=l_fake.item(), loss_discriminator_real=l_real.item())
    out = dict(logs=logs, fw={k: v.detach() for k, v in fw.items()})
    if keep_grads:
        out["grads"] = {**dict(zip(enc_k, g_enc)), **dict(zip(dec_k, g_dec)), **dict(zip(dis_k, g_dis)),
                        **dict(zip(wd_k, g_wd))}
    return out


# ----------------------------------------------------------------------------------------------
# summaries used by golden fixtures
# ----------------------------------------------------------------------------------------------
def tensor_summary(t: Tensor, n: int = 8) -> np.ndarray:
    """[L2 norm, sum, first n, last n] as float64 -- a compact fingerprint for golden fixtures."""
    f = t.detach().double().reshape(-1)
    head = f[:n]
    tail = f[-n:]
    pad = n - head.numel()
    if pad > 0:
        head = torch.cat([head, torch.zeros(pad, dtype=torch.float64)])
        tail = torch.cat([tail, torch.zeros(pad, dtype=torch.float64)])
    return torch.cat([f.norm().reshape(1), f.sum().reshape(1), head, tail]).numpy()
