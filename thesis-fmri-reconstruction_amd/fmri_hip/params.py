"""Architecture description + flat fp32 parameter storage of the engine.

The model classes of the reference read ``configs.models_config`` at construction time
(models/vae_gan.py:8,18,74,79,107,112-119,146-157); ``ArchConfig.from_module`` does the same so the
engine is config-driven.  Parameter names / shapes / order are exactly the reference ``state_dict()``
 so checkpoints are interchangeable.

Every sub-network keeps its parameters in ONE flat fp32 buffer (``FlatGroup``): the optimizer is one
fused kernel launch per sub-network and the data-parallel gradient exchange is one RCCL all-reduce per
sub-network (SUM, because the reference losses are batch sums -- train_vgan_stage1.py:369-372).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np
import torch


@dataclass(frozen=True)
class ArchConfig:
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
    def from_module(mc) -> "ArchConfig":
        """Build from a ``configs.models_config``-style namespace (same attribute names)."""
        return ArchConfig(
            image_size=int(mc.image_size), fc_input=int(mc.fc_input), fc_output=int(mc.fc_output),
            fc_input_gan=int(mc.fc_input_gan), fc_output_gan=int(mc.fc_output_gan), stride_gan=int(mc.stride_gan),
            latent_dim=int(mc.latent_dim), output_pad_dec=tuple(bool(v) for v in mc.output_pad_dec),
            encoder_channels=tuple(mc.encoder_channels), decoder_channels=tuple(mc.decoder_channels),
            discrim_channels=tuple(mc.discrim_channels), kernel_size=int(mc.kernel_size), stride=int(mc.stride),
            padding=int(mc.padding))

    @staticmethod
    def px64() -> "ArchConfig":
        return ArchConfig()

    @staticmethod
    def px100() -> "ArchConfig":
        return ArchConfig(image_size=100, fc_input=13, fc_output=1024, fc_input_gan=7, fc_output_gan=256,
                          stride_gan=2, latent_dim=512, output_pad_dec=(False, True, True),
                          decoder_channels=(256, 128, 64, 3))

    @staticmethod
    def px128() -> "ArchConfig":
        return ArchConfig(image_size=128, fc_input=16, fc_input_gan=8, fc_output_gan=512, stride_gan=2)


# ------------------------------------------------------------------------------------------------
# state-dict specs: (key, shape, kind); kind in w|wn|b|gamma|beta|rm|rv|nbt
# ------------------------------------------------------------------------------------------------
def _bn(pre, c):
    return [(pre + "weight", (c,), "gamma"), (pre + "bias", (c,), "beta"), (pre + "running_mean", (c,), "rm"),
            (pre + "running_var", (c,), "rv"), (pre + "num_batches_tracked", (), "nbt")]


def encoder_spec(cfg: ArchConfig, channel_in: int = 3):
    k, out, cin = cfg.kernel_size, [], channel_in
    for i, c in enumerate(cfg.encoder_channels[:3]):
        out.append((f"conv.{i}.conv.weight", (c, cin, k, k), "w"))
        out += _bn(f"conv.{i}.bn.", c)
        cin = c
    out.append(("fc.0.weight", (cfg.fc_output, cfg.fc_input * cfg.fc_input * cin), "w"))
    out += _bn("fc.1.", cfg.fc_output)
    for h in ("l_mu", "l_var"):
        out += [(f"{h}.weight", (cfg.latent_dim, cfg.fc_output), "w"), (f"{h}.bias", (cfg.latent_dim,), "b")]
    return out


def decoder_spec(cfg: ArchConfig, size: int = None):
    k = cfg.kernel_size
    size = cfg.encoder_channels[2] if size is None else size
    feat = cfg.fc_input * cfg.fc_input * size
    out = [("fc.0.weight", (feat, cfg.latent_dim), "w")] + _bn("fc.1.", feat)
    chans = [(size, size), (size, cfg.decoder_channels[1]), (cfg.decoder_channels[1], cfg.decoder_channels[2])]
    for i, (ci, co) in enumerate(chans):
        out.append((f"conv.{i}.conv.weight", (ci, co, k, k), "w"))
        out += _bn(f"conv.{i}.bn.", co)
    out += [("conv.3.0.weight", (cfg.decoder_channels[3], cfg.decoder_channels[2], 5, 5), "w"),
            ("conv.3.0.bias", (cfg.decoder_channels[3],), "b")]
    return out


def discriminator_spec(cfg: ArchConfig):
    k, d = cfg.kernel_size, cfg.discrim_channels
    out = [("conv.0.0.weight", (d[0], 3, 5, 5), "w"), ("conv.0.0.bias", (d[0],), "b")]
    cin = d[0]
    for i in (1, 2, 3):
        out.append((f"conv.{i}.conv.weight", (d[i], cin, k, k), "w"))
        out += _bn(f"conv.{i}.bn.", d[i])
        cin = d[i]
    out.append(("fc.0.weight", (cfg.fc_output_gan, cfg.fc_input_gan * cfg.fc_input_gan * cin), "w"))
    out += _bn("fc.1.", cfg.fc_output_gan)
    out += [("fc.3.weight", (1, cfg.fc_output_gan), "w"), ("fc.3.bias", (1,), "b")]
    return out


def cognitive_encoder_spec(cfg: ArchConfig, n_voxels: int):
    out = [("fc1.0.weight", (1024, n_voxels), "w")] + _bn("fc1.1.", 1024)
    for h in ("l_mu", "l_var"):
        out += [(f"{h}.weight", (cfg.latent_dim, 1024), "w"), (f"{h}.bias", (cfg.latent_dim,), "b")]
    return out


def wae_discriminator_spec(cfg: ArchConfig, dim_h: int = 512):
    dims = [cfg.latent_dim, dim_h, dim_h, dim_h, dim_h, 1]
    out = []
    for j, idx in enumerate((0, 2, 4, 6, 8)):
        out += [(f"main.{idx}.weight", (dims[j + 1], dims[j]), "wn"), (f"main.{idx}.bias", (dims[j + 1],), "b")]
    return out


def recipe_fill(spec, rs: np.random.RandomState, perturb: bool = False) -> Dict[str, np.ndarray]:
    """Deterministic init recipe (numpy MT19937): U(+-1/sqrt(fan)/sqrt(3)) weights as
    ``VaeGan.init_parameters`` (models/vae_gan.py:252-264); N(0, 0.0099999) for the WAE latent
    discriminator (:522-525); biases 0, gamma 1, beta 0 (optionally jittered)."""
    out = {}
    for key, shape, kind in spec:
        if kind == "w":
            s = 1.0 / math.sqrt(float(np.prod(shape[1:]))) / math.sqrt(3.0)
            v = rs.uniform(-s, s, shape).astype(np.float32)
        elif kind == "wn":
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
        else:
            v = np.zeros(shape, np.int64)
        out[key] = v
    return out


class FlatGroup:
    """Flat fp32 storage for one sub-network: trainable params (+grads) in one buffer, BN buffers aside."""

    def __init__(self, spec, device):
        self.spec = list(spec)
        self.device = torch.device(device)
        self.pkeys = [k for k, _, kind in spec if kind in ("w", "wn", "b", "gamma", "beta")]
        sizes = {k: int(np.prod(s)) if len(s) else 1 for k, s, _ in spec}
        # Memory order = spec order, except that the two latent heads sit side by side (l_mu.weight | l_var.weight, then
        # l_mu.bias | l_var.bias): the fused N = 2z head GEMM then works on VIEWS of the masters and their gradients
        # (nets.FusedHeads) instead of concatenated copies.  Keys and state-dict order are untouched.
        order = list(self.pkeys)
        heads = ["l_mu.weight", "l_var.weight", "l_mu.bias", "l_var.bias"]
        self.heads_adjacent = (all(h in sizes for h in heads) and sizes[heads[0]] == sizes[heads[1]]
                               and sizes[heads[0]] % 4 == 0 and sizes[heads[2]] == sizes[heads[3]] and sizes[heads[2]] % 4 == 0)
        if self.heads_adjacent:
            order = [k for k in order if k not in heads] + heads
        self.offsets, off = {}, 0
        for k in order:
            self.offsets[k] = off
            off += (sizes[k] + 3) // 4 * 4          # keep every view 16-byte aligned
        self.numel = off
        self.data = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.views: Dict[str, torch.Tensor] = {}
        self.grads: Dict[str, torch.Tensor] = {}
        self.bufs: Dict[str, torch.Tensor] = {}
        for k, shape, kind in spec:
            if k in self.offsets:
                o = self.offsets[k]
                self.views[k] = self.data[o:o + sizes[k]].view(shape)
                self.grads[k] = self.grad[o:o + sizes[k]].view(shape)
            elif kind == "rm":
                self.bufs[k] = torch.zeros(shape, dtype=torch.float32, device=self.device)
            elif kind == "rv":
                self.bufs[k] = torch.ones(shape, dtype=torch.float32, device=self.device)
            else:
                self.bufs[k] = torch.zeros(shape, dtype=torch.int64, device=self.device)
        self.opt_state: Dict[str, torch.Tensor] = {}
        self.opt_steps = 0
        self.version = 0      # bumped whenever ``data`` changes -> packed fp16 copies are stale
        self.buf_version = 0  # bumped whenever ``bufs`` are written from outside (load_state_dict)
        self.flush_hooks = [] # callables bringing ``bufs`` up to date before they are read (lazy engine-order shadows)

    # ---- state dict ------------------------------------------------------------------------
    def state_dict(self, prefix: str = "") -> Dict[str, torch.Tensor]:
        for h in self.flush_hooks:
            h()
        out = {}
        for k, _, _ in self.spec:
            out[prefix + k] = (self.views[k] if k in self.views else self.bufs[k]).detach().clone()
        return out

    @torch.no_grad()
    def load_state_dict(self, sd: Dict[str, torch.Tensor], prefix: str = "", strict: bool = True):
        for k, shape, _ in self.spec:
            name = prefix + k
            if name not in sd:
                if strict:
                    raise KeyError(f"missing key {name}")
                continue
            src = sd[name]
            if not torch.is_tensor(src):
                src = torch.as_tensor(src)
            dst = self.views[k] if k in self.views else self.bufs[k]
            if tuple(src.shape) != tuple(dst.shape):
                raise ValueError(f"shape mismatch for {name}: {tuple(src.shape)} vs {tuple(dst.shape)}")
            dst.copy_(src.to(device=self.device, dtype=dst.dtype))
        self.version += 1
        self.buf_version += 1

    def load_recipe(self, rs: np.random.RandomState, perturb: bool = False):
        sd = {k: torch.from_numpy(v) for k, v in recipe_fill(self.spec, rs, perturb).items()}
        self.load_state_dict(sd)

    def drop_pending(self):
        """Forget weight gradients that were queued for fmri_apply_batch (ops.begin_grads) and never applied; their
        persistent buffers go back to the state the next weight-gradient launch expects."""
        for p in getattr(self, "pending", None) or ():
            if getattr(p[0], "_fmri_clear", False):
                p[0].zero_()
            if getattr(p[0], "_fmri_hold", None) is not None:
                p[0]._fmri_hold["busy"] = False
        self.pending = []
        self.materialized = []
        self.defer_grads = False
        self.grad_gate = None

    def zero_grad(self):
        self.drop_pending()
        self.grad.zero_()
        self._cleared = "all"
        self.grads_consumed = False

    def check_grads_readable(self):
        """Raises when the last backward pass over this group ended in the one-launch update (ops.apply_group): its weight
        gradients went from the GEMM layout straight into the optimizer and ``grads`` still holds an older pass."""
        if getattr(self, "grads_consumed", False) or getattr(self, "defer_grads", False):
            raise RuntimeError("the gradients of the last step() were consumed by its fused update and never stored in the "
                               "reference layout: run forward() / gate() / backward() (and apply()) to inspect them")


# ------------------------------------------------------------------------------------------------
# algorithmic FLOP counter: forward MACs x 2 per image, every layer instance counted once
# ------------------------------------------------------------------------------------------------
def _conv_out(n: int, k: int, s: int, p: int) -> int:
    return (n + 2 * p - k) // s + 1


def forward_flops(cfg: ArchConfig, n_voxels: int = 4096) -> Dict[str, float]:
    """Per-image forward FLOPs of the sub-networks: E (Encoder), D (Decoder, one call), S (Discriminator,
    REC+GAN counted as one pass over one image), C (CognitiveEncoder), W (WaeDiscriminator)."""
    k, s, p = cfg.kernel_size, cfg.stride, cfg.padding
    kk = k * k
    h, cin, E = cfg.image_size, 3, 0.0
    for c in cfg.encoder_channels[:3]:
        h = _conv_out(h, k, s, p)
        E += 2.0 * h * h * c * cin * kk
        cin = c
    E += 2.0 * cfg.fc_input ** 2 * cin * cfg.fc_output + 2 * 2.0 * cfg.fc_output * cfg.latent_dim
    size = cfg.encoder_channels[2]
    D = 2.0 * cfg.latent_dim * cfg.fc_input ** 2 * size
    h = cfg.fc_input
    for (ci, co), op in zip([(size, size), (size, cfg.decoder_channels[1]),
                             (cfg.decoder_channels[1], cfg.decoder_channels[2])], cfg.output_pad_dec):
        D += 2.0 * h * h * ci * co * kk                      # transposed conv: MACs = input pixels x Cin x Cout x taps
        h = (h - 1) * s - 2 * p + k + (1 if op else 0)
    D += 2.0 * h * h * cfg.decoder_channels[3] * cfg.decoder_channels[2] * 25
    d = cfg.discrim_channels
    h = _conv_out(cfg.image_size, 5, cfg.stride_gan, 2)
    S = 2.0 * h * h * d[0] * 3 * 25
    cin = d[0]
    for i in (1, 2, 3):
        h = _conv_out(h, k, s, p)
        S += 2.0 * h * h * d[i] * cin * kk
        cin = d[i]
    S += 2.0 * cfg.fc_input_gan ** 2 * cin * cfg.fc_output_gan + 2.0 * cfg.fc_output_gan
    C = 2.0 * n_voxels * 1024 + 2 * 2.0 * 1024 * cfg.latent_dim
    W = 2.0 * (cfg.latent_dim * 512 + 3 * 512 * 512 + 512)
    return dict(E=E, D=D, S=S, C=C, W=W)


def stage1_step_flops(cfg: ArchConfig) -> float:
    """Stage-I step = 3 x (E + 2 D + 3 S) per image (forward + wgrad + dgrad, each layer once)."""
    f = forward_flops(cfg)
    return 3.0 * (f["E"] + 2.0 * f["D"] + 3.0 * f["S"])


def stage2_step_flops(cfg: ArchConfig, n_voxels: int = 4096) -> float:
    """Stage-II step per sample: forward C + E(teacher) + 3 D + 3 S; discriminator weight + data gradient
    2 x 3 S; decoder data gradient on the x_tilde path D (decoder frozen); cognitive encoder weight + data gradient 2 C."""
    f = forward_flops(cfg, n_voxels)
    return (f["C"] + f["E"] + 3.0 * f["D"] + 3.0 * f["S"]) + 6.0 * f["S"] + f["D"] + 2.0 * f["C"]


def stage3_step_flops(cfg: ArchConfig, n_voxels: int = 4096) -> float:
    """Stage-III step per sample (cognitive encoder frozen): forward C + 2 D + 3 S, weight + data gradients of the
    discriminator (2 x 3 S) and of the two decoder calls (2 x 2 D)."""
    f = forward_flops(cfg, n_voxels)
    return f["C"] + 6.0 * f["D"] + 9.0 * f["S"]


def dual1_step_flops(cfg: ArchConfig) -> float:
    """Dual WAE + VAE/GAN Stage-I step per image: the Stage-I step, the third decoder call on the encoder means
    (forward only: it moves BatchNorm statistics, train/wae_vgan_stage1.py:406) and the latent discriminator (D phase on
    2 rows per image forward + weight + data gradient, penalty pass forward + data gradient)."""
    f = forward_flops(cfg)
    return stage1_step_flops(cfg) + f["D"] + (2.0 * 3.0 + 2.0) * f["W"]
