"""Drop-in replacement for the reference's ``models/vae_gan.py`` on top of the MI355X HIP engine.

Same class names, constructor / forward signatures, attribute names, output tuples and ``state_dict()``
keys as the reference (models/vae_gan.py:11-656), so the reference's training and inference scripts import
it unchanged::

    sys.path.insert(0, ".../thesis-fmri-reconstruction_amd")
    import configs.models_config as config
    from models.vae_gan import Encoder, Decoder, Discriminator, CognitiveEncoder, VaeGan, VaeGanCognitive, ...

Each sub-network is an ``nn.Module`` whose parameters/buffers are ordinary (reference-shaped, fp32) tensors;
on a GPU they alias the flat fp32 master buffer of an engine net (fmri_hip.nets) and ``forward`` runs that
net's hand-written HIP kernels through one ``torch.autograd.Function`` per sub-network, so
``loss.backward(retain_graph=True)``, ``requires_grad`` toggling, ``zero_grad``, ``p.grad.data.clamp_``,
torch optimizers, ``.train()/.eval()``, ``.to(device)``, sub-module sharing and ``state_dict`` round-trips
behave as callers of the reference expect.  There is no CPU path: calling a model on CPU tensors raises.

The fast path for training is ``fmri_hip.steps`` (fused one-forward / two-stream-backward steps); this
module is the compatibility surface.
"""
import numpy
import torch
import torch.nn as nn
from torch.autograd import Function

import configs.models_config as config
from fmri_hip import nets as _nets
from fmri_hip import ops as _ops
from fmri_hip.params import ArchConfig


def _arch() -> ArchConfig:
    return ArchConfig.from_module(config)


def _unit_scale(t: torch.Tensor) -> torch.Tensor:
    """Device-side power-of-two factor (fp32 scalar tensor) bringing a cotangent to ~unit RMS * 16 before the fp16
    backward -- computed and applied on the device: no host synchronisation inside ``backward()``."""
    r = t.float().pow(2).mean().sqrt()
    f = torch.exp2(torch.round(torch.log2(16.0 / r)))
    return torch.where(torch.isfinite(f) & (f > 0), f, torch.ones_like(f))


class _EngineBacked(nn.Module):
    """nn.Module whose parameters live in (alias) an engine net's flat buffers."""

    def _make_net(self, device):
        raise NotImplementedError

    def _tensors(self):
        """[(state-dict key, tensor)] of this module, cached: ``state_dict()`` walks the module tree and builds an
        OrderedDict with hooks on every call (~0.3 ms for a sub-network, twice per forward before this cache).  The cache
        is keyed on the identity of the registered Parameter / buffer objects, which is cheap to re-check."""
        ids = tuple(map(id, self.parameters())) + tuple(map(id, self.buffers()))
        c = self.__dict__.get("_tcache")
        if c is None or c[0] != ids:
            c = (ids, list(self.state_dict(keep_vars=True).items()))
            self.__dict__["_tcache"] = c
        return c[1]

    def _engine(self):
        items = self._tensors()
        dev = items[0][1].device
        if dev.type != "cuda":
            raise RuntimeError("models.vae_gan (HIP engine): parameters/inputs must live on an MI355X device; "
                               "there is no CPU fallback")
        net = self.__dict__.get("_net")
        bound = net is not None and net.group.device == dev
        if bound:
            g = net.group
            for k, t in items:
                tgt = g.views[k] if k in g.views else g.bufs[k]
                if t.data_ptr() != tgt.data_ptr():
                    bound = False
                    break
        if not bound:
            net = self._make_net(dev)
            g = net.group
            g.load_state_dict({k: v.detach() for k, v in items})
            for k, t in items:
                t.data = g.views[k] if k in g.views else g.bufs[k]
            self.__dict__["_net"] = net
            self.__dict__["_pver"] = -1
            self.__dict__["_plist"] = None
        ver = 0
        for _, t in items:
            ver += t._version
        if ver != self.__dict__["_pver"]:
            net.group.version += 1
            self.__dict__["_pver"] = ver
        for bn in net.all_bns():
            bn.eval_mode = not self.training
        return net

    def _param_list(self, net):
        pl = self.__dict__.get("_plist")
        if pl is None or pl[0] is not self.__dict__.get("_tcache"):
            sd = dict(self._tensors())
            pl = (self.__dict__["_tcache"], [sd[k] for k in net.group.pkeys])
            self.__dict__["_plist"] = pl
        return pl[1]

    def _engine_params_changed(self):
        """Call after the engine itself modified parameters (fused steps)."""
        self.__dict__["_pver"] = -1


def _grad_views(net, params, flat):
    """Per-parameter views of a flat gradient tensor laid out like the sub-network's FlatGroup (None where the
    parameter does not require grad)."""
    g = net.group
    out = []
    for k, p in zip(g.pkeys, params):
        if p.requires_grad:
            o = g.offsets[k]
            out.append(flat[o:o + p.numel()].view(p.shape))
        else:
            out.append(None)
    return out


def _collect_grads(net, params, f=None):
    """Parameter gradients of the engine's last backward; ``f``: device scalar the cotangent was multiplied by.  ONE
    device op over the sub-network's flat gradient buffer; the per-parameter gradients are views of the result."""
    g = net.group
    flat = g.grad.clone() if f is None else g.grad * (1.0 / f)
    return _grad_views(net, params, flat)


def _check_train(net, what):
    if any(bn.eval_mode for bn in net.all_bns()):
        raise RuntimeError(f"{what}: backward through an eval-mode forward is not supported")


# ------------------------------------------------------------------------------------------------
# autograd bridges (one per sub-network)
# ------------------------------------------------------------------------------------------------
class _EncoderFn(Function):
    @staticmethod
    def forward(ctx, mod, x, *params):
        net = mod._engine()
        x16 = _ops.images_to_nhwc(x) if x.dim() == 4 else _ops.rows_to_f16(x)
        head32, ectx = net.forward(x16)
        ctx.net, ctx.ectx, ctx.params = net, ectx, params
        z = head32.shape[1] // 2
        return head32[:, :z].clone(), head32[:, z:].clone()

    @staticmethod
    def backward(ctx, dmu, dlv):
        net = ctx.net
        _check_train(net, "Encoder")
        if not any(p.requires_grad for p in ctx.params):
            return (None, None) + (None,) * len(ctx.params)
        dhead = torch.cat([dmu, dlv], 1).float().contiguous()
        f = _unit_scale(dhead)
        net.group.zero_grad()
        net.backward(ctx.ectx, (dhead * f).half(), 1.0)
        return (None, None, *_collect_grads(net, ctx.params, f))


class _DecoderFn(Function):
    @staticmethod
    def forward(ctx, mod, z, *params):
        net = mod._engine()
        # (a sampled latent can leave fp16's range -- sigma = exp(0.5 logvar) -- so its rows are stored range-scaled)
        z16, zs = _ops.rows_to_f16_ranged(z)
        y16, dctx = net.forward(z16, 1, zscale=zs)
        ctx.net, ctx.dctx, ctx.params, ctx.zdim = net, dctx, params, z.shape[1]
        ctx.need_dz = z.requires_grad
        return _ops.nhwc_to_images(y16, net.c3.cout)

    @staticmethod
    def backward(ctx, dimg):
        net = ctx.net
        _check_train(net, "Decoder")
        train = any(p.requires_grad for p in ctx.params)
        if not train and not ctx.need_dz:
            return (None, None) + (None,) * len(ctx.params)
        f = _unit_scale(dimg)
        cot = _ops.images_to_nhwc((dimg * f).contiguous())
        net.group.zero_grad()
        res = net.backward(ctx.dctx, cot, [dict(g=0, scale=1.0, train=train, need_dz=ctx.need_dz)])
        dz = res.get(0) / f if ctx.need_dz else None
        return (None, dz, *_collect_grads(net, ctx.params, f))


# FMRI_API_REUSE=off: every call recomputes (A/B timing of the two reuse paths below; results are identical)
import os as _os
import weakref
_REUSE = _os.environ.get("FMRI_API_REUSE") != "off"


def _same_direction(new, old):
    """(r, ok): ``new == r * old`` up to fp32 rounding?  The three ``backward(retain_graph=True)`` calls of the
    reference loop send the SAME cotangent through the discriminator up to a scalar (d mse, then lambda * d mse;
    -(1 - lambda) * d L_dis, then d L_dis: train_vgan_stage1.py:369-372, :410-432) and every backward operator is linear
    in it, so the second traversal can be the first one's result times r.  One host synchronisation (the scripts'
    equilibrium gate already reads two scalars per step)."""
    if new.shape != old.shape:
        return 0.0, False
    a, b = new.reshape(-1), old.reshape(-1)
    bb = torch.dot(b, b)
    r = torch.dot(a, b) / bb
    err = (a - r * b).pow(2).sum()
    aa = torch.dot(a, a)
    rr, ee, na, nb = torch.stack([r, err, aa, bb]).tolist()
    ok = nb > 0.0 and na > 0.0 and ee <= 1e-10 * na and rr == rr
    return rr, ok


class _DiscriminatorFn(Function):
    @staticmethod
    def forward(ctx, mod, mode, xo, xp, xs, *params):
        net = mod._engine()
        B, _, H, W = xo.shape
        ctx.net, ctx.params, ctx.mode, ctx.B = net, params, mode, B
        ctx.needs = (xo.requires_grad, xp.requires_grad, xs.requires_grad)
        ctx.cache = None
        ctx.gver = net.group.version
        # the conv stack of a 'REC' call is kept for a 'GAN' call on the SAME three tensors (models/vae_gan.py:284-285 calls
        # the discriminator twice on identical inputs): the second call only runs the fc head and applies the conv
        # blocks' second running-statistics update.  The memo holds the input tensors, so their storage cannot be
        # re-used for other data in between; any in-place write changes ``_version`` and invalidates it.
        memo = mod.__dict__.get("_memo")
        key = (id(xo), xo._version, id(xp), xp._version, id(xs), xs._version, net.group.version, mod.training)
        # the memo holds WEAK references to the three inputs (a strong one to ``xp`` would keep the whole autograd graph
        # behind it -- encoder, decoder and their saved activations -- alive until the next discriminator call): a dead
        # reference, i.e. an ``id`` that may have been re-used, never matches
        if (_REUSE and mode == "GAN" and net.level == 3 and memo is not None and memo["key"] == key
                and memo["xs"][0]() is xo and memo["xs"][1]() is xp and memo["xs"][2]() is xs):
            sctx = dict(memo["sctx"])                 # own dict: the head's entries belong to this call
            mod.__dict__["_memo"] = None
            if mod.training:
                net.conv_running_again(sctx, 1)
            logit32 = net.forward_head(sctx, True, 1)
        else:
            d = torch.empty(3 * B, H, W, 8, dtype=torch.float16, device=xo.device)
            for i, t in enumerate((xo, xp, xs)):
                _ops.images_to_nhwc(t, out=d[i * B:(i + 1) * B])
            if mode == "REC":
                feat, _, sctx = net.forward(d, conv_updates=1, head=False)
                ctx.sctx = sctx
                # (recon_level < 3: the REC pass stopped below the last block, a GAN call starts over)
                mod.__dict__["_memo"] = (dict(key=key, xs=tuple(weakref.ref(t) for t in (xo, xp, xs)), sctx=sctx)
                                         if net.level == 3 else None)
                return _ops.nhwc_to_images(feat, feat.shape[-1]).reshape(3 * B, -1)
            mod.__dict__["_memo"] = None
            _, logit32, sctx = net.forward(d, conv_updates=1, fc_updates=1)
        ctx.sctx = sctx
        prob = torch.sigmoid(logit32)
        ctx.prob = prob
        return prob

    @staticmethod
    def backward(ctx, dout):
        net, B = ctx.net, ctx.B
        _check_train(net, "Discriminator")
        train = any(p.requires_grad for p in ctx.params)
        want_img = any(ctx.needs)
        none = (None,) * (5 + len(ctx.params))
        if not train and not want_img:
            return none
        # a second traversal with a scalar multiple of the cotangent of the first one (same weights, same
        # requires_grad pattern): its results times that scalar
        # (the live ``_version`` sum of the parameters: an optimizer step BETWEEN two backward calls on the same forward
        # changes the weights without passing through ``_engine()``, which is where ``group.version`` is refreshed)
        sig = (train, ctx.needs, tuple(p.requires_grad for p in ctx.params), net.group.version,
               sum(p._version for p in ctx.params))
        c = ctx.cache
        if _REUSE and c is not None and c["sig"] == sig:
            r, ok = _same_direction(dout, c["dout"])
            if ok:
                gi = [None, None, None]
                if c["full"] is not None:
                    full = c["full"] * r
                    for i in range(3):
                        if ctx.needs[i]:
                            gi[i] = full[i * B:(i + 1) * B]
                return (None, None, gi[0], gi[1], gi[2], *_grad_views(net, ctx.params, c["flat"] * r))
        rows = slice(0, 3 * B) if want_img else None
        net.group.zero_grad()
        if ctx.mode == "REC":
            feat = ctx.sctx["raws"][net.level - 1]
            n3, h, w, c_ = feat.shape
            f = _unit_scale(dout)
            dfeat16 = _ops.images_to_nhwc((dout * f).reshape(n3, c_, h, w).contiguous())
            _, dimg = net.backward(ctx.sctx, None, 1.0, dfeat16, 1.0, False, rows, img_streams=(False, True),
                                   train_b=train)
        else:
            dlogit = (dout * ctx.prob * (1.0 - ctx.prob)).float()
            f = _unit_scale(dlogit)
            dl16 = torch.zeros(3 * B, 8, dtype=torch.float16, device=dout.device)
            dl16[:, :1] = (dlogit * f).half()
            dimg, _ = net.backward(ctx.sctx, dl16, 1.0, None, 1.0, train, rows)
        gi = [None, None, None]
        full = None
        if want_img:
            full = _ops.nhwc_to_images(dimg, 3, 1.0) / f
            for i in range(3):
                if ctx.needs[i]:
                    gi[i] = full[i * B:(i + 1) * B]
        flat = net.group.grad * (1.0 / f)
        # private copies: autograd may adopt the returned tensors as ``.grad`` and the scripts modify those in place
        ctx.cache = dict(sig=sig, dout=dout.detach().clone(), flat=flat.clone(),
                         full=None if full is None else full.clone())
        return (None, None, gi[0], gi[1], gi[2], *_grad_views(net, ctx.params, flat))


class _WaeDiscriminatorFn(Function):
    @staticmethod
    def forward(ctx, mod, z, *params):
        net = mod._engine()
        logit32, wctx = net.forward(_ops.rows_to_f16(z))
        prob = torch.sigmoid(logit32)
        ctx.net, ctx.wctx, ctx.params, ctx.prob, ctx.need_dz = net, wctx, params, prob, z.requires_grad
        return prob

    @staticmethod
    def backward(ctx, dout):
        net = ctx.net
        train = any(p.requires_grad for p in ctx.params)
        if not train and not ctx.need_dz:
            return (None, None) + (None,) * len(ctx.params)
        dlogit = (dout * ctx.prob * (1.0 - ctx.prob)).float()
        f = _unit_scale(dlogit)
        dl16 = torch.zeros(dlogit.shape[0], 8, dtype=torch.float16, device=dout.device)
        dl16[:, :1] = (dlogit * f).half()
        net.group.zero_grad()
        dz = net.backward(ctx.wctx, dl16, 1.0, train, ctx.need_dz)
        return (None, dz / f if dz is not None else None, *_collect_grads(net, ctx.params, f))


# ------------------------------------------------------------------------------------------------
# parameter holders mirroring the reference module tree (so state_dict keys are identical)
# ------------------------------------------------------------------------------------------------
class _BlockNet:
    """Engine objects of ONE conv / deconv + BatchNorm + ReLU block (EncoderBlock / DecoderBlock used on their own)."""

    def __init__(self, kind, cin, cout, out_pad, device):
        from fmri_hip.params import FlatGroup, _bn
        shape = (cout, cin, config.kernel_size, config.kernel_size) if kind == "conv" else \
            (cin, cout, config.kernel_size, config.kernel_size)
        self.group = FlatGroup([("conv.weight", shape, "w")] + _bn("bn.", cout), device)
        self.conv = _ops.ConvLayer(self.group, "conv.weight", None, kind, cin, cout, config.kernel_size, config.stride,
                                   config.padding, out_pad)
        self.bn = _ops.BatchNorm(self.group, "bn.", cout)
        self.cout = cout

    def all_bns(self):
        return [self.bn]

    def forward(self, x):
        raw = self.conv.forward(_ops.images_to_nhwc(x), bn_groups=0 if self.bn.eval_mode else 1)
        act, _ = self.bn.forward(raw, relu=True, updates=1, stat_acc=self.conv.take_stats())
        return _ops.nhwc_to_images(act, self.cout), raw


class _Block(_EngineBacked):
    """A block called on its own runs its forward on the engine (train-mode batch statistics, running statistics
    updated, eval mode honoured) but is not differentiable: training goes through the parent networks (Encoder /
    Decoder / Discriminator), whose autograd bridges cover whole sub-networks."""

    def _forward(self, ten):
        if torch.is_grad_enabled() and ten.requires_grad:
            # a caller differentiating THROUGH a lone block would silently get no gradient: refuse instead
            raise RuntimeError(f"{type(self).__name__} called on its own is forward-only on the HIP engine (no autograd "
                               "graph is recorded); differentiate through Encoder / Decoder / Discriminator, or wrap "
                               "the call in torch.no_grad() / detach the input")
        act, raw = self._engine().forward(ten)
        self._engine_params_changed()
        return act, raw


class EncoderBlock(_Block):
    """conv(k5,s2,p2,no bias) + BN(momentum .9) + ReLU (reference models/vae_gan.py:11-35)."""

    def __init__(self, channel_in, channel_out):
        super(EncoderBlock, self).__init__()
        self.conv = nn.Conv2d(in_channels=channel_in, out_channels=channel_out, kernel_size=config.kernel_size,
                              padding=config.padding, stride=config.stride, bias=False)
        self.bn = nn.BatchNorm2d(num_features=channel_out, momentum=0.9)

    def _make_net(self, device):
        return _BlockNet("conv", self.conv.in_channels, self.conv.out_channels, 0, device)

    def forward(self, ten, out=False, t=False):
        """``out=True`` also returns the raw convolution output (models/vae_gan.py:23-30)."""
        act, raw = self._forward(ten)
        if out:
            return act, _ops.nhwc_to_images(raw, self.conv.out_channels)
        return act


class DecoderBlock(_Block):
    """deconv(k5,s2,p2,output_padding) + BN + ReLU (reference models/vae_gan.py:38-60)."""

    def __init__(self, channel_in, channel_out, out=False):
        super(DecoderBlock, self).__init__()
        self.conv = nn.ConvTranspose2d(channel_in, channel_out, kernel_size=config.kernel_size,
                                       padding=config.padding, stride=config.stride,
                                       output_padding=1 if out else 0, bias=False)
        self.bn = nn.BatchNorm2d(channel_out, momentum=0.9)

    def _make_net(self, device):
        return _BlockNet("deconv", self.conv.in_channels, self.conv.out_channels, self.conv.output_padding[0], device)

    def forward(self, ten):
        return self._forward(ten)[0]


class Encoder(_EngineBacked):
    """Visual encoder (reference models/vae_gan.py:63-96)."""

    def __init__(self, channel_in=3, z_size=128):
        super(Encoder, self).__init__()
        self.__dict__["_cfg"] = _arch()
        self.__dict__["_z"], self.__dict__["_cin"] = z_size, channel_in
        self.size = channel_in
        layers_list = []
        for i in range(3):
            layers_list.append(EncoderBlock(channel_in=self.size, channel_out=config.encoder_channels[i]))
            self.size = config.encoder_channels[i]
        self.conv = nn.Sequential(*layers_list)
        self.fc = nn.Sequential(nn.Linear(in_features=config.fc_input * config.fc_input * self.size,
                                          out_features=config.fc_output, bias=False),
                                nn.BatchNorm1d(num_features=config.fc_output, momentum=0.9),
                                nn.ReLU(True))
        self.l_mu = nn.Linear(in_features=config.fc_output, out_features=z_size)
        self.l_var = nn.Linear(in_features=config.fc_output, out_features=z_size)

    def _make_net(self, device):
        cfg = self._cfg
        if cfg.latent_dim != self._z:
            cfg = ArchConfig(**{**cfg.__dict__, "latent_dim": self._z})
        return _nets.EncoderNet(cfg, device, self._cin)

    def forward(self, ten):
        net = self._engine()
        return _EncoderFn.apply(self, ten, *self._param_list(net))


class Decoder(_EngineBacked):
    """Decoder (reference models/vae_gan.py:99-132)."""

    def __init__(self, z_size, size):
        super(Decoder, self).__init__()
        self.__dict__["_cfg"] = _arch()
        self.__dict__["_z"], self.__dict__["_size0"] = z_size, size
        self.fc = nn.Sequential(nn.Linear(in_features=z_size, out_features=config.fc_input * config.fc_input * size,
                                          bias=False),
                                nn.BatchNorm1d(num_features=config.fc_input * config.fc_input * size, momentum=0.9),
                                nn.ReLU(True))
        self.size = size
        layers_list = []
        layers_list.append(DecoderBlock(channel_in=self.size, channel_out=self.size, out=config.output_pad_dec[0]))
        layers_list.append(DecoderBlock(channel_in=self.size, channel_out=config.decoder_channels[1],
                                        out=config.output_pad_dec[1]))
        self.size = config.decoder_channels[1]
        layers_list.append(DecoderBlock(channel_in=self.size, channel_out=config.decoder_channels[2],
                                        out=config.output_pad_dec[2]))
        self.size = config.decoder_channels[2]
        layers_list.append(nn.Sequential(
            nn.Conv2d(in_channels=self.size, out_channels=config.decoder_channels[3], kernel_size=5, stride=1,
                      padding=2),
            nn.Tanh()))
        self.conv = nn.Sequential(*layers_list)

    def _make_net(self, device):
        cfg = self._cfg
        if cfg.latent_dim != self._z:
            cfg = ArchConfig(**{**cfg.__dict__, "latent_dim": self._z})
        return _nets.DecoderNet(cfg, device, self._size0)

    def forward(self, ten):
        net = self._engine()
        return _DecoderFn.apply(self, ten, *self._param_list(net))


class Discriminator(_EngineBacked):
    """Image discriminator with the 'REC' / 'GAN' modes (reference models/vae_gan.py:135-187)."""

    def __init__(self, channel_in=3, recon_level=3):
        super(Discriminator, self).__init__()
        self.__dict__["_cfg"] = _arch()
        self.size = channel_in
        self.recon_levl = recon_level
        self.conv = nn.ModuleList()
        self.conv.append(nn.Sequential(
            nn.Conv2d(in_channels=3, out_channels=config.discrim_channels[0], kernel_size=5, stride=config.stride_gan,
                      padding=2),
            nn.ReLU(inplace=True)))
        self.size = config.discrim_channels[0]
        for i in (1, 2, 3):
            self.conv.append(EncoderBlock(channel_in=self.size, channel_out=config.discrim_channels[i]))
            self.size = config.discrim_channels[i]
        self.fc = nn.Sequential(
            nn.Linear(in_features=config.fc_input_gan * config.fc_input_gan * self.size,
                      out_features=config.fc_output_gan, bias=False),
            nn.BatchNorm1d(num_features=config.fc_output_gan, momentum=0.9),
            nn.ReLU(inplace=True),
            nn.Linear(in_features=config.fc_output_gan, out_features=1),
        )

    def _make_net(self, device):
        return _nets.DiscriminatorNet(self._cfg, device, self.recon_levl)

    def forward(self, ten_orig, ten_predicted, ten_sampled, mode='REC'):
        net = self._engine()
        return _DiscriminatorFn.apply(self, "REC" if mode == "REC" else "GAN", ten_orig, ten_predicted, ten_sampled,
                                      *self._param_list(net))


class CognitiveEncoder(_EngineBacked):
    """fMRI -> latent encoder (reference models/vae_gan.py:190-232)."""

    def __init__(self, input_size, z_size=128, channel_in=3):
        super(CognitiveEncoder, self).__init__()
        self.__dict__["_cfg"] = _arch()
        self.__dict__["_z"], self.__dict__["_v"] = z_size, input_size
        self.size = channel_in
        self.fc1 = nn.Sequential(nn.Linear(in_features=input_size, out_features=1024, bias=False),
                                 nn.BatchNorm1d(num_features=1024, momentum=0.9),
                                 nn.ReLU(True))
        self.l_mu = nn.Linear(in_features=1024, out_features=z_size)
        self.l_var = nn.Linear(in_features=1024, out_features=z_size)

    def _make_net(self, device):
        cfg = self._cfg
        if cfg.latent_dim != self._z:
            cfg = ArchConfig(**{**cfg.__dict__, "latent_dim": self._z})
        return _nets.CognitiveEncoderNet(cfg, self._v, device)

    def forward(self, ten):
        net = self._engine()
        return _EncoderFn.apply(self, ten, *self._param_list(net))


class WaeDiscriminator(_EngineBacked):
    """Latent-space discriminator MLP (reference models/vae_gan.py:499-529)."""

    def __init__(self, z_size=128, dim_h=512):
        super(WaeDiscriminator, self).__init__()
        self.__dict__["_cfg"] = _arch()
        self.n_z = z_size
        self.dim_h = dim_h
        self.main = nn.Sequential(
            nn.Linear(self.n_z, self.dim_h), nn.ReLU(True),
            nn.Linear(self.dim_h, self.dim_h), nn.ReLU(True),
            nn.Linear(self.dim_h, self.dim_h), nn.ReLU(True),
            nn.Linear(self.dim_h, self.dim_h), nn.ReLU(True),
            nn.Linear(self.dim_h, 1), nn.Sigmoid())
        for m in self.modules():
            if isinstance(m, nn.Linear):
                m.weight.data.normal_(0.0, 0.0099999)
                m.bias.data.zero_()

    def _make_net(self, device):
        cfg = self._cfg
        if cfg.latent_dim != self.n_z:
            cfg = ArchConfig(**{**cfg.__dict__, "latent_dim": self.n_z})
        return _nets.WaeDiscriminatorNet(cfg, device, self.dim_h)

    def forward(self, x):
        net = self._engine()
        return _WaeDiscriminatorFn.apply(self, x, *self._param_list(net))


def _init_parameters(model):
    """VaeGan.init_parameters (reference models/vae_gan.py:252-264)."""
    for m in model.modules():
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d, nn.Linear)):
            if hasattr(m, "weight") and m.weight is not None and m.weight.requires_grad:
                scale = 1.0 / numpy.sqrt(numpy.prod(m.weight.shape[1:]))
                scale /= numpy.sqrt(3)
                nn.init.uniform_(m.weight, -scale, scale)
            if hasattr(m, "bias") and m.bias is not None and m.bias.requires_grad:
                nn.init.constant_(m.bias, 0.0)


def _reparameterize(mu, logvar):
    """reference models/vae_gan.py:266-269 (one normal_() draw shaped like logvar)."""
    std = logvar.mul(0.5).exp_()
    eps = std.data.new(std.size()).normal_()
    return eps.mul(std).add_(mu)


def _gan_loss(x, x_tilde, disc_layer_original, disc_layer_predicted, disc_layer_sampled, disc_class_original,
              disc_class_predicted, disc_class_sampled, mus, variances):
    """VaeGan.loss / VaeGanCognitive.loss (reference models/vae_gan.py:302-320, :411-432)."""
    nle = 0.5 * (x.view(len(x), -1) - x_tilde.view(len(x_tilde), -1)) ** 2
    kl = -0.5 * torch.sum(-variances.exp() - torch.pow(mus, 2) + variances + 1, 1)
    mse = torch.sum(0.5 * (disc_layer_original - disc_layer_predicted) ** 2, 1)
    bce_dis_original = -torch.log(disc_class_original + 1e-3)
    bce_dis_predicted = -torch.log(1 - disc_class_predicted + 1e-3)
    bce_dis_sampled = -torch.log(1 - disc_class_sampled + 1e-3)
    return nle, kl, mse, bce_dis_original, bce_dis_predicted, bce_dis_sampled


class VaeGan(nn.Module):
    """Stage-I VAE/GAN wrapper (reference models/vae_gan.py:235-320)."""

    def __init__(self, device, z_size=128, recon_level=3):
        super(VaeGan, self).__init__()
        self.z_size = z_size
        self.encoder = Encoder(z_size=self.z_size).to(device)
        self.decoder = Decoder(z_size=self.z_size, size=self.encoder.size).to(device)
        self.discriminator = Discriminator(channel_in=3, recon_level=recon_level).to(device)
        self.init_parameters()
        self.device = device

    def init_parameters(self):
        _init_parameters(self)

    def reparameterize(self, mu, logvar):
        return _reparameterize(mu, logvar)

    def forward(self, x, gen_size=10):
        if x is not None:
            x = x.to(self.device)
        if self.training:
            mus, log_variances = self.encoder(x)
            z = self.reparameterize(mus, log_variances)
            x_tilde = self.decoder(z)
            z_p = torch.randn(len(x), self.z_size).to(self.device).requires_grad_(True)
            x_p = self.decoder(z_p)
            disc_layer = self.discriminator(x, x_tilde, x_p, "REC")
            disc_class = self.discriminator(x, x_tilde, x_p, "GAN")
            return x_tilde, disc_class, disc_layer, mus, log_variances
        if x is None:
            z_p = torch.randn(gen_size, self.z_size).to(self.device)
            return self.decoder(z_p)
        mus, log_variances = self.encoder(x)
        z = self.reparameterize(mus, log_variances)
        return self.decoder(z)

    loss = staticmethod(_gan_loss)


class VaeGanCognitive(nn.Module):
    """Dual-VAE/GAN wrapper for Stage II / III (reference models/vae_gan.py:323-432)."""

    def __init__(self, device, encoder, decoder, discriminator, z_size=128, recon_level=3, teacher_net=None, stage=1,
                 mode='vae'):
        super(VaeGanCognitive, self).__init__()
        self.device = device
        self.z_size = z_size
        self.encoder = encoder
        self.decoder = decoder
        self.discriminator = discriminator
        self.teacher_net = teacher_net
        self.stage = stage
        self.mode = mode

    def reparameterize(self, mu, logvar):
        return _reparameterize(mu, logvar)

    def forward(self, sample, gen_size=10):
        if sample is None:
            z_p = torch.randn(gen_size, self.z_size).to(self.device)
            return self.decoder(z_p)
        x = sample['fmri'].to(self.device)
        gt_x = sample['image'].to(self.device)
        if not self.training:
            mus, log_variances = self.encoder(x)
            z = self.reparameterize(mus, log_variances)
            return self.decoder(z)
        if self.mode == 'vae':
            mus, log_variances = self.encoder(x)
            z = self.reparameterize(mus, log_variances)
            x_tilde = self.decoder(z)
            if self.teacher_net is not None and self.stage == 2:
                for param in self.teacher_net.encoder.parameters():
                    param.requires_grad = False
                mu_teacher, logvar_teacher = self.teacher_net.encoder(gt_x)
                z_teacher = self.reparameterize(mu_teacher, logvar_teacher)
                gt_x = self.decoder(z_teacher)
        elif self.mode == 'wae':
            mus, log_variances = self.encoder(x)
            x_tilde = self.decoder(mus)
            mu_teacher, logvar_teacher = self.teacher_net.encoder(gt_x)
            gt_x = self.decoder(mu_teacher)
        z_p = torch.randn(len(x), self.z_size).to(self.device).requires_grad_(True)
        x_p = self.decoder(z_p)
        disc_layer = self.discriminator(gt_x, x_tilde, x_p, "REC")
        disc_class = self.discriminator(gt_x, x_tilde, x_p, "GAN")
        return gt_x, x_tilde, disc_class, disc_layer, mus, log_variances

    loss = staticmethod(_gan_loss)


class WaeGan(nn.Module):
    """WAE with GAN-based latent penalty: container for encoder / decoder / latent discriminator
    (reference models/vae_gan.py:435-496).  The training scripts call the sub-modules directly; the
    reference's train-mode ``forward`` is dead code that raises and does so here too."""

    def __init__(self, device, z_size=128):
        super(WaeGan, self).__init__()
        self.z_size = z_size
        self.encoder = Encoder(z_size=self.z_size).to(device)
        self.decoder = Decoder(z_size=self.z_size, size=self.encoder.size).to(device)
        self.discriminator = WaeDiscriminator(z_size=self.z_size).to(device)
        self.init_parameters()
        self.device = device

    def init_parameters(self):
        _init_parameters(self)

    def forward(self, x, gen_size=10):
        if x is not None:
            x = x.to(self.device)
        if self.training:
            raise TypeError("WaeGan.forward in train mode is unused in the reference (it calls the 1-argument "
                            "WaeDiscriminator.forward with 3 arguments); call encoder/decoder/discriminator directly")
        if x is None:
            raise TypeError("WaeGan.forward(None) is not supported by the reference either (randn_like(None))")
        mus, log_variances = self.encoder(x)
        return self.decoder(mus)


class WaeGanCognitive(nn.Module):
    """WAE/GAN container for Stage II / III (reference models/vae_gan.py:532-578)."""

    def __init__(self, device, encoder, decoder, z_size=128, recon_level=3):
        super(WaeGanCognitive, self).__init__()
        self.z_size = z_size
        self.encoder = encoder
        self.discriminator = WaeDiscriminator(z_size=self.z_size).to(device)
        self.device = device
        self.decoder = decoder
        for param in self.decoder.parameters():
            param.requires_grad = False

    def reparameterize(self, mu, logvar):
        return _reparameterize(mu, logvar)

    def forward(self, x, gen_size=10):
        if x is not None:
            x = x.to(self.device)
        if self.training:
            raise TypeError("WaeGanCognitive.forward in train mode is unused/broken in the reference; call the "
                            "sub-modules directly as train_wae_stage2.py does")
        mus, log_variances = self.encoder(x)
        return self.decoder(mus)


class DCGan(nn.Module):
    """Plain DCGAN wiring (reference models/vae_gan.py:581-622)."""

    def __init__(self, device, decoder, discriminator, z_size=128, recon_level=3):
        super(DCGan, self).__init__()
        self.device = device
        self.z_size = z_size
        self.decoder = decoder
        self.discriminator = discriminator

    def forward(self, sample, gen_size=10):
        if sample is None:
            return self.decoder(torch.randn(gen_size, self.z_size).to(self.device))
        gt_x = sample.to(self.device)
        if self.training:
            z_p = torch.randn(len(gt_x), self.z_size).to(self.device).requires_grad_(True)
            x_tilde = self.decoder(z_p)
            disc_layer = self.discriminator(gt_x, x_tilde, x_tilde, "REC")
            disc_class = self.discriminator(gt_x, x_tilde, x_tilde, "GAN")
            return gt_x, x_tilde, disc_class, disc_layer
        return self.decoder(torch.randn(gt_x.shape[0], self.z_size).to(self.device))


# names used in the project brief
VisualEncoder = Encoder
CognitiveVaeGan = VaeGanCognitive
