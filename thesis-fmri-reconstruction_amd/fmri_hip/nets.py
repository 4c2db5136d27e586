"""Sub-networks of the VAE/GAN as explicit forward / backward passes over the HIP operators.

Mirrors (reference models/vae_gan.py): Encoder :63-96, Decoder :99-132, Discriminator :135-187,
CognitiveEncoder :190-232, WaeDiscriminator :499-529.  Parameter names/shapes are the reference
state-dict entries; activations are fp16 NHWC, statistics/losses/master weights fp32.

Backward passes are written by hand (no autograd inside the engine): each returns/accumulates exactly
the gradients the training-step bodies consume, and supports several cotangent "streams" through one
saved forward pass (the two-stream trick of SURVEY 8a-14).
"""
from __future__ import annotations

from typing import List, Optional

import os

import torch

from . import lib
from .ops import (ACT_NONE, ACT_RELU, ACT_TANH, BatchNorm, ConvLayer, DenseLayer, act_backward, join_side, pad8,
                  repack_group)
from .params import (ArchConfig, FlatGroup, cognitive_encoder_spec, decoder_spec, discriminator_spec, encoder_spec,
                     wae_discriminator_spec)

_P = lib.ptr


class FusedHeads:
    """l_mu | l_var as one N = 2z GEMM (models/vae_gan.py:84-85,91-92).  The group lays the two heads out side by side
    (FlatGroup.heads_adjacent), so the concatenated weight / bias and their gradients are views of the flat buffers: no
    copies, the weight gradient accumulates in place, the fp16 copies are part of the group's batched repack.  (Groups
    with another layout fall back to concatenated scratch copies refreshed when the masters change.)"""

    def __init__(self, group: FlatGroup, k_in: int, z: int):
        self.group, self.z, self.k_in = group, z, k_in
        dev = group.device
        self.direct = bool(getattr(group, "heads_adjacent", False))
        if self.direct:
            ow, ob = group.offsets["l_mu.weight"], group.offsets["l_mu.bias"]
            assert group.offsets["l_var.weight"] == ow + z * k_in and group.offsets["l_var.bias"] == ob + z
            self.wcat = group.data[ow:ow + 2 * z * k_in].view(2 * z, k_in)
            self.gw = group.grad[ow:ow + 2 * z * k_in].view(2 * z, k_in)
            self.bcat = group.data[ob:ob + 2 * z]
            self.gb = group.grad[ob:ob + 2 * z]
            self.dense = DenseLayer(group, (self.wcat, self.gw), (self.bcat, self.gb), k_in, 2 * z)
            return
        self.wcat = torch.empty(2 * z, k_in, dtype=torch.float32, device=dev)
        self.bcat = torch.empty(2 * z, dtype=torch.float32, device=dev)
        self.gw = torch.zeros(2 * z, k_in, dtype=torch.float32, device=dev)
        self._vg = _Versioned(group)
        self.dense = DenseLayer(self._vg, (self.wcat, self.gw), (self.bcat, None), k_in, 2 * z)
        self._v = -1

    def refresh(self):
        """Bring the concatenated scratch weights and their fp16 GEMM copies up to date now (instead of lazily at the
        next forward)."""
        if self.direct:
            return                                   # views of the masters; packed with the group
        self._sync()
        repack_group(self._vg)

    def _sync(self):
        if self.direct:
            return
        if self._v != self.group.version:
            v = self.group.views
            z = self.z
            self.wcat[:z].copy_(v["l_mu.weight"])
            self.wcat[z:].copy_(v["l_var.weight"])
            self.bcat[:z].copy_(v["l_mu.bias"])
            self.bcat[z:].copy_(v["l_var.bias"])
            self._v = self.group.version

    def forward(self, h16):
        self._sync()
        _, head32 = self.dense.forward(h16, ACT_NONE, want16=False, want32=True)
        return head32                                   # [B, 2z] fp32 (mu | logvar), bias added

    def backward(self, h16, dhead16, scale, need_dgrad=True):
        if self.direct:
            self.dense.wgrad(h16, dhead16, scale)        # side stream, accumulates into the flat gradient buffer
            self.dense.bias_grad(dhead16, scale)
            if need_dgrad:
                dh, _ = self.dense.dgrad(dhead16)
                return dh
            return None
        self._sync()
        g = self.group.grads
        z = self.z
        self.gw.zero_()
        self.dense._wgrad(h16, dhead16, scale)      # current stream: gw is read right below
        g["l_mu.weight"].add_(self.gw[:z])
        g["l_var.weight"].add_(self.gw[z:])
        bs = dhead16.float().sum(0) * (1.0 / scale)
        g["l_mu.bias"].add_(bs[:z])
        g["l_var.bias"].add_(bs[z:2 * z])
        if need_dgrad:
            dh, _ = self.dense.dgrad(dhead16)
            return dh
        return None


_MLP_ON = os.environ.get("FMRI_MLP") != "off"        # fused latent-discriminator kernels (csrc/mlp.hip)


def _bwd_epi(**kw):
    """``bn_bwd`` argument of ``ConvLayer.dgrad`` when the BatchNorm-backward epilogue is switched on (ops._EPI_BWD)."""
    from . import ops
    return kw if ops._EPI_BWD else None


def refresh_net(net):
    """Everything a sub-network derives from its master parameters -- fp16 GEMM weights, (C,H,W)-permuted BatchNorm
    vectors, the fused heads' concatenated weights -- refreshed NOW on the current stream.  They are otherwise
    refreshed lazily inside the next forward; a recorded forward (HIP graph) must not contain those launches."""
    repack_group(net.group)
    for bn in net.all_bns():
        bn._params()
        if getattr(bn, "_lazy", False):
            bn._running_in()             # reloads only after an outside write of the buffers (load_state_dict)
    heads = getattr(net, "heads", None)
    if heads is not None:
        heads.refresh()


class _Versioned:
    """Adapter so scratch-weight layers follow the owning group's version counter."""

    def __init__(self, group):
        self._g = group

    @property
    def version(self):
        return self._g.version


# ------------------------------------------------------------------------------------------------
class EncoderNet:
    def __init__(self, cfg: ArchConfig, device, channel_in: int = 3):
        self.cfg = cfg
        self.group = FlatGroup(encoder_spec(cfg, channel_in), device)
        g = self.group
        self.convs, self.bns = [], []
        cin = channel_in
        for i, c in enumerate(cfg.encoder_channels[:3]):
            self.convs.append(ConvLayer(g, f"conv.{i}.conv.weight", None, "conv", cin, c, cfg.kernel_size, cfg.stride,
                                        cfg.padding))
            self.bns.append(BatchNorm(g, f"conv.{i}.bn.", c))
            cin = c
        self.size = cin
        hw = cfg.fc_input * cfg.fc_input
        self.fc = DenseLayer(g, "fc.0.weight", None, hw * cin, cfg.fc_output, in_perm=(cin, hw))
        self.fc_bn = BatchNorm(g, "fc.1.", cfg.fc_output)
        self.heads = FusedHeads(g, cfg.fc_output, cfg.latent_dim)

    def all_bns(self):
        return self.bns + [self.fc_bn]

    def forward(self, x16: torch.Tensor, train_stats: bool = True, updates: Optional[int] = None):
        """x16 [B,H,W,8] fp16 -> (head32 [B,2z], ctx).  ``updates`` = number of running-stat updates this call
        stands for (the WAE scripts run the same encoder pass two or three times per step)."""
        acts, raws, svs = [x16], [], []
        h = x16
        upd = (1 if train_stats else 0) if updates is None else updates
        for conv, bn in zip(self.convs, self.bns):
            if bn.eval_mode and not bn.perm:
                # eval: the BatchNorm rides in the convolution's epilogue where its kernel has one (no raw tensor then)
                out = conv.forward(h, affine=bn.eval_affine())
                if conv.aff_applied:
                    raws.append(None); svs.append(None); acts.append(out)
                    h = out
                    continue
                raw = out
            else:
                raw = conv.forward(h, bn_groups=0 if bn.eval_mode else 1)
            h, sv = bn.forward(raw, relu=True, updates=upd, stat_acc=conv.take_stats())
            raws.append(raw)
            svs.append(sv)
            acts.append(h)
        flat = h.reshape(h.shape[0], -1)
        raw_fc, _ = self.fc.forward(flat)
        hfc, svfc = self.fc_bn.forward(raw_fc, relu=True, updates=upd)
        head32 = self.heads.forward(hfc)
        return head32, dict(acts=acts, raws=raws, svs=svs, flat=flat, raw_fc=raw_fc, hfc=hfc, svfc=svfc)

    def backward(self, *args, join: bool = True, **kwargs):
        """``_backward`` + join of the side stream its weight gradients were issued on (ops.side_run).  ``join=False``
        leaves them in flight (the caller joins with ops.join_side() before the gradients are read)."""
        out = self._backward(*args, **kwargs)
        if join:
            join_side()
        return out

    def _backward(self, ctx, dhead16: torch.Tensor, scale: float, after_fc=None, after_fc_join: bool = True):
        """Accumulate encoder parameter gradients of (1/scale)*<dhead16, head>.  ``after_fc`` is called once the
        gradients of fc.0 / fc.1 / l_mu / l_var -- the tail of the flat buffer from ``fc.0.weight`` on, 93 % of its
        bytes -- are issued, so a data-parallel run can start reducing them under the conv backward;
        ``after_fc_join`` first joins the side stream they were issued on (a hook that itself runs on the side stream
        does not need that)."""
        dh = self.heads.backward(ctx["hfc"], dhead16, scale)
        draw_fc, _ = self.fc_bn.backward(ctx["raw_fc"], dh, ctx["svfc"], True, scale)
        self.fc.wgrad(ctx["flat"], draw_fc, scale)
        if after_fc is not None:
            if after_fc_join:
                join_side()                             # fc.0 weight gradient (side stream) is final
            after_fc()
        dflat, _ = self.fc.dgrad(draw_fc)
        d = dflat.reshape(ctx["acts"][3].shape)
        stat = None
        # Issue order inside a layer: the data gradient (main stream: the next link of the backward chain) BEFORE the weight
        # gradient (side stream) -- both become ready at the same moment, and the one issued first gets the free CUs first
        # (same-call A/B of the order, three pairs: 6.540-6.546 ms against 6.562-6.593 per step).
        for i in (2, 1, 0):
            draw, _ = self.bns[i].backward(ctx["raws"][i], d, ctx["svs"][i], True, scale, stat=stat)
            if i > 0:
                _, hi, wi, _ = ctx["acts"][i].shape
                # the data gradient's epilogue masks with block i-1's ReLU and reduces its BatchNorm backward sums
                d = self.convs[i].dgrad(draw, hi, wi, bn_bwd=_bwd_epi(bn=self.bns[i - 1], x=ctx["raws"][i - 1],
                                                                      groups=[(0, ctx["svs"][i - 1])]))
                stat = self.convs[i].take_bwd_stats()
            self.convs[i].wgrad(ctx["acts"][i], draw, scale)


class CognitiveEncoderNet:
    def __init__(self, cfg: ArchConfig, n_voxels: int, device):
        self.cfg, self.n_voxels = cfg, n_voxels
        self.group = FlatGroup(cognitive_encoder_spec(cfg, n_voxels), device)
        g = self.group
        self.fc1 = DenseLayer(g, "fc1.0.weight", None, n_voxels, 1024)
        self.fc1_bn = BatchNorm(g, "fc1.1.", 1024)
        self.heads = FusedHeads(g, 1024, cfg.latent_dim)

    def all_bns(self):
        return [self.fc1_bn]

    def forward(self, fmri16: torch.Tensor, train_stats: bool = True, updates: Optional[int] = None):
        raw, _ = self.fc1.forward(fmri16)
        upd = (1 if train_stats else 0) if updates is None else updates
        h, sv = self.fc1_bn.forward(raw, relu=True, updates=upd)
        return self.heads.forward(h), dict(x=fmri16, raw=raw, h=h, sv=sv)

    def backward(self, *args, join: bool = True, **kwargs):
        """``_backward`` + join of the side stream its weight gradients were issued on (ops.side_run).  ``join=False``
        leaves them in flight (the caller joins with ops.join_side() before the gradients are read)."""
        out = self._backward(*args, **kwargs)
        if join:
            join_side()
        return out

    def _backward(self, ctx, dhead16, scale):
        dh = self.heads.backward(ctx["h"], dhead16, scale)
        draw, _ = self.fc1_bn.backward(ctx["raw"], dh, ctx["sv"], True, scale)
        self.fc1.wgrad(ctx["x"], draw, scale)


# ------------------------------------------------------------------------------------------------
class DecoderNet:
    def __init__(self, cfg: ArchConfig, device, size: Optional[int] = None):
        self.cfg = cfg
        size = cfg.encoder_channels[2] if size is None else size
        self.group = FlatGroup(decoder_spec(cfg, size), device)
        g = self.group
        hw = cfg.fc_input * cfg.fc_input
        self.size0 = size
        self.fc = DenseLayer(g, "fc.0.weight", None, cfg.latent_dim, hw * size, out_perm=(size, hw))
        self.fc_bn = BatchNorm(g, "fc.1.", hw * size, perm=(size, hw))
        chans = [(size, size), (size, cfg.decoder_channels[1]), (cfg.decoder_channels[1], cfg.decoder_channels[2])]
        self.deconvs, self.bns = [], []
        for i, (ci, co) in enumerate(chans):
            self.deconvs.append(ConvLayer(g, f"conv.{i}.conv.weight", None, "deconv", ci, co, cfg.kernel_size,
                                          cfg.stride, cfg.padding, 1 if cfg.output_pad_dec[i] else 0))
            self.bns.append(BatchNorm(g, f"conv.{i}.bn.", co))
        self.c3 = ConvLayer(g, "conv.3.0.weight", "conv.3.0.bias", "conv", cfg.decoder_channels[2],
                            cfg.decoder_channels[3], 5, 1, 2)

    def all_bns(self):
        return [self.fc_bn] + self.bns

    def forward(self, z16: torch.Tensor, groups: int, out: Optional[torch.Tensor] = None, train_stats: bool = True,
                stat_order=None, zscale: Optional[torch.Tensor] = None):
        """z16 [G*B, zp]: G independent decoder calls (own BN batch statistics each, reference
        models/vae_gan.py:279,282) executed as one batch.  ``stat_order`` = order in which the groups'
        running-stat updates are applied (the reference's call order).  ``zscale`` (device fp32 [>= G]): group g's rows
        hold zscale[g] * z (``ops.latent_ranged``: a latent whose sigma left fp16's range); None = all 1.
        Returns (images fp16 [G*B,H,W,8], ctx)."""
        GB = z16.shape[0]
        B = GB // groups
        f = self.cfg.fc_input
        upd = 1 if train_stats else 0
        order = list(range(groups)) if stat_order is None else list(stat_order)
        raw_fc, _ = self.fc.forward(z16)
        act_fc = torch.empty_like(raw_fc)
        grows = lambda t: [t[gi * B:(gi + 1) * B] for gi in range(groups)]
        # the call groups of one layer are ready together: with SyncBN their statistics travel in ONE all-reduce
        ins = None if zscale is None else [zscale[gi:gi + 1] for gi in range(groups)]
        sv_fc = self.fc_bn.forward_groups(grows(raw_fc), True, upd, grows(act_fc), [None] * groups, order, in_scales=ins)
        h = act_fc.reshape(GB, f, f, self.size0)
        acts, raws, svs = [h], [], []
        for dc, bn in zip(self.deconvs, self.bns):
            if bn.eval_mode and not bn.perm:
                # eval: the BatchNorm rides in the transposed convolution's epilogue where its kernel has one
                out_e = dc.forward(h, affine=bn.eval_affine())
                if dc.aff_applied:
                    raws.append(None); svs.append([None] * groups); acts.append(out_e)
                    h = out_e
                    continue
                raw = out_e
            else:
                raw = dc.forward(h, bn_groups=0 if bn.eval_mode else groups)
            act = torch.empty_like(raw)
            sl = bn.forward_groups(grows(raw), True, upd, grows(act), [dc.take_stats(gi) for gi in range(groups)], order)
            raws.append(raw)
            svs.append(sl)
            acts.append(act)
            h = act
        y = self.c3.forward(h, ACT_TANH, out=out)
        return y, dict(z=z16, raw_fc=raw_fc, sv_fc=sv_fc, acts=acts, raws=raws, svs=svs, y=y, B=B, groups=groups,
                       zscale=zscale)

    def backward(self, *args, join: bool = True, **kwargs):
        """``_backward`` + join of the side stream its weight gradients were issued on (ops.side_run).  ``join=False``
        leaves them in flight (the caller joins with ops.join_side() before the gradients are read)."""
        out = self._backward(*args, **kwargs)
        if join:
            join_side()
        return out

    def _backward(self, ctx, cot: torch.Tensor, entries: List[dict]):
        """cot [E*B,H,W,8] fp16 stacks E cotangent blocks w.r.t. the decoder output; entries[e] =
        dict(g=<forward group whose activations it belongs to>, scale=<float>, train=<accumulate param grads>,
        need_dz=<bool>).  Returns {e: dz fp32 [B, z] (true scale)} for entries with need_dz."""
        B = ctx["B"]
        E = len(entries)
        rows = lambda t, i: t[i * B:(i + 1) * B]
        # weight-gradient runs: consecutive training blocks with the same scale through consecutive forward groups are
        # one launch over all their rows (2 x 256 images as one 512-image launch: 638 vs 806 us for the decoder's four
        # conv layers, tools/probes/wgrad_n512.py)
        runs, e = [], 0
        while e < E:
            if not entries[e]["train"]:
                e += 1
                continue
            f = e
            while (f + 1 < E and entries[f + 1]["train"] and entries[f + 1]["scale"] == entries[e]["scale"]
                   and entries[f + 1]["g"] == entries[f]["g"] + 1):
                f += 1
            runs.append((e, f))
            e = f + 1

        def wgrads(layer, x_all, d_all):
            for e0, e1 in runs:
                g0 = entries[e0]["g"]
                layer.wgrad(x_all[g0 * B:(g0 + e1 - e0 + 1) * B], d_all[e0 * B:(e1 + 1) * B], entries[e0]["scale"])

        y = ctx["y"]
        dpre = torch.empty_like(cot)
        for e, en in enumerate(entries):
            colsum = None
            if en["train"]:
                colsum = torch.empty(2 * pad8(self.c3.cout), dtype=torch.float32, device=cot.device)
            # the bias gradient of conv.3 rides in the fold of the column sums (its first cout entries)
            act_backward(rows(y, en["g"]), rows(cot, e), ACT_TANH, colsum, out=rows(dpre, e),
                         dbias=self.c3.bg if en["train"] else None, dbias_scale=1.0 / en["scale"])
        wgrads(self.c3, ctx["acts"][3], dpre)
        _, hi, wi, _ = ctx["acts"][3].shape
        d = self.c3.dgrad(dpre, hi, wi)
        stat = None
        def bn_backward(bn, raw_all, sv_all, d_all, draw_all, stat):
            """BatchNorm backward of every cotangent block of one layer.  With SyncBN the reductions of all blocks run
            first (phase 1) into one [2E][C] buffer that is exchanged ONCE, then the apply passes (phase 2)."""
            sync = bn.reducer is not None
            sums_all = torch.empty(2 * E, bn.C, dtype=torch.float32, device=d_all.device) if sync else None
            for phase in ((1, 2) if sync else (3,)):
                if phase == 2:
                    bn.reducer(sums_all)
                e = 0
                while e < E:
                    en = entries[e]
                    nx = entries[e + 1] if e + 1 < E else None
                    sm = sums_all[2 * e:] if sync else None
                    if nx is not None and nx["g"] == en["g"] and not (nx["train"] and en["train"]):
                        # two adjacent blocks through the same forward activations: one pass over them for both; the
                        # gamma / beta gradients come from whichever of the two trains
                        ps = 1 if nx["train"] else 0
                        tr = nx if nx["train"] else en
                        bn.backward2(rows(raw_all, en["g"]), d_all[e * B:(e + 2) * B], sv_all[en["g"]], True,
                                     tr["scale"] if tr["train"] else None, out=draw_all[e * B:(e + 2) * B],
                                     param_stream=ps, stat=stat, stat_group=e, sums=sm[:4] if sync else None, phase=phase)
                        e += 2
                    else:
                        bn.backward(rows(raw_all, en["g"]), rows(d_all, e), sv_all[en["g"]], True,
                                    en["scale"] if en["train"] else None, out=rows(draw_all, e), stat=stat,
                                    stat_group=e, sums=sm[:2] if sync else None, phase=phase)
                        e += 1

        for i in (2, 1, 0):
            draw = torch.empty_like(d)
            bn_backward(self.bns[i], ctx["raws"][i], ctx["svs"][i], d, draw, stat)
            _, hi, wi, _ = ctx["acts"][i].shape
            if i > 0:
                # block i-1's ReLU mask and BatchNorm backward sums come out of this data gradient's epilogue: one
                # statistics group per cotangent block (its forward call's rows of the saved tensor and statistics)
                d = self.deconvs[i].dgrad(draw, hi, wi, bn_bwd=_bwd_epi(
                    bn=self.bns[i - 1], x=ctx["raws"][i - 1],
                    groups=[(en["g"] * B, ctx["svs"][i - 1][en["g"]]) for en in entries]))
                stat = self.deconvs[i].take_bwd_stats()
            else:
                d = self.deconvs[i].dgrad(draw, hi, wi)
            wgrads(self.deconvs[i], ctx["acts"][i], draw)       # (after the data gradient: see EncoderNet._backward)
        dflat = d.reshape(E * B, -1)
        draw_fc = torch.empty_like(dflat)
        out = {}
        bn_backward(self.fc_bn, ctx["raw_fc"], ctx["sv_fc"], dflat, draw_fc, None)
        wgrads(self.fc, ctx["z"], draw_fc)
        for e, en in enumerate(entries):
            if en.get("need_dz"):
                _, dz32 = self.fc.dgrad(rows(draw_fc, e).contiguous(), want32=True)
                out[e] = dz32 * (1.0 / en["scale"])
                if ctx.get("zscale") is not None:
                    # the group's rows were stored as s * z: d/dz = s * d/d(stored rows)
                    out[e] *= ctx["zscale"][en["g"]:en["g"] + 1]
        return out


# ------------------------------------------------------------------------------------------------
class DiscriminatorNet:
    def __init__(self, cfg: ArchConfig, device, recon_level: int = 3):
        # models/vae_gan.py:139-173: the 'REC' output is the RAW convolution output of block ``recon_level``.  Block 0 is
        # an nn.Sequential, which does not take the reference's ``lay(ten, True)`` call (TypeError), and a level past
        # the last block makes its forward return None: levels 1..3 are the ones the reference can run.
        if recon_level not in (1, 2, 3):
            raise ValueError(f"recon_level={recon_level}: the reference Discriminator works for levels 1, 2, 3 only "
                             "(level 0 raises a TypeError there, levels > 3 return None; models/vae_gan.py:164-173)")
        self.level = recon_level
        self.cfg = cfg
        self.group = FlatGroup(discriminator_spec(cfg), device)
        g = self.group
        d = cfg.discrim_channels
        self.c0 = ConvLayer(g, "conv.0.0.weight", "conv.0.0.bias", "conv", 3, d[0], 5, cfg.stride_gan, 2)
        self.convs, self.bns = [], []
        cin = d[0]
        for i in (1, 2, 3):
            self.convs.append(ConvLayer(g, f"conv.{i}.conv.weight", None, "conv", cin, d[i], cfg.kernel_size,
                                        cfg.stride, cfg.padding))
            self.bns.append(BatchNorm(g, f"conv.{i}.bn.", d[i]))
            cin = d[i]
        hw = cfg.fc_input_gan * cfg.fc_input_gan
        self.fc0 = DenseLayer(g, "fc.0.weight", None, hw * cin, cfg.fc_output_gan, in_perm=(cin, hw))
        self.fc_bn = BatchNorm(g, "fc.1.", cfg.fc_output_gan)
        self.fc3 = DenseLayer(g, "fc.3.weight", "fc.3.bias", cfg.fc_output_gan, 1)

    def all_bns(self):
        return self.bns + [self.fc_bn]

    def forward(self, x16: torch.Tensor, train_stats: bool = True, conv_updates: int = 2, fc_updates: int = 1,
                head: bool = True):
        """x16 [3B,H,W,8] = cat(orig, predicted, sampled).  One pass produces both reference outputs:
        the raw conv-3 features ('REC', models/vae_gan.py:166-173) and the class logits ('GAN', :176-183).
        Conv BN layers receive the reference's two running-stat updates per step in the fused
        step; the per-mode API calls pass conv_updates=1 and head=False for 'REC'."""
        a0 = self.c0.forward(x16, ACT_RELU)
        acts, raws, svs = [a0], [], []
        h = a0
        lv = self.level
        for li, (conv, bn) in enumerate(zip(self.convs, self.bns)):
            if not head and li >= lv:
                break                     # a 'REC' call ends at block ``recon_level`` (models/vae_gan.py:166-173)
            if bn.eval_mode and not bn.perm and li != lv - 1:   # (block ``recon_level``'s RAW output is the 'REC' tensor)
                out_e = conv.forward(h, affine=bn.eval_affine())
                if conv.aff_applied:
                    raws.append(None); svs.append(None); acts.append(out_e)
                    h = out_e
                    continue
                raw = out_e
            else:
                raw = conv.forward(h, bn_groups=0 if bn.eval_mode else 1)
            # one pass standing for the reference's REC + GAN passes (conv_updates = 2): the REC pass never reaches the
            # blocks above ``recon_level``, they take the GAN pass's update only
            upd = conv_updates if li < lv else min(conv_updates, 1)
            h, sv = bn.forward(raw, relu=True, updates=upd if train_stats else 0, stat_acc=conv.take_stats())
            raws.append(raw)
            svs.append(sv)
            acts.append(h)
        ctx = dict(x=x16, acts=acts, raws=raws, svs=svs)
        if not head:
            return raws[lv - 1], None, ctx
        return raws[lv - 1], self.forward_head(ctx, train_stats, fc_updates), ctx

    def forward_head(self, ctx, train_stats: bool = True, fc_updates: int = 1):
        """The 'GAN' head (flatten -> fc.0 -> BN1d -> ReLU -> fc.3, models/vae_gan.py:176-183) on the conv activations of
        an earlier ``forward(..., head=False)`` (``ctx``, completed in place): class logits [3B, 1] fp32."""
        h = ctx["acts"][3]
        flat = h.reshape(h.shape[0], -1)
        raw_fc, _ = self.fc0.forward(flat)
        hfc, svfc = self.fc_bn.forward(raw_fc, relu=True, updates=fc_updates if train_stats else 0)
        _, logit32 = self.fc3.forward(hfc, ACT_NONE, want16=False, want32=True)
        ctx.update(flat=flat, raw_fc=raw_fc, hfc=hfc, svfc=svfc)
        return logit32

    def conv_running_again(self, ctx, updates: int = 1):
        """Apply the conv blocks' BatchNorm running-statistics update once more with the batch statistics saved in ``ctx``
        (a repeated forward call on the same input, see BatchNorm.update_running_again)."""
        for bn, sv in zip(self.bns, ctx["svs"]):
            if sv is not None:
                bn.update_running_again(sv, updates)

    def backward(self, *args, join: bool = True, **kwargs):
        """``_backward`` + join of the side stream its weight gradients were issued on (ops.side_run).  ``join=False``
        leaves them in flight (the caller joins with ops.join_side() before the gradients are read)."""
        out = self._backward(*args, **kwargs)
        if join:
            join_side()
        return out

    def _backward(self, ctx, dlogit16: Optional[torch.Tensor], scale_a: float, dfeat16: Optional[torch.Tensor],
                 scale_b: float, train: bool, img_rows: Optional[slice], img_streams=(True, True),
                 train_b: bool = False):
        """Two cotangent streams through one saved forward:
             A: d(sum bce)/d logit  (dlogit16 [3B,8], scale_a) -- accumulates discriminator grads if ``train``
             B: d(sum mse)/d raw conv-3 features (dfeat16 [3B,h,w,c], scale_b) -- data gradient only
           Returns (dimg_A, dimg_B): cotangents w.r.t. input images ``img_rows`` (None if not requested)."""
        n3 = ctx["x"].shape[0]
        streams = []
        top = 2                             # block whose raw output the cotangents enter at (index into convs / raws)
        if dlogit16 is None and dfeat16 is not None:
            top = self.level - 1             # a 'REC' call alone: the feature cotangent enters at block ``recon_level``
        elif dfeat16 is not None and self.level != 3:
            raise NotImplementedError("two cotangent streams through one discriminator pass (the fused steps) are "
                                      "implemented for recon_level=3; other levels run through the module API")
        if dlogit16 is not None:
            if train:
                self.fc3.wgrad(ctx["hfc"], dlogit16, scale_a)
                self.fc3.bias_grad(dlogit16, scale_a)
            dh, _ = self.fc3.dgrad(dlogit16)
            draw_fc, _ = self.fc_bn.backward(ctx["raw_fc"], dh, ctx["svfc"], True, scale_a if train else None)
            if train:
                self.fc0.wgrad(ctx["flat"], draw_fc, scale_a)
            dflat, _ = self.fc0.dgrad(draw_fc)
            d3 = dflat.reshape(ctx["acts"][3].shape)
            # caller-provided stack [A rows | B rows] with B already in place (steps._start_cotangents): A goes into its
            # first half and the two-stream batch needs no concatenation copy
            stack = getattr(dfeat16, "_fmri_stack", None) if dfeat16 is not None else None
            if stack is not None and stack.shape[0] != 2 * n3:
                stack = None
            draw3, _ = self.bns[2].backward(ctx["raws"][2], d3, ctx["svs"][2], True, scale_a if train else None,
                                            out=stack[:n3].reshape(ctx["raws"][2].shape) if stack is not None else None)
            streams.append(dict(d=draw3, scale=scale_a, train=train, img=img_streams[0]))
        else:
            stack = None
        if dfeat16 is not None:
            streams.append(dict(d=dfeat16.reshape(ctx["raws"][top].shape), scale=scale_b, train=train_b,
                                img=img_streams[1]))
        S = len(streams)
        if S > 1 and stack is not None:
            d = stack.reshape((2 * n3,) + tuple(ctx["raws"][2].shape[1:]))
        else:
            d = torch.cat([s["d"] for s in streams], 0) if S > 1 else streams[0]["d"]
        rows = lambda t, i: t[i * n3:(i + 1) * n3]
        # conv3 .. conv1 (from block ``top`` down)
        for li in range(top, -1, -1):
            def wgrads(li=li, d=d):             # (issued after the layer's data gradient: see EncoderNet._backward)
                for si, s in enumerate(streams):
                    if s["train"]:
                        self.convs[li].wgrad(ctx["acts"][li], rows(d, si), s["scale"])
            _, hi, wi, _ = ctx["acts"][li].shape
            if li == 0:
                wgrads()
            if li > 0:
                # the data gradient's epilogue masks with block li-1's ReLU and reduces its BatchNorm backward sums, one
                # statistics group per cotangent stream (all read the same saved forward tensor)
                epi = _bwd_epi(bn=self.bns[li - 1], x=ctx["raws"][li - 1], groups=[(0, ctx["svs"][li - 1])] * S)
                ztail = getattr(dfeat16, "_fmri_zero_tail", 0) if (li == 2 and dfeat16 is not None and S > 1) else 0
                if ztail and epi is None and d.shape[0] == S * n3:
                    # stream B enters at conv3's raw output with exact zeros on the sampled images' rows -- the last rows
                    # of the stack: conv3's data gradient (per image) skips them, its result there is zero
                    live = d.shape[0] - ztail
                    dact = torch.empty(d.shape[0], hi, wi, self.convs[li].cinp, dtype=torch.float16, device=d.device)
                    self.convs[li].dgrad(d[:live], hi, wi, out=dact[:live])
                    dact[live:].zero_()
                else:
                    dact = self.convs[li].dgrad(d, hi, wi, bn_bwd=epi)
                wgrads()
                stat = self.convs[li].take_bwd_stats()
                dn = torch.empty_like(dact)
                if S == 2 and not streams[1]["train"]:
                    # both streams in one pass over the saved forward tensor (gamma / beta gradients from stream A only)
                    self.bns[li - 1].backward2(ctx["raws"][li - 1], dact, ctx["svs"][li - 1], True,
                                               streams[0]["scale"] if streams[0]["train"] else None, out=dn, stat=stat)
                else:
                    for si, s in enumerate(streams):
                        self.bns[li - 1].backward(ctx["raws"][li - 1], rows(dact, si), ctx["svs"][li - 1], True,
                                                  s["scale"] if s["train"] else None, out=rows(dn, si), stat=stat,
                                                  stat_group=si)
                d = dn
        # conv1 data gradient, conv0 (bias + ReLU).  Below the last BatchNorm the images are independent, so a stream
        # that does not train the discriminator only needs the rows whose image gradient is wanted.
        a0 = ctx["acts"][0]
        _, hi, wi, _ = a0.shape
        full = slice(0, n3)
        need = [full if s["train"] else (img_rows if (img_rows is not None and s["img"]) else None) for s in streams]
        # conv0's ReLU backward rides in the epilogue of conv1's data gradient where its kernel has one (act_applied),
        # and the bias gradient comes out of the weight-gradient kernel: no pass over the 64 x 64 x 32 cotangent
        masked = [False] * S
        if all(nd == full for nd in need):
            dact = self.convs[0].dgrad(d, hi, wi)
            dacts = [rows(dact, si) for si in range(S)]
        else:
            dacts = []
            for si, nd in enumerate(need):
                if nd is None:
                    dacts.append(None)
                    continue
                dacts.append(self.convs[0].dgrad(rows(d, si)[nd], hi, wi, relu_y=a0[nd]))
                masked[si] = self.convs[0].act_applied
        outs = []
        for si, s in enumerate(streams):
            if need[si] is None:
                outs.append(None)
                continue
            if masked[si]:
                dpre = dacts[si]
                if s["train"]:
                    self.c0.wgrad(ctx["x"], dpre, s["scale"], bias_too=True)
            else:
                colsum = None
                if s["train"]:
                    colsum = torch.empty(2 * self.c0.coutp, dtype=torch.float32, device=d.device)
                dpre = act_backward(a0[need[si]], dacts[si], ACT_RELU, colsum,
                                    dbias=self.c0.bg if s["train"] else None, dbias_scale=1.0 / s["scale"])
                if s["train"]:
                    self.c0.wgrad(ctx["x"], dpre, s["scale"])
            if img_rows is not None and s["img"]:
                _, xh, xw, _ = ctx["x"].shape
                sub = dpre[img_rows] if need[si] == full else dpre
                outs.append(self.c0.dgrad(sub.contiguous(), xh, xw))
            else:
                outs.append(None)
        res = [None, None]
        k = 0
        if dlogit16 is not None:
            res[0] = outs[k]
            k += 1
        if dfeat16 is not None:
            res[1] = outs[k]
        return res[0], res[1]


# ------------------------------------------------------------------------------------------------
class WaeDiscriminatorNet:
    """Latent-space discriminator MLP z -> 512 -> 512 -> 512 -> 512 -> 1 (models/vae_gan.py:499-529)."""

    def __init__(self, cfg: ArchConfig, device, dim_h: int = 512):
        self.cfg = cfg
        self.group = FlatGroup(wae_discriminator_spec(cfg, dim_h), device)
        dims = [cfg.latent_dim, dim_h, dim_h, dim_h, dim_h, 1]
        self.layers = [DenseLayer(self.group, f"main.{idx}.weight", f"main.{idx}.bias", dims[j], dims[j + 1])
                       for j, idx in enumerate((0, 2, 4, 6, 8))]

    def all_bns(self):
        return []

    # The fused kernels (csrc/mlp.hip: one launch forward, one for the backward data path) take this network's own
    # packed weights; FMRI_MLP=off (or a width they do not cover) keeps the layer-by-layer path below.
    def _fused_ok(self, z16) -> bool:
        dims = [l.n_out for l in self.layers]
        return (_MLP_ON and len(self.layers) == 5 and dims[:4] == [512] * 4 and dims[4] == 1
                and z16.shape[1] % 64 == 0 and 64 <= z16.shape[1] <= 256)

    @staticmethod
    def _ptrs(tensors):
        import ctypes
        arr = (ctypes.c_void_p * len(tensors))()
        for i, t in enumerate(tensors):
            arr[i] = None if t is None else t.data_ptr()
        return arr

    def forward(self, z16: torch.Tensor):
        if self._fused_ok(z16):
            import ctypes
            M, Zp = z16.shape
            dev = z16.device
            ws = [l.pw_f.get() for l in self.layers]
            kps = (ctypes.c_int * 5)(*[l.pw_f.kpads[0] for l in self.layers])
            hbuf = torch.empty(4, M, 512, dtype=torch.float16, device=dev)
            logit32 = torch.empty(M, 1, dtype=torch.float32, device=dev)
            lib.call("fmri_mlp_fwd", _P(z16), M, Zp, 512, self._ptrs(ws), kps, self._ptrs([l.b for l in self.layers]),
                     self._ptrs([hbuf[i] for i in range(4)]), _P(logit32))
            return logit32, dict(hs=[z16] + [hbuf[i] for i in range(4)], fused=True)
        hs = [z16]
        h = z16
        for l in self.layers[:-1]:
            h, _ = l.forward(h, ACT_RELU)
            hs.append(h)
        _, logit32 = self.layers[-1].forward(h, ACT_NONE, want16=False, want32=True)
        return logit32, dict(hs=hs)

    def backward(self, *args, join: bool = True, **kwargs):
        """``_backward`` + join of the side stream its weight gradients were issued on (ops.side_run).  ``join=False``
        leaves them in flight (the caller joins with ops.join_side() before the gradients are read)."""
        out = self._backward(*args, **kwargs)
        if join:
            join_side()
        return out

    def _backward(self, ctx, dlogit16, scale, train: bool, need_dz: bool):
        if ctx.get("fused"):
            return self._backward_fused(ctx, dlogit16, scale, train, need_dz)
        d = dlogit16
        hs = ctx["hs"]
        for j in range(4, -1, -1):
            l = self.layers[j]
            if train:
                l.wgrad(hs[j], d, scale)
                l.bias_grad(d, scale)
            if j == 0 and not need_dz:
                return None
            if j == 0:
                _, dz32 = l.dgrad(d, want32=True)
                return dz32 * (1.0 / scale)
            dn, _ = l.dgrad(d)
            d = act_backward(hs[j], dn, ACT_RELU)
        return None

    def _backward_fused(self, ctx, dlogit16, scale, train: bool, need_dz: bool):
        """One launch for the data path (cotangents of the four hidden pre-activations, bias gradients, dz); the five
        weight gradients -- reductions over all rows -- are fmri_wgrad launches on the side stream."""
        import ctypes
        hs = ctx["hs"]
        z16 = hs[0]
        M, Zp = z16.shape
        dev = z16.device
        Z = self.layers[0].k_in
        dbuf = torch.empty(4, M, 512, dtype=torch.float16, device=dev)
        dz32 = torch.empty(M, Z, dtype=torch.float32, device=dev) if need_dz else None
        wds = [(self.layers[i].pw_d.get() if (i > 0 or need_dz) else None) for i in range(4)]
        kpd = (ctypes.c_int * 4)(*[self.layers[i].pw_d.kpads[0] for i in range(4)])
        from . import ops
        det = ops.deterministic()        # bias gradients as fixed-order column sums instead of the kernel's atomics
        dbias = self._ptrs([l.bg for l in self.layers]) if (train and not det) else None
        lib.call("fmri_mlp_bwd", _P(dlogit16), dlogit16.shape[1], M, Zp, Z, 512, self._ptrs(hs[1:5]),
                 _P(self.layers[4].pw_f.get()), self._ptrs(wds), kpd, self._ptrs([dbuf[i] for i in range(4)]), dbias,
                 _P(dz32), 1.0 / scale)
        if train:
            deltas = [dbuf[0], dbuf[1], dbuf[2], dbuf[3], dlogit16]
            for j in range(5):
                self.layers[j].wgrad(hs[j], deltas[j], scale)
                if det:
                    self.layers[j].bias_grad(deltas[j], scale)
        return dz32
