"""Fused training steps: our restatement of the inline loop bodies of the reference scripts.

    Stage1Step   train/train_vgan_stage1.py:316-432   (mode 'vae-gan')
    Stage2Step   train/train_vgan_stage2.py:321-407
    Stage3Step   train/train_vgan_stage3.py:324-411
    (WAE Stage I/II/III and the Dual WAE+VAE/GAN step live in wae_steps.py)

Contract: one forward -> the three gradient sets, each of its own loss w.r.t. its own
sub-network, all evaluated at the pre-update weights -> gated optimizer steps.  The discriminator is run
once (REC+GAN fused, BN running stats updated twice like the reference); its backward carries two
cotangent streams (A = d L_dis, B = d sum(mse)) so that decoder gets lambda*B-(1-lambda)*A and the
encoder gets B (+KL) from a single saved forward.  Nothing in a step synchronises with the host: the
equilibrium gate is evaluated on the device and gates the fused optimizer kernels through a flag, and the
fp16 cotangent streams are normalised by device-side factors (see csrc/loss.hip) that the optimizer
kernels divide out again.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np
import torch

from . import lib, ops
from .nets import (CognitiveEncoderNet, DecoderNet, DiscriminatorNet, EncoderNet, WaeDiscriminatorNet,
                   refresh_net)
from .ops import axpby, images_to_nhwc, nhwc_to_images, pad8, require_gpu, rows_to_f16
from .params import ArchConfig

_P = lib.ptr

# slots of the fp32 scalar block (csrc/loss.hip enum Slot)
(S_BCE_O, S_BCE_P, S_BCE_S, S_KL, S_MSE, S_NLE, S_LENC, S_LDIS, S_LDEC, S_DL2, S_NA, S_NB, S_RATIO,
 S_ONE, S_ESQ, S_NE, S_NP, S_C1, S_C2, S_C3, S_GDEC, S_KLW) = range(22)
S_ZMAX = 22         # slots [22, 26): max |z| of the latent batch of decoder group 0..3 (ops.latent_ranged; zero at step start)
# loss compositions of train/train_vgan_stage1.py:359-388 (csrc/loss.hip enum Mode)
MODES = {"vae-gan": 0, "beta-vae": 1, "dcgan": 2, "vae": 3}
N_REDUCED = 10      # slots [0, N_REDUCED) are sums over the batch -> all-reduced in data-parallel runs
LOG_KEYS = ("bce_orig", "bce_pred", "bce_samp", "kl", "mse", "nle", "loss_encoder", "loss_discriminator",
            "loss_decoder")


@dataclass
class GanHyper:
    """configs/gan_config.py:19-31."""
    lr: float = 1e-4
    lambda_mse: float = 1e-6
    margin: float = 0.35
    equilibrium: float = 0.68
    alpha: float = 0.9
    eps: float = 1e-8
    beta: float = 1.0      # KL factor of mode 'beta-vae' (configs/gan_config.py:32)


@dataclass
class Scales:
    """Static power-of-two factors on top of the device-side unit-RMS normalisation of each cotangent
    stream (stored = true * norm * scale): they centre the streams inside fp16's normal range."""
    a: float = 512.0       # d L_dis stream through the discriminator (starts at unit RMS per logit)
    b: float = 16.0        # d sum(mse) stream through discriminator / decoder (unit RMS per feature)
    dec: float = 2048.0    # lambda*B - (1-lambda)*A through the decoder (carries norm nA)
    enc: float = 16.0      # encoder backward (carries norm nB)
    p: float = 16.0        # d nle / d x_tilde through the decoder (mode 'vae'; carries norm nP)


class _Optim:
    """Fused RMSprop / Adam over a FlatGroup, optionally gated by a device flag; ``gdev`` is the device
    normalisation factor still carried by the gradients (divided out inside the kernel).  The learning rate (and
    Adam's step count) live in device memory: ``set_lr`` -- the per-epoch ExponentialLR / StepLR of the scripts
    (train_vgan_stage1.py:448-450) -- reaches a step that was recorded into a HIP graph."""

    def __init__(self, group, kind="rmsprop", lr=1e-4, alpha=0.9, eps=1e-8, betas=(0.5, 0.999)):
        self.g, self.kind, self.alpha, self.eps, self.betas = group, kind, alpha, eps, betas
        self.s1 = torch.zeros_like(group.data)
        self.s2 = torch.zeros_like(group.data) if kind == "adam" else None
        self.t = 0
        self._lr = float(lr)
        self.lr_dev = torch.full((1,), float(lr), dtype=torch.float32, device=group.device)
        self.t_dev = torch.zeros(1, dtype=torch.int32, device=group.device) if kind == "adam" else None

    @property
    def lr(self) -> float:
        return self._lr

    def set_lr(self, lr: float):
        self._lr = float(lr)
        self.lr_dev.fill_(float(lr))

    def step(self, flag: Optional[torch.Tensor] = None, clamp: float = 0.0, gdev: Optional[torch.Tensor] = None):
        g = self.g
        if self.kind == "rmsprop" and ops.apply_group(g, self.s1, self.lr_dev, self.alpha, self.eps, flag, gdev, clamp):
            return                               # deferred gradients: update + fp16 copies in one launch (ops.begin_grads)
        if self.kind == "rmsprop":
            lib.note(bytes=20.0 * g.numel)       # read p, g, v; write p, v
            lib.call("fmri_rmsprop_dev", _P(g.data), _P(g.grad), _P(self.s1), g.numel, _P(self.lr_dev), self.alpha,
                     self.eps, 1.0, _P(gdev), clamp, _P(flag))
        else:
            ops.flush_pending(g)                 # (Adam has no fused form: deferred gradients -> reference layout)
            self.t += 1
            b1, b2 = self.betas
            lib.call("fmri_counter_inc", _P(self.t_dev))
            lib.note(bytes=28.0 * g.numel)       # read p, g, m, v; write p, m, v
            lib.call("fmri_adam_dev", _P(g.data), _P(g.grad), _P(self.s1), _P(self.s2), g.numel, _P(self.lr_dev), b1, b2,
                     self.eps, _P(self.t_dev), 1.0, _P(gdev), clamp, _P(flag))
        g.version += 1


class _SegmentRecorder:
    """Records a step as an alternating list of HIP graphs and eagerly issued collectives.

    Collectives stay outside the graphs (they are issued exactly as in the eager step, RCCL sees nothing new); every
    run of kernel launches between two collectives becomes one graph.  A data-parallel step is then ~37 graph launches
    + 36 collective calls on the host instead of ~400 kernel launches."""

    def __init__(self):
        self.pool = torch.cuda.graph_pool_handle()
        self.items = []          # torch.cuda.CUDAGraph or a zero-argument callable
        self.cur = None
        # The collectives run on a stream of their own, never on the one being recorded: the backend records its work
        # events on the stream a blocking collective is issued from, its watchdog THREAD polls them, and HIP refuses to
        # query an event whose stream has meanwhile entered capture (hipErrorCapturedEvent) -- the watchdog then
        # takes the process down (seen once in ~6 runs of the one-rank RCCL test, at the capture that follows a
        # SyncBN exchange).
        self.comm = torch.cuda.Stream()

    def _on_comm(self, fn):
        cur = torch.cuda.current_stream()
        self.comm.wait_stream(cur)
        with torch.cuda.stream(self.comm):
            fn()
        cur.wait_stream(self.comm)

    def begin(self):
        self.cur = torch.cuda.CUDAGraph()
        self.cur.capture_begin(pool=self.pool, capture_error_mode="thread_local")

    def end(self):
        ops.join_side()          # a capture cannot end with forked side-stream work (weight gradients) un-joined
        self.cur.capture_end()
        self.items.append(self.cur)
        self.cur = None

    def collective(self, fn):
        self.end()
        run = lambda: self._on_comm(fn)
        run()                    # communicator warm, same call order as at replay; operates on not-yet-computed data
        self.items.append(run)
        self.begin()

    def replay(self):
        for it in self.items:
            if isinstance(it, torch.cuda.CUDAGraph):
                it.replay()
            else:
                it()


_SMALL_GROUP = {}


def _small_group(dist):
    """The second communicator (small messages), created once per process and default group -- not once per step object."""
    key = id(dist.group.WORLD)
    if key not in _SMALL_GROUP:
        _SMALL_GROUP.clear()
        _SMALL_GROUP[key] = dist.new_group()
    return _SMALL_GROUP[key]


class _Dist:
    """Data-parallel glue (one process per GPU, RCCL): SUM all-reduce of flat gradients / loss scalars and
    the SyncBN statistic exchange.  Inactive (world size 1) unless torch.distributed is initialised."""

    def __init__(self, enabled: bool, sync_bn: bool = True):
        import torch.distributed as dist
        self.dist = dist
        # FMRI_FORCE_DIST=1: take the collective path even with one rank (single-GPU rehearsal of the RCCL calls)
        force = os.environ.get("FMRI_FORCE_DIST") == "1"
        self.on = enabled and dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force)
        self.world = dist.get_world_size() if self.on else 1
        self.sync_bn = sync_bn
        self.pending = []
        # A communicator of its own for the SMALL collectives of the main stream (loss scalars, the stream-norm scalar,
        # SyncBN sums).  Collectives of one communicator run in issue order on its stream: behind a 44 MB gradient
        # all-reduce that is itself waiting for the side stream's weight gradients, a 40-byte all-reduce of the main
        # stream would stall the whole backward pass (head-of-line blocking); on a second communicator it does not.
        # FMRI_SMALL_COMM=off: everything on the default communicator (collectives then run in issue order, the order of
        # the program on every rank) -- the fallback should two communicators driven from two streams ever misbehave on a
        # real multi-GPU node; costs the head-of-line blocking described above
        self.small = _small_group(dist) if (self.on and os.environ.get("FMRI_SMALL_COMM") != "off") else None
        self.recorder: Optional[_SegmentRecorder] = None     # set while a step is recorded into graph segments

    SMALL = 1 << 16          # elements: at most this many go through the small-message communicator

    def all_reduce(self, t: torch.Tensor):
        """Blocking SUM all-reduce on the CURRENT stream (the host does not wait): scalars / statistics on the
        small-message communicator, gradient buffers on the default one."""
        if not self.on:
            return
        grp = self.small if t.numel() <= self.SMALL else None
        if self.recorder is not None:
            self.recorder.collective(lambda: self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=grp))
        else:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=grp)

    def all_reduce_max(self, t: torch.Tensor):
        """Blocking MAX all-reduce of a few scalars on the current stream (the latent range of a SyncBN step)."""
        if not self.on:
            return
        fn = lambda: self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.small)
        if self.recorder is not None:
            self.recorder.collective(fn)
        else:
            fn()

    def all_reduce_async(self, t: torch.Tensor):
        """Start a SUM all-reduce (same communicator, so it queues behind / ahead of the blocking ones in program
        order on every rank) and return immediately; `wait_all` joins it with the compute stream."""
        if not self.on:
            return

        def issue():
            self.pending.append(self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, async_op=True))
        if self.recorder is not None:
            self.recorder.collective(issue)
        else:
            issue()

    def wait_all(self):
        if not self.on:
            return

        def join():
            for h in self.pending:
                h.wait()
            self.pending.clear()
        if self.recorder is not None:
            self.recorder.collective(join)
        else:
            join()

    def bn_reducer(self):
        if not (self.on and self.sync_bn):
            return None

        def red(sums):
            self.all_reduce(sums)
            return self.world
        return red


def _attach_reducers(nets, d: _Dist):
    r = d.bn_reducer()
    for n in nets:
        for bn in n.all_bns():
            bn.reducer = r


class _GanStepBase:
    """Shared pieces of the Stage-I/II/III steps: loss kernels, gate, scalar block, logging."""

    def _init_common(self, device, hp, scales, distributed, sync_bn, nets):
        # fresh instances: GanHyper / Scales are mutable (the per-epoch decays are written as step.set_hyper(...))
        self.hp = GanHyper() if hp is None else hp
        self.sc = Scales() if scales is None else scales
        self.mode = "vae-gan"
        self.device = torch.device(device)
        self.scal = torch.zeros(32, dtype=torch.float32, device=device)
        # [lambda_mse, equilibrium, margin, beta] on the device: read by the gate kernel, so a recorded step follows
        # the per-epoch decays (train_vgan_stage1.py:451-458)
        self.hp_dev = torch.tensor([self.hp.lambda_mse, self.hp.equilibrium, self.hp.margin, self.hp.beta],
                                   dtype=torch.float32, device=device)
        self.flags = torch.zeros(2, dtype=torch.int32, device=device)
        # per decoder call group: the power of two its latent rows are stored scaled by (ops.latent_ranged); the groups
        # fed with caller noise keep 1
        self.zs = torch.ones(4, dtype=torch.float32, device=device)
        self.esq64 = torch.zeros(1, dtype=torch.float64, device=device)      # sum of squares of _renorm
        self.dd = _Dist(distributed, sync_bn)
        _attach_reducers(nets, self.dd)
        self.fw: Dict[str, object] = {}

    def _reduce_async(self, group, part=None):
        """Data parallel: the sub-network's deferred weight gradients -> reference layout (one launch), then the
        asynchronous SUM all-reduce of the gradient buffer (or of ``part`` of it).  Nothing on one GPU: the gradients
        stay deferred until the update (ops.apply_group)."""
        if self.dd.on:
            ops.materialize_grads(group)
            self.dd.all_reduce_async(group.grad if part is None else part)

    def _slot(self, i):
        return self.scal[i:i + 1]

    def _latent(self, head32, eps, z16_rows, group: int, kl_slot: Optional[int]):
        """z = eps * sigma + mu of decoder call group ``group`` into ``z16_rows``, range-safe (ops.latent_ranged): the
        rows are stored at the power-of-two scale self.zs[group]; with SyncBN the decoder's batch is the global one, so
        the range is the global maximum."""
        B, Z = head32.shape[0], head32.shape[1] // 2
        ops.latent_ranged(head32, eps, B, Z, z16_rows, self._slot(S_ZMAX + group), self.zs[group:group + 1],
                          kl_total=None if kl_slot is None else self._slot(kl_slot), sample=True,
                          max_reduce=self.dd.all_reduce_max if (self.dd.on and self.dd.sync_bn) else None)

    def set_hyper(self, lr: Optional[float] = None, lambda_mse: Optional[float] = None,
                  equilibrium: Optional[float] = None, margin: Optional[float] = None, beta: Optional[float] = None):
        """The epoch-end updates of the scripts (lr_*.step(), margin *= decay_margin, equilibrium *= decay_equilibrium,
        lambda_mse *= decay_mse; train_vgan_stage1.py:448-458) -- host copies and the device values the kernels read."""
        hp = self.hp
        if lr is not None:
            hp.lr = float(lr)
            for o in (getattr(self, n, None) for n in ("opt_enc", "opt_dec", "opt_dis")):
                if o is not None:
                    o.set_lr(lr)
        for name, v in (("lambda_mse", lambda_mse), ("equilibrium", equilibrium), ("margin", margin), ("beta", beta)):
            if v is not None:
                setattr(hp, name, float(v))
        self.hp_dev.copy_(torch.tensor([hp.lambda_mse, hp.equilibrium, hp.margin, hp.beta], dtype=torch.float32))

    def _gan_losses(self, feat, logit32, B, x16, xt16, H, W):
        """BCE / feature-mse / pixel terms of VaeGan.loss into the scalar block (+ all-reduce)."""
        dev = feat.device
        F = feat[0].numel()
        prob = torch.empty(3 * B, dtype=torch.float32, device=dev)
        lib.call("fmri_gan_head_parts", _P(logit32), 1, B, _P(prob), _P(self.scal), self._dis_parts())
        lib.call("fmri_feat_mse", _P(feat), B, F, None, _P(self._slot(S_MSE)))
        lib.call("fmri_pixel_sq", _P(x16), _P(xt16), B * H * W, 3, 8, _P(self._slot(S_NLE)), None, 1.0)
        if not getattr(self, "_defer_loss_reduce", False):
            self.dd.all_reduce(self.scal[:N_REDUCED])
        return prob, F

    def _dis_parts(self) -> int:
        """Terms of the back-propagated discriminator loss: orig | pred | sampled ('dcgan' / 'vae': orig + sampled)."""
        return 5 if self.mode in ("dcgan", "vae") else 7

    def _gate(self, B_global, F, gate_on=True, force_dis=-1, force_dec=-1):
        fw = self.fw
        lib.call("fmri_compose_gate_dev", _P(self.scal), _P(self.flags), float(B_global), float(F),
                 float(3 * fw["H"] * fw["W"]), _P(self.hp_dev), MODES[self.mode], 1 if gate_on else 0, force_dis,
                 force_dec)

    def _start_cotangents(self, feat, logit32, B):
        """fp16 starting cotangents of stream A (logits) and stream B (raw conv-3 features)."""
        dev = feat.device
        dlogit16 = torch.empty(3 * B, 8, dtype=torch.float16, device=dev)
        lib.call("fmri_gan_head_bwd_parts", _P(logit32), 1, B, _P(dlogit16), 8, self.sc.a, _P(self._slot(S_NA)),
                 self._dis_parts())
        # stream B's cotangent is written straight into the second half of the buffer that stacks both streams for the
        # discriminator's conv backward (DiscriminatorNet fills the first half with stream A: no concatenation copy)
        stack = torch.empty((2 * feat.shape[0],) + tuple(feat.shape[1:]), dtype=feat.dtype, device=dev)
        dfeat16 = stack[feat.shape[0]:]
        dfeat16._fmri_stack = stack
        dfeat16._fmri_zero_tail = B          # the rows of the sampled images are exact zeros (fmri_feat_mse_bwd)
        lib.call("fmri_feat_mse_bwd", _P(feat), B, feat[0].numel(), _P(dfeat16), self.sc.b, _P(self._slot(S_NB)))
        return dlogit16, dfeat16

    def _renorm(self, x32: torch.Tensor, scale: float, factor_in, rows_global: int):
        """fp32 cotangent -> unit-RMS fp16 (times ``scale``); S_NE <- (*factor_in) / rms.  The sum of squares is
        all-reduced so that every data-parallel rank applies the same factor."""
        n = x32.numel()
        # the sum of squares in double precision (the KL term's 0.5 * (exp(logvar) - 1), squared, leaves fp32 at
        # logvar > 44), cleared by the launch itself: a second backward() after one gate() starts from zero as well
        esq = self.esq64
        lib.call("fmri_sumsq_f64", _P(x32), n, _P(esq), 1)
        self.dd.all_reduce(esq)
        out = torch.empty(x32.shape, dtype=torch.float16, device=x32.device)
        count = float(rows_global) * (n // x32.shape[0])
        lib.call("fmri_renorm_f64", _P(x32), _P(out), n, float(scale), _P(esq), count, _P(factor_in),
                 _P(self._slot(S_NE)))
        return out

    def capture(self, *static_inputs, warmup: int = 2):
        """Record one ``step(*static_inputs)`` into a HIP graph and return a zero-argument callable that replays it.

        Every launch of a step is enqueue-only and nothing inside a step synchronises with the host (the equilibrium
        gate, the stream normalisation and the optimizer gating all live on the device), so the ~370 launches of a
        step can be replayed as one graph: the step time then no longer depends on how fast the host can issue them.
        Inputs are read from ``static_inputs`` at every replay -- copy each new batch into those tensors.  Learning
        rates, lambda, margin, equilibrium and beta are device-resident (``set_lr`` / ``set_hyper``): a replayed step follows
        their schedules (Adam's step count of the WAE steps lives on the device too, ``wae_steps``).  In a data-parallel run the collectives
        are kept out of the graphs (see _SegmentRecorder)."""
        # The weight-gradient side stream (ops.side_run) is switched off while recording: a replayed HIP graph runs
        # its parallel branches no faster than a chain (measured 8.8 ms chained, 9.0-9.5 ms with the fork/join
        # branches, 8.4 ms eager with two streams), so the recorded step keeps everything on one stream.
        side_was = ops._SIDE["on"]
        ops.join_side()
        ops._SIDE["on"] = False
        try:
            return self._capture(static_inputs, warmup)
        finally:
            ops._SIDE["on"] = side_was

    def _capture(self, static_inputs, warmup):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self.step(*static_inputs)
        torch.cuda.current_stream().wait_stream(side)
        if self.dd.on:
            # data-parallel: the collectives stay eager, the launches between them become graphs
            rec = _SegmentRecorder()
            self.dd.recorder = rec
            try:
                with torch.cuda.stream(side):
                    rec.begin()
                    self.step(*static_inputs)
                    rec.end()
            except BaseException:
                if rec.cur is not None:          # leave no stream capture open behind a failed recording
                    try:
                        rec.cur.capture_end()
                    except Exception:
                        pass
                raise
            finally:
                self.dd.recorder = None
            torch.cuda.current_stream().wait_stream(side)
            self._graph = rec
            return self._versioned(rec.replay)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self.step(*static_inputs)
        self._graph = graph
        return self._versioned(graph.replay)

    def _versioned(self, replay):
        """A replay updates the master weights without passing through ``_Optim.step``: bump the groups' version
        counters so that the first eager step afterwards re-packs its fp16 weights."""
        groups = [o.g for o in (getattr(self, n, None) for n in ("opt_enc", "opt_dec", "opt_dis", "opt_wd"))
                  if o is not None]

        def run():
            for h in getattr(self, "_pre_replay", []):
                h()
            replay()
            for g in groups:
                g.version += 1
        return run

    def logs(self):
        v = self.scal.tolist()
        out = {k: v[i] for i, k in enumerate(LOG_KEYS)}
        f = self.flags.tolist()
        out["train_dis"], out["train_dec"] = bool(f[0]), bool(f[1])
        return out


class Stage1Step(_GanStepBase):
    """Stage-I VAE/GAN step (image -> image)."""

    def __init__(self, cfg: ArchConfig, device, hp: Optional[GanHyper] = None, scales: Optional[Scales] = None,
                 distributed: bool = False, sync_bn: bool = True, mode: str = "vae-gan", gate_skip: bool = True):
        """``mode``: the loss composition of train_vgan_stage1.py:359-388 -- 'vae-gan' (default), 'beta-vae' (KL weight
        hp.beta / batch), 'dcgan' (pixel nle, encoder not trained), 'vae' (pixel nle, discriminator not trained unless
        the gate re-arms both).  ``gate_skip``: in ``step`` the weight-gradient GEMMs of the decoder / discriminator are
        conditioned on the equilibrium gate's device flags (fmri_wgrad_if) -- a sub-network the gate does not train in a
        step gets no gradients, as in the reference (`if train_dec: loss_decoder.backward()`,
        train_vgan_stage1.py:420-431); False: they always run and only the update is conditional."""
        if mode not in MODES:
            raise ValueError(f"mode must be one of {sorted(MODES)}")
        self.cfg = cfg
        self.gate_skip = bool(gate_skip)
        self.enc = EncoderNet(cfg, device)
        self.dec = DecoderNet(cfg, device, self.enc.size)
        self.dec.fc_bn.enable_lazy_running()
        self._pre_replay = [self.dec.fc_bn._running_in]      # reloads after an outside write of the buffers only
        self.dis = DiscriminatorNet(cfg, device)
        self._init_common(device, hp, scales, distributed, sync_bn, (self.enc, self.dec, self.dis))
        self.mode = mode
        hp = self.hp
        self.opt_enc = _Optim(self.enc.group, "rmsprop", hp.lr, hp.alpha, hp.eps)
        self.opt_dec = _Optim(self.dec.group, "rmsprop", hp.lr, hp.alpha, hp.eps)
        self.opt_dis = _Optim(self.dis.group, "rmsprop", hp.lr, hp.alpha, hp.eps)
        self.enc_updates = 1                 # encoder passes per batch in the script (BN running-stat updates)
        self.extra_mu_decoder_pass = False   # DualStage1Step (wae_steps.py)

    # ---- parameters -----------------------------------------------------------------------------
    def load_recipe(self, seed: int, perturb: bool = False):
        rs = np.random.RandomState(seed)
        for n in (self.enc, self.dec, self.dis):
            n.group.load_recipe(rs, perturb)

    def state_dict(self):
        sd = {}
        sd.update(self.enc.group.state_dict("encoder."))
        sd.update(self.dec.group.state_dict("decoder."))
        sd.update(self.dis.group.state_dict("discriminator."))
        return sd

    def load_state_dict(self, sd):
        self.enc.group.load_state_dict(sd, "encoder.")
        self.dec.group.load_state_dict(sd, "decoder.")
        self.dis.group.load_state_dict(sd, "discriminator.")

    # ---- the step ---------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, eps: torch.Tensor, z_p: torch.Tensor):
        require_gpu(x)
        cfg = self.cfg
        B, _, H, W = x.shape
        Z, zp = cfg.latent_dim, pad8(cfg.latent_dim)
        dev = x.device
        self.scal.zero_()
        # decoder groups: z (x_tilde), z_p (x_p) and -- Dual step only -- mu (wae_vgan_stage1.py:406, BN statistics)
        G = 3 if self.extra_mu_decoder_pass else 2
        dec_out = torch.empty((1 + G) * B, H, W, 8, dtype=torch.float16, device=dev)
        disc_in = dec_out[:3 * B]
        images_to_nhwc(x, out=disc_in[:B])
        head32, ectx = self.enc.forward(disc_in[:B], updates=self.enc_updates)
        z16 = torch.empty(G * B, zp, dtype=torch.float16, device=dev)
        eps = eps.contiguous().float()
        self._latent(head32, eps, z16[:B], 0, S_KL)
        lib.call("fmri_rows_f32_to_f16", _P(z_p.contiguous().float()), _P(z16[B:]), B, Z, zp, 1.0)
        if G == 3:
            lib.call("fmri_latent_fwd", _P(head32), None, B, Z, zp, _P(z16[2 * B:]), None, None, 0)
        _, dctx = self.dec.forward(z16, G, out=dec_out[B:], zscale=self.zs)
        feat, logit32, sctx = self.dis.forward(disc_in)
        prob, F = self._gan_losses(feat, logit32, B, disc_in[:B], disc_in[B:2 * B], H, W)
        self.fw = dict(B=B, H=H, W=W, F=F, disc_in=disc_in, head32=head32, eps=eps, ectx=ectx, dctx=dctx, sctx=sctx,
                       feat=feat, logit32=logit32, prob=prob)
        return self.fw

    def gate(self, B_global: int):
        self._gate(B_global, self.fw["F"], True)

    def backward(self, extra_dmu: Optional[torch.Tensor] = None, early_apply: bool = False):
        """``extra_dmu`` [B, z] fp32: an additional true-scale cotangent on the encoder means (Dual step).
        ``early_apply`` (one GPU, used by ``step``): the discriminator's and the decoder's optimizer update + weight
        repack are queued on the side stream right behind their last weight gradient, i.e. they run under the backward
        pass of the next sub-network instead of after the whole backward (nothing later in the step reads those weights);
        ``apply`` then only updates the encoder."""
        if self.mode in ("dcgan", "vae"):
            return self._backward_pixel(early_apply, extra_dmu)
        fw, sc, hp, cfg = self.fw, self.sc, self.hp, self.cfg
        B, H, W = fw["B"], fw["H"], fw["W"]
        Z = cfg.latent_dim
        dev = fw["disc_in"].device
        # ``early_apply`` also says that nobody reads reference-layout gradients between this pass and the updates: on one
        # GPU the weight gradients then stay in their GEMM layout until the sub-network's one fmri_apply_batch launch
        fuse = early_apply and self.dd.recorder is None
        gs = self.gate_skip
        ops.begin_grads(self.enc.group, fuse)
        ops.begin_grads(self.dec.group, fuse, gate=self.flags[1:2] if gs else None)
        ops.begin_grads(self.dis.group, fuse, gate=self.flags[0:1] if gs else None)
        dlogit16, dfeat16 = self._start_cotangents(fw["feat"], fw["logit32"], B)
        # weight gradients run on the side stream (ops.side_run) and are joined once, at the end of the backward pass, so
        # that a sub-network's last weight gradients overlap the next one's backward
        dp = self.dd.on
        # ``early``: a sub-network's gradient reduction (data parallel), optimizer update and weight repack are queued on the
        # SIDE stream right behind its last weight gradient -- the main stream neither joins the side stream nor waits for
        # the collective before it goes on with the next sub-network's backward pass.  (Recording into graph segments
        # keeps everything on one stream: the round-3 order with asynchronous collectives joined at the end.)
        early = early_apply and ops._SIDE["on"] and self.dd.recorder is None
        self._applied_early = early
        dimg_a, dimg_b = self.dis.backward(fw["sctx"], dlogit16, sc.a, dfeat16, sc.b, True, slice(B, 3 * B),
                                           join=dp and not early)
        if early:
            ops.side_run(dev, lambda: self._reduce_apply(self.opt_dis, self.dis, self.flags[0:1], S_NA))
        else:
            self._reduce_async(self.dis.group)
        # decoder cotangent, stored = dec * nA * (lambda*B_true - (1-lambda)*A_true); the weights lambda*nA/nB and
        # 1-lambda are device scalars written by the gate kernel (a recorded step follows the lambda schedule)
        cot = torch.empty(3 * B, H, W, 8, dtype=torch.float16, device=dev)
        # block order [x_tilde: feature loss (encoder path) | x_tilde: decoder loss | x_p: decoder loss]: the two blocks
        # that go through the SAME forward activations (group 0) are adjacent, so the decoder's BatchNorm backward takes
        # them in one pass (BatchNorm.backward2), and the two training blocks are adjacent, so every decoder weight
        # gradient is one launch over 2B rows
        cot[:B].copy_(dimg_b[:B])
        axpby(dimg_b, dimg_a, sc.dec / sc.b, -sc.dec / sc.a, out=cot[B:], a_dev=self._slot(S_C1),
              b_dev=self._slot(S_C2))
        entries = [dict(g=0, scale=sc.b, train=False, need_dz=True), dict(g=0, scale=sc.dec, train=True),
                   dict(g=1, scale=sc.dec, train=True)]
        dz = self.dec.backward(fw["dctx"], cot, entries, join=dp and not early)[0]  # = nB * dz_true
        if early:
            ops.side_run(dev, lambda: self._reduce_apply(self.opt_dec, self.dec, self.flags[1:2], S_GDEC))
        else:
            self._reduce_async(self.dec.group)
        dhead32 = torch.empty(B, 2 * Z, dtype=torch.float32, device=dev)
        # KL weight: 1, or beta / batch for 'beta-vae' (train_vgan_stage1.py:360-362).  The gate kernel writes it to
        # the device slot S_KLW from the device-resident hyper-parameters, so that a recorded (HIP-graph) step follows
        # set_hyper(beta=...) in the encoder GRADIENT as well as in the logged loss: weight = S_KLW * nB on the device
        kl_dev = self._slot(S_NB)
        if self.mode == "beta-vae":
            kl_dev = torch.mul(self._slot(S_KLW), self._slot(S_NB))
        lib.call("fmri_latent_bwd", _P(fw["head32"]), _P(fw["eps"]), _P(dz), Z, 1.0, 1.0, _P(kl_dev), B, Z,
                 1.0, None, _P(dhead32), 1)                              # = nB * dhead_true
        if extra_dmu is not None:
            dhead32[:, :Z].addcmul_(extra_dmu, self._slot(S_NB))            # carried at the same device factor nB
        dhead16 = self._renorm(dhead32, sc.enc, self._slot(S_NB), B * self.dd.world)   # S_NE = nB * nE
        eg = self.enc.group
        tail = eg.offsets["fc.0.weight"]
        if early and dp:
            # the fc.0 ... l_var tail (93 % of the buffer) is reduced on the side stream as soon as the fc weight gradient
            # is final there, under the conv backward; the conv head right behind the last weight gradient
            def reduce_part(part):                  # (side stream) the weight gradients queued so far -> reference layout
                ops.materialize_grads(eg)
                self.dd.all_reduce(part)
            self.enc.backward(fw["ectx"], dhead16, sc.enc, join=False, after_fc_join=False,
                              after_fc=lambda: ops.side_run(dev, lambda: reduce_part(eg.grad[tail:])))
            ops.side_run(dev, lambda: reduce_part(eg.grad[:tail]))
            ops.join_side()
            return
        self.enc.backward(fw["ectx"], dhead16, sc.enc,                  # grads = S_NE * true
                          after_fc=(lambda: self._reduce_async(eg, eg.grad[tail:])) if dp else None)
        if dp:
            ops.materialize_grads(eg)
        self.dd.all_reduce(eg.grad[:tail])
        self.dd.wait_all()

    def _backward_pixel(self, early_apply: bool, extra_dmu: Optional[torch.Tensor] = None):
        """Modes 'dcgan' and 'vae' (train_vgan_stage1.py:374-388): the reconstruction term is the pixel nle.
        ``extra_dmu``: see ``backward`` (mode 'vae' of the Dual step; 'dcgan' does not train the encoder).

        dcgan: decoder <- lambda*d nle - (1-lambda)*d(bce_orig + bce_sampled) on [x_tilde ; x_p], discriminator <-
               d(bce_orig + bce_sampled), encoder not trained.
        vae  : encoder <- d(KL + nle) through the decoder, decoder <- lambda*d nle, discriminator <- d(bce_orig +
               bce_sampled), applied only if the gate re-armed it (flags[0])."""
        fw, sc, cfg = self.fw, self.sc, self.cfg
        B, H, W = fw["B"], fw["H"], fw["W"]
        Z = cfg.latent_dim
        d_in = fw["disc_in"]
        dev = d_in.device
        dp = self.dd.on
        self._applied_early = False
        for n in (self.enc, self.dec, self.dis):
            n.group.zero_grad()
        dlogit16 = torch.empty(3 * B, 8, dtype=torch.float16, device=dev)
        lib.call("fmri_gan_head_bwd_parts", _P(fw["logit32"]), 1, B, _P(dlogit16), 8, sc.a, _P(self._slot(S_NA)),
                 self._dis_parts())
        x16, xt16 = d_in[:B], d_in[B:2 * B]
        if self.mode == "dcgan":
            dimg_a, _ = self.dis.backward(fw["sctx"], dlogit16, sc.a, None, sc.b, True, slice(B, 3 * B), join=dp)
            self.dd.all_reduce_async(self.dis.group.grad)
            # stored = dec * nA * (lambda * d nle - (1-lambda) * A), d nle / d x_tilde = x_tilde - x
            dnle = axpby(xt16, x16, 1.0, -1.0)
            cot = torch.empty(2 * B, H, W, 8, dtype=torch.float16, device=dev)
            axpby(dnle, dimg_a[:B], sc.dec, -sc.dec / sc.a, out=cot[:B], a_dev=self._slot(S_C3),
                  b_dev=self._slot(S_C2))
            axpby(dimg_a[B:], None, -sc.dec / sc.a, 0.0, out=cot[B:], a_dev=self._slot(S_C2))
            entries = [dict(g=0, scale=sc.dec, train=True), dict(g=1, scale=sc.dec, train=True)]
            self.dec.backward(fw["dctx"], cot, entries, join=dp)
            self.dd.all_reduce_async(self.dec.group.grad)
            self.dd.wait_all()
            return
        # vae
        self.dis.backward(fw["sctx"], dlogit16, sc.a, None, sc.b, True, None, join=dp)
        self.dd.all_reduce_async(self.dis.group.grad)
        # stored = p * nP * (x_tilde - x); the decoder gradients carry nP and are scaled by lambda in the optimizer
        cot = axpby(xt16, x16, sc.p, -sc.p, a_dev=self._slot(S_NP), b_dev=self._slot(S_NP))
        entries = [dict(g=0, scale=sc.p, train=True, need_dz=True)]
        dz = self.dec.backward(fw["dctx"], cot, entries, join=dp)[0]        # = nP * dz_true
        self.dd.all_reduce_async(self.dec.group.grad)
        dhead32 = torch.empty(B, 2 * Z, dtype=torch.float32, device=dev)
        lib.call("fmri_latent_bwd", _P(fw["head32"]), _P(fw["eps"]), _P(dz), Z, 1.0, 1.0, _P(self._slot(S_NP)), B, Z,
                 1.0, None, _P(dhead32), 1)                                  # = nP * dhead_true
        if extra_dmu is not None:
            dhead32[:, :Z].addcmul_(extra_dmu, self._slot(S_NP))                # carried at the same device factor nP
        dhead16 = self._renorm(dhead32, sc.enc, self._slot(S_NP), B * self.dd.world)    # S_NE = nP * nE
        eg = self.enc.group
        tail = eg.offsets["fc.0.weight"]
        self.enc.backward(fw["ectx"], dhead16, sc.enc,
                          after_fc=(lambda: self.dd.all_reduce_async(eg.grad[tail:])) if dp else None)
        self.dd.all_reduce(eg.grad[:tail])
        self.dd.wait_all()

    def _reduce_apply(self, opt, net, flag, slot):
        """(current stream = the side stream) SUM all-reduce of the sub-network's gradient buffer over the ranks -- nothing
        on one GPU --, then its optimizer update and the refresh of everything derived from its weights."""
        if self.dd.on:
            ops.materialize_grads(net.group)         # deferred weight gradients -> reference layout, one launch
        self.dd.all_reduce(net.group.grad)
        self._apply_one(opt, net, flag, slot)

    def _apply_one(self, opt, net, flag, slot):
        """Optimizer update of one sub-network + refresh of everything derived from its weights (current stream)."""
        opt.step(flag, gdev=self._slot(slot))
        refresh_net(net)

    def apply(self):
        if self.mode != "dcgan":                         # 'dcgan': train_enc = False (train_vgan_stage1.py:376)
            self.opt_enc.step(None, gdev=self._slot(S_NE))
        if getattr(self, "_applied_early", False):
            self._applied_early = False
            return
        self.opt_dec.step(self.flags[1:2], gdev=self._slot(S_GDEC))
        self.opt_dis.step(self.flags[0:1], gdev=self._slot(S_NA))

    def capture_forward(self, x, eps, z_p, warmup: int = 2):
        """Hybrid launch mode (one GPU): the forward pass + gate -- a dependent chain with nothing to overlap -- is
        recorded into a HIP graph, the backward pass and the updates stay eagerly issued launches on two streams
        (``ops.side_run``).  Halves the Python work per step, which is what decides whether a slow host can keep the
        two-stream backward fed.  Also available to data-parallel runs with per-rank BN statistics (no collective in
        the forward pass).  Returns a zero-argument callable running one full step on the static inputs."""
        if self.dd.on and self.dd.sync_bn:
            raise RuntimeError("capture_forward: the forward pass holds SyncBN collectives (use capture())")
        # data parallel with per-rank BN statistics: the forward pass's only collective is the sum of the loss scalars at
        # its very end; that all-reduce, the gate and the backward pass with its gradient reductions stay eager
        gate_in_graph = not self.dd.on
        nets = (self.enc, self.dec, self.dis)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self.step(x, eps, z_p)
            ops.join_side()
            for n in nets:
                refresh_net(n)               # so that the recorded forward contains no refresh launches
        torch.cuda.current_stream().wait_stream(side)
        B = x.shape[0]
        graph = torch.cuda.CUDAGraph()
        # thread_local: the communication backend's watchdog thread may touch the device while this thread records
        self._defer_loss_reduce = not gate_in_graph      # the forward's only collective (sum of the loss scalars)
        try:
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                self.forward(x, eps, z_p)
                if gate_in_graph:
                    self.gate(B)
        finally:
            self._defer_loss_reduce = False

        fw_captured = self.fw                 # the tensors the recorded forward writes

        def run():
            for n in nets:
                refresh_net(n)               # no-ops for the sub-networks the last backward refreshed early
            graph.replay()
            self.fw = fw_captured            # an eager step() / forward() in between rebinds self.fw to its own batch
            if not gate_in_graph:
                self.dd.all_reduce(self.scal[:N_REDUCED])
                self.gate(B * self.dd.world)
            self.backward(early_apply=True)
            self.apply()
            return self.scal
        self._fwd_graph = graph
        return run

    def step(self, x, eps, z_p):
        """One full training step; returns the device scalar block (see LOG_KEYS) without syncing."""
        fw = self.forward(x, eps, z_p)
        self.gate(fw["B"] * self.dd.world)
        self.backward(early_apply=True)
        self.apply()
        return self.scal

    # ---- reference-shaped views of the last forward (API / parity tests) -----------------------------
    def outputs(self):
        fw, cfg = self.fw, self.cfg
        B, Z = fw["B"], cfg.latent_dim
        d = fw["disc_in"]
        feat = fw["feat"]
        n3, h, w, c = feat.shape
        return dict(
            x_tilde=nhwc_to_images(d[B:2 * B], 3), x_p=nhwc_to_images(d[2 * B:], 3),
            disc_class=fw["prob"].reshape(3 * B, 1).clone(),
            disc_layer=nhwc_to_images(feat, c).reshape(n3, -1),
            mus=fw["head32"][:, :Z].clone(), log_variances=fw["head32"][:, Z:].clone())

    def named_grads(self):
        """True-scale gradients (the device normalisation factors divided out) -- syncs; tests/API only.  After
        ``backward()``; ``step()`` does not leave any (FlatGroup.check_grads_readable)."""
        s = self.scal.tolist()
        out = {}
        for pre, n, f in (("encoder.", self.enc, s[S_NE]), ("decoder.", self.dec, s[S_GDEC]),
                          ("discriminator.", self.dis, s[S_NA])):
            n.group.check_grads_readable()
            for k, v in n.group.grads.items():
                out[pre + k] = v / f
        return out


class CognitiveStep(_GanStepBase):
    """Stage-II / Stage-III steps of the Dual-VAE/GAN (fMRI -> image), `VaeGanCognitive` wiring
    (models/vae_gan.py:352-395) + the loop bodies train/train_vgan_stage2.py:321-407 (stage=2: decoder
    frozen, teacher distillation, encoder + discriminator trained, no gate, grads clamped to +-1) and
    train/train_vgan_stage3.py:324-411 (stage=3: cognitive encoder frozen, decoder + discriminator trained,
    gate on, clamp +-1)."""

    def __init__(self, cfg: ArchConfig, n_voxels: int, device, stage: int, hp: Optional[GanHyper] = None,
                 scales: Optional[Scales] = None, distributed: bool = False, sync_bn: bool = True,
                 gate_skip: bool = True, mode: str = "vae-gan"):
        """``mode``: 'vae-gan' (default) or 'vae' -- the scripts' `--mode vae` (train_vgan_stage2.py:234-238,362-366;
        train_vgan_stage3.py:370-374): no teacher net (the discriminator's "real" slot is the ground-truth image), the
        reconstruction term is the PIXEL nle instead of the feature mse, the discriminator loss bce_orig + bce_sampled.
        Stage II: encoder <- d(KL + nle) through the frozen decoder, discriminator trained (the script's train_dis = False
        is overwritten two lines later).  Stage III: decoder <- lambda * d nle, discriminator updated only in a step whose
        gate re-arms both."""
        assert stage in (2, 3)
        if mode not in ("vae-gan", "vae"):
            raise ValueError("mode must be 'vae-gan' or 'vae'")
        self.cfg, self.stage, self.n_voxels = cfg, stage, n_voxels
        self.gate_skip = bool(gate_skip)
        self.cog = CognitiveEncoderNet(cfg, n_voxels, device)
        self.dec = DecoderNet(cfg, device, cfg.encoder_channels[2])
        self.dec.fc_bn.enable_lazy_running()
        self._pre_replay = [self.dec.fc_bn._running_in]      # reloads after an outside write of the buffers only
        self.dis = DiscriminatorNet(cfg, device)
        self.teacher_enc = EncoderNet(cfg, device) if (stage == 2 and mode != "vae") else None
        nets = [self.cog, self.dec, self.dis] + ([self.teacher_enc] if self.teacher_enc is not None else [])
        self._init_common(device, hp, scales, distributed, sync_bn, nets)
        self.mode = mode
        hp = self.hp
        self.opt_enc = _Optim(self.cog.group, "rmsprop", hp.lr, hp.alpha, hp.eps)
        self.opt_dec = _Optim(self.dec.group, "rmsprop", hp.lr, hp.alpha, hp.eps)
        self.opt_dis = _Optim(self.dis.group, "rmsprop", hp.lr, hp.alpha, hp.eps)

    def load_recipe(self, seed: int, perturb: bool = False):
        """Teacher VaeGan weights from seed, cognitive encoder from seed+100 (the golden-fixture recipe)."""
        rs = np.random.RandomState(seed)
        enc_tmp = self.teacher_enc if self.teacher_enc is not None else EncoderNet(self.cfg, self.device)
        for n in (enc_tmp, self.dec, self.dis):
            n.group.load_recipe(rs, perturb)
        self.cog.group.load_recipe(np.random.RandomState(seed + 100), perturb)

    def state_dict(self):
        sd = {}
        sd.update(self.cog.group.state_dict("encoder."))
        sd.update(self.dec.group.state_dict("decoder."))
        sd.update(self.dis.group.state_dict("discriminator."))
        if self.teacher_enc is not None:
            sd.update(self.teacher_enc.group.state_dict("teacher_net.encoder."))
            sd.update(self.dec.group.state_dict("teacher_net.decoder."))
            sd.update(self.dis.group.state_dict("teacher_net.discriminator."))
        return sd

    def load_state_dict(self, sd):
        """A Stage-II / Stage-III checkpoint as the scripts write it (``model.state_dict()`` of ``VaeGanCognitive``):
        ``encoder.`` = cognitive encoder, ``decoder.``, ``discriminator.`` and, for stage 2, ``teacher_net.encoder.``
        (train/train_vgan_stage3.py:241 loads the Stage-II file this way; its ``teacher_net.decoder./discriminator.``
        entries alias ``decoder.`` / ``discriminator.`` in Stage II and are not needed in Stage III)."""
        self.cog.group.load_state_dict(sd, "encoder.")
        self.dec.group.load_state_dict(sd, "decoder.")
        self.dis.group.load_state_dict(sd, "discriminator.")
        if self.teacher_enc is not None:
            self.teacher_enc.group.load_state_dict(sd, "teacher_net.encoder.")

    def load_teacher(self, sd):
        """A Stage-I ``VaeGan`` checkpoint as the teacher (train/train_vgan_stage2.py:212-217,230): its decoder and
        discriminator become the model's own (frozen decoder, trained discriminator), its encoder the teacher encoder."""
        if self.teacher_enc is not None:
            self.teacher_enc.group.load_state_dict(sd, "encoder.")
        self.dec.group.load_state_dict(sd, "decoder.")
        self.dis.group.load_state_dict(sd, "discriminator.")

    def forward(self, fmri: torch.Tensor, image: torch.Tensor, eps: torch.Tensor, z_p: torch.Tensor,
                eps_teacher: Optional[torch.Tensor] = None):
        require_gpu(fmri)
        cfg = self.cfg
        B, _, H, W = image.shape
        Z, zp = cfg.latent_dim, pad8(cfg.latent_dim)
        dev = image.device
        self.scal.zero_()
        disc_in = torch.empty(3 * B, H, W, 8, dtype=torch.float16, device=dev)
        fmri16 = rows_to_f16(fmri)
        head32, cctx = self.cog.forward(fmri16)
        eps = eps.contiguous().float()
        if self.teacher_enc is not None:
            # decoder groups in disc_in row order: 0 = teacher reconstruction ("real"), 1 = x_tilde, 2 = x_p
            z16 = torch.empty(3 * B, zp, dtype=torch.float16, device=dev)
            img16 = images_to_nhwc(image)
            head_t, _ = self.teacher_enc.forward(img16)
            self._latent(head_t, eps_teacher.contiguous().float(), z16[:B], 0, None)
            self._latent(head32, eps, z16[B:2 * B], 1, S_KL)
            lib.call("fmri_rows_f32_to_f16", _P(z_p.contiguous().float()), _P(z16[2 * B:]), B, Z, zp, 1.0)
            # reference call order of the decoder: x_tilde, teacher reconstruction, x_p (vae_gan.py:365,377,390)
            _, dctx = self.dec.forward(z16, 3, out=disc_in, stat_order=(1, 0, 2), zscale=self.zs)
            g_tilde, g_p = 1, 2
        else:
            z16 = torch.empty(2 * B, zp, dtype=torch.float16, device=dev)
            images_to_nhwc(image, out=disc_in[:B])
            self._latent(head32, eps, z16[:B], 0, S_KL)
            lib.call("fmri_rows_f32_to_f16", _P(z_p.contiguous().float()), _P(z16[B:]), B, Z, zp, 1.0)
            _, dctx = self.dec.forward(z16, 2, out=disc_in[B:], zscale=self.zs)
            g_tilde, g_p = 0, 1
        feat, logit32, sctx = self.dis.forward(disc_in)
        prob, F = self._gan_losses(feat, logit32, B, disc_in[:B], disc_in[B:2 * B], H, W)
        self.fw = dict(B=B, H=H, W=W, F=F, disc_in=disc_in, head32=head32, eps=eps, cctx=cctx, dctx=dctx, sctx=sctx,
                       feat=feat, logit32=logit32, prob=prob, g_tilde=g_tilde, g_p=g_p)
        return self.fw

    def gate(self, B_global: int):
        if self.stage == 2:
            self._gate(B_global, self.fw["F"], False, 1, 0)          # train_dis = True, train_dec = False
        else:
            self._gate(B_global, self.fw["F"], True)

    def backward(self, fuse: bool = False):
        """``fuse`` (used by ``step``): the updates follow right behind and nobody reads reference-layout gradients in
        between -- weight gradients stay in their GEMM layout until the sub-network's one fmri_apply_batch launch
        (ops.begin_grads)."""
        fw, sc, hp, cfg = self.fw, self.sc, self.hp, self.cfg
        B, H, W, Z = fw["B"], fw["H"], fw["W"], cfg.latent_dim
        dev = fw["disc_in"].device
        fuse = fuse and self.dd.recorder is None
        if self.mode == "vae":
            return self._backward_pixel(fuse)
        dlogit16, dfeat16 = self._start_cotangents(fw["feat"], fw["logit32"], B)
        gs = self.gate_skip                  # (see Stage1Step: no gradients for a sub-network the gate does not train)
        if self.stage == 2:
            ops.begin_grads(self.cog.group, fuse)
            ops.begin_grads(self.dis.group, fuse, gate=self.flags[0:1] if gs else None)
            _, dimg_b = self.dis.backward(fw["sctx"], dlogit16, sc.a, dfeat16, sc.b, True, slice(B, 2 * B),
                                          img_streams=(False, True))
            self._reduce_async(self.dis.group)                  # under the decoder / cognitive-encoder backward
            entries = [dict(g=fw["g_tilde"], scale=sc.b, train=False, need_dz=True)]
            dz = self.dec.backward(fw["dctx"], dimg_b, entries)[0]
            dhead32 = torch.empty(B, 2 * Z, dtype=torch.float32, device=dev)
            lib.call("fmri_latent_bwd", _P(fw["head32"]), _P(fw["eps"]), _P(dz), Z, 1.0, 1.0, _P(self._slot(S_NB)), B,
                     Z, 1.0, None, _P(dhead32), 1)
            dhead16 = self._renorm(dhead32, sc.enc, self._slot(S_NB), B * self.dd.world)
            self.cog.backward(fw["cctx"], dhead16, sc.enc)
            self._reduce_async(self.cog.group)
            self.dd.wait_all()
        else:
            ops.begin_grads(self.dec.group, fuse, gate=self.flags[1:2] if gs else None)
            ops.begin_grads(self.dis.group, fuse, gate=self.flags[0:1] if gs else None)
            dimg_a, dimg_b = self.dis.backward(fw["sctx"], dlogit16, sc.a, dfeat16, sc.b, True, slice(B, 3 * B))
            self._reduce_async(self.dis.group)                  # under the decoder backward
            cot = axpby(dimg_b, dimg_a, sc.dec / sc.b, -sc.dec / sc.a, a_dev=self._slot(S_C1), b_dev=self._slot(S_C2))
            entries = [dict(g=0, scale=sc.dec, train=True), dict(g=1, scale=sc.dec, train=True)]
            self.dec.backward(fw["dctx"], cot, entries)
            self._reduce_async(self.dec.group)
            self.dd.wait_all()

    def _backward_pixel(self, fuse: bool):
        """mode 'vae' (train_vgan_stage2.py:362-366, train_vgan_stage3.py:370-374): the reconstruction term is the pixel
        nle, d nle / d x_tilde = x_tilde - x_gt; the discriminator's loss is bce_orig + bce_sampled (no image gradient
        is needed: nothing upstream of it trains on a discriminator term)."""
        fw, sc, cfg = self.fw, self.sc, self.cfg
        B, H, W, Z = fw["B"], fw["H"], fw["W"], cfg.latent_dim
        d_in = fw["disc_in"]
        dev = d_in.device
        gs = self.gate_skip
        dlogit16 = torch.empty(3 * B, 8, dtype=torch.float16, device=dev)
        lib.call("fmri_gan_head_bwd_parts", _P(fw["logit32"]), 1, B, _P(dlogit16), 8, sc.a, _P(self._slot(S_NA)),
                 self._dis_parts())
        x16, xt16 = d_in[:B], d_in[B:2 * B]
        if self.stage == 2:
            ops.begin_grads(self.cog.group, fuse)
            ops.begin_grads(self.dis.group, fuse, gate=self.flags[0:1] if gs else None)
        else:
            ops.begin_grads(self.dec.group, fuse, gate=self.flags[1:2] if gs else None)
            ops.begin_grads(self.dis.group, fuse, gate=self.flags[0:1] if gs else None)
        self.dis.backward(fw["sctx"], dlogit16, sc.a, None, sc.b, True, None)
        self._reduce_async(self.dis.group)
        # stored = p * nP * (x_tilde - x_gt)
        cot = axpby(xt16, x16, sc.p, -sc.p, a_dev=self._slot(S_NP), b_dev=self._slot(S_NP))
        if self.stage == 2:
            entries = [dict(g=fw["g_tilde"], scale=sc.p, train=False, need_dz=True)]
            dz = self.dec.backward(fw["dctx"], cot, entries)[0]                  # = nP * dz_true
            dhead32 = torch.empty(B, 2 * Z, dtype=torch.float32, device=dev)
            lib.call("fmri_latent_bwd", _P(fw["head32"]), _P(fw["eps"]), _P(dz), Z, 1.0, 1.0, _P(self._slot(S_NP)), B,
                     Z, 1.0, None, _P(dhead32), 1)
            dhead16 = self._renorm(dhead32, sc.enc, self._slot(S_NP), B * self.dd.world)   # S_NE = nP * nE
            self.cog.backward(fw["cctx"], dhead16, sc.enc)
            self._reduce_async(self.cog.group)
        else:
            # the decoder gradients carry nP; the optimizer divides by S_GDEC = nP / lambda (loss_decoder = lambda * nle)
            entries = [dict(g=fw["g_tilde"], scale=sc.p, train=True)]
            self.dec.backward(fw["dctx"], cot, entries)
            self._reduce_async(self.dec.group)
        self.dd.wait_all()

    def apply(self):
        if self.stage == 2:
            self.opt_enc.step(None, clamp=1.0, gdev=self._slot(S_NE))
            self.opt_dis.step(self.flags[0:1], clamp=1.0, gdev=self._slot(S_NA))
        else:
            # (S_GDEC: nA in mode 'vae-gan', nP / lambda in mode 'vae' -- the gate kernel writes it)
            self.opt_dec.step(self.flags[1:2], clamp=1.0, gdev=self._slot(S_GDEC))
            self.opt_dis.step(self.flags[0:1], clamp=1.0, gdev=self._slot(S_NA))

    def step(self, fmri, image, eps, z_p, eps_teacher=None):
        fw = self.forward(fmri, image, eps, z_p, eps_teacher)
        self.gate(fw["B"] * self.dd.world)
        self.backward(fuse=True)
        self.apply()
        return self.scal

    def outputs(self):
        fw, cfg = self.fw, self.cfg
        B, Z = fw["B"], cfg.latent_dim
        d = fw["disc_in"]
        feat = fw["feat"]
        n3, h, w, c = feat.shape
        return dict(gt_x=nhwc_to_images(d[:B], 3), x_tilde=nhwc_to_images(d[B:2 * B], 3),
                    x_p=nhwc_to_images(d[2 * B:], 3), disc_class=fw["prob"].reshape(3 * B, 1).clone(),
                    disc_layer=nhwc_to_images(feat, c).reshape(n3, -1),
                    mus=fw["head32"][:, :Z].clone(), log_variances=fw["head32"][:, Z:].clone())

    def named_grads(self):
        s = self.scal.tolist()
        groups = ((("encoder.", self.cog, s[S_NE]), ("discriminator.", self.dis, s[S_NA])) if self.stage == 2 else
                  (("decoder.", self.dec, s[S_GDEC]), ("discriminator.", self.dis, s[S_NA])))
        out = {}
        for pre, n, f in groups:
            n.group.check_grads_readable()
            for k, v in n.group.grads.items():
                out[pre + k] = v / f
        return out
