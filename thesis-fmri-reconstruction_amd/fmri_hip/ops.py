"""Layer-level operators of the engine: thin Python objects that own packed fp16 weights and enqueue
the HIP kernels of libfmri_hip.so (C ABI in include/fmri_hip.h) on torch's current stream.

Activations are fp16 NHWC tensors ``[N, H, W, Cp]`` (Cp = channels padded to 8; dense activations are
``[M, Kp]``).  Cotangents are fp16 too and carry an explicit python-float ``scale`` (stored = true*scale)
chosen by the step code so that they stay inside fp16's normal range; weight gradients are unscaled when
they are unpacked into the fp32 ``.grad`` buffers.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

import ctypes
import os

import torch

from . import lib

ACT_NONE, ACT_RELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3
MODE_CONV, MODE_TCONV2, MODE_CONV_FLIP = 0, 1, 2

_P = lib.ptr


def pad8(c: int) -> int:
    return (c + 7) // 8 * 8


def ceil_to(v: int, m: int) -> int:
    return (v + m - 1) // m * m


def tile_for(c: int) -> int:
    return 128 if c >= 128 else (64 if c >= 64 else 32)


_ZERO = {}


def zero_page(device) -> torch.Tensor:
    key = str(device)
    if key not in _ZERO:
        _ZERO[key] = torch.zeros(256, dtype=torch.float16, device=device)
    return _ZERO[key]


def require_gpu(t: torch.Tensor):
    if not t.is_cuda:
        raise RuntimeError("fmri_hip: the HIP engine needs tensors on an MI355X device (no CPU fallback)")
    # every launch goes to the CURRENT device's current stream (lib.stream): a tensor of another device would be
    # addressed from the wrong GPU and lose its stream ordering
    if t.device.index != torch.cuda.current_device():
        raise RuntimeError(f"fmri_hip: tensor on {t.device} but the current device is cuda:{torch.cuda.current_device()}"
                           " -- call torch.cuda.set_device() (or use `with torch.cuda.device(...)`) first")


# ------------------------------------------------------------------------------------------------
# packed weights
# ------------------------------------------------------------------------------------------------
@dataclass
class PackSpec:
    """dst[(ta*A+a)][(tb*Bp+b)] = src[a*sa + ta*sta + b*sb + t(tb)*stb] (see csrc/layout.hip)."""
    sa: int
    sta: int
    A: int
    TA: int
    sb: int
    stb: int
    B: int
    KW: int = 1
    py: int = 0
    px: int = 0
    step: int = 1
    TH: int = 1
    TW: int = 1

    @property
    def rows(self):
        return self.TA * self.A

    @property
    def kcols(self):
        return self.TH * self.TW * pad8(self.B)


class PackedWeight:
    """fp16 GEMM-layout copy of one fp32 master weight in one orientation (1 or 4 class blocks)."""

    def __init__(self, master: torch.Tensor, group, specs: List[PackSpec], rows_pad: int, kpads: List[int],
                 offsets: List[int]):
        self.master, self.group = master, group
        self.specs, self.rows_pad, self.kpads, self.offsets = specs, rows_pad, kpads, offsets
        total = offsets[-1] + rows_pad * kpads[-1]
        # zero once: the pack kernels' fast paths only rewrite the valid region, padding stays zero
        self.buf = torch.zeros(total, dtype=torch.float16, device=master.device)
        self.version = -1
        if not hasattr(group, "packed"):
            group.packed = []
        group.packed.append(self)

    def _items(self):
        for sp, kp, off in zip(self.specs, self.kpads, self.offsets):
            yield (_P(self.master), self.buf.data_ptr() + 2 * off, sp.sa, sp.sta, sp.sb, sp.stb, sp.A, sp.TA, sp.B, sp.KW,
                   sp.py, sp.px, sp.step, sp.TH, sp.TW, self.rows_pad, kp)

    def get(self) -> torch.Tensor:
        if self.version != self.group.version:
            _repack_group(self.group)
        return self.buf


_TRANSPOSE_ON = os.environ.get("FMRI_PACK_TRANSPOSE") != "off"


def _repack_group(group, skip=()):
    """Refresh every fp16 GEMM copy of a sub-network's weights after its master buffer changed: one batched launch
    (fmri_pack_weight_batch, device-resident table built once) + the few weights that need another pack path.
    ``skip``: PackedWeights that are already current (written by fmri_apply_batch together with the update)."""
    packed = group.packed
    tabs = group.__dict__.setdefault("_pack_tables", {})
    key = (len(packed),) + tuple(sorted(id(pw) for pw in skip))
    tab = tabs.get(key)
    if tab is None:
        L = lib.load()
        nbytes = L.fmri_pack_entry_bytes()
        rows, singles, tiles = [], [], 0
        skipped = set(id(pw) for pw in skip)
        transposes = []                  # (src ptr, dst ptr, R, C, source rows, readable width, ld_src, ld_dst)
        for pw in packed:
            if id(pw) in skipped:
                continue
            src = getattr(pw, "transpose_of", None)
            if src is not None and _TRANSPOSE_ON:
                # a dense layer's second orientation: the transpose of its first fp16 copy
                R, Cc = src.specs[0].rows, pw.specs[0].rows
                transposes.append((_P(src.buf), _P(pw.buf), R, Cc, src.rows_pad, src.kpads[0], src.kpads[0], pw.kpads[0]))
                continue
            src = getattr(pw, "taps_of", None)
            if src is not None and _TRANSPOSE_ON:
                # the four parity classes of a stride-2 transposed convolution: tap t' of a class block is the transpose
                # of tap t = (py + 2 ty) * k + (px + 2 tx) of the single-block copy [rows][t * Bp + b]
                ssp = src.specs[0]
                sbp, sbase = pad8(ssp.B), _P(src.buf) + 2 * src.offsets[0]
                for sp, kp, off in zip(pw.specs, pw.kpads, pw.offsets):
                    assert sp.B == ssp.rows and sp.rows == ssp.B, "class block and single block disagree"
                    for ty in range(sp.TH):
                        for tx in range(sp.TW):
                            t = (sp.py + sp.step * ty) * sp.KW + (sp.px + sp.step * tx)
                            transposes.append((sbase + 2 * t * sbp, _P(pw.buf) + 2 * (off + (ty * sp.TW + tx) * pad8(sp.B)),
                                               ssp.rows, sp.rows, src.rows_pad, src.kpads[0] - t * sbp, src.kpads[0], kp))
                continue
            for item in pw._items():
                host = ctypes.create_string_buffer(nbytes)
                n = L.fmri_pack_entry_fill(host, *item, tiles)
                if n < 0:
                    lib.check(n, "fmri_pack_entry_fill")
                if n > 0:
                    rows.append(host.raw)
                    tiles += n
                else:
                    singles.append(item)
        dev_tab = None
        if rows:
            dev_tab = torch.frombuffer(bytearray(b"".join(rows)), dtype=torch.uint8).to(group.device)
        t_tab, t_tiles = None, 0
        if transposes:
            tb, trows = L.fmri_transpose_entry_bytes(), []
            for item in transposes:
                host = ctypes.create_string_buffer(tb)
                n = L.fmri_transpose_entry_fill(host, *item, t_tiles)
                if n <= 0:
                    lib.check(n if n < 0 else -1, "fmri_transpose_entry_fill")
                trows.append(host.raw)
                t_tiles += n
            t_tab = torch.frombuffer(bytearray(b"".join(trows)), dtype=torch.uint8).to(group.device)
        tab = tabs[key] = dict(table=dev_tab, n=len(rows), tiles=tiles, singles=singles, t_table=t_tab,
                               t_n=len(transposes), t_tiles=t_tiles)
    if tab["n"]:
        lib.call("fmri_pack_weight_batch", _P(tab["table"]), tab["n"], tab["tiles"])
    for item in tab["singles"]:
        lib.call("fmri_pack_weight", *item)
    if tab["t_n"]:                                   # (after the packs: the transposes' sources are current now)
        lib.call("fmri_transpose_f16_batch", _P(tab["t_table"]), tab["t_n"], tab["t_tiles"])
    for pw in packed:
        pw.version = group.version


def repack_group(group):
    """Refresh the fp16 GEMM copies of ``group`` now (they are otherwise refreshed lazily at their next use)."""
    if group.packed and any(pw.version != group.version for pw in group.packed):
        _repack_group(group)


def _single(master, group, sp: PackSpec, tile: int) -> PackedWeight:
    rows_pad = ceil_to(sp.rows, tile)
    return PackedWeight(master, group, [sp], rows_pad, [ceil_to(sp.kcols, 64)], [0])


def _tconv(master, group, k, pad, sa, A, sb, B, tile) -> PackedWeight:
    """4 parity-class blocks for a stride-2 transposed convolution producing A channels from B channels."""
    rows_pad = ceil_to(A, tile)
    specs, kpads, offs = [], [], []
    for cy in range(2):
        for cx in range(2):
            g = lib.tconv_class(k, pad, cy, cx, pad8(B), rows_pad)
            specs.append(PackSpec(sa=sa, sta=0, A=A, TA=1, sb=sb, stb=1, B=B, KW=k, py=g["py"], px=g["px"], step=2,
                                  TH=g["th"], TW=g["tw"]))
            kpads.append(g["kpad"])
            offs.append(g["w_off"])
    return PackedWeight(master, group, specs, rows_pad, kpads, offs)


def _choose_splits(blocks: int, ksteps: int) -> int:
    if blocks >= 192 or ksteps < 8:
        return 1
    s = min(ksteps // 4, (512 + blocks - 1) // blocks)
    s = max(s, 1)
    per = (ksteps + s - 1) // s
    return (ksteps + per - 1) // per


def igemm_route(N, Hi, Wi, Ci, Ho, Wo, CoStore, Co, k, stride, pad, mode, act, out_f32, splits, tile, w_elems=0,
                has_bias=False, stat_rows_cap=0, stat_group_n=0, want_bn_bwd=False, want_act_y=False,
                want_affine=False) -> str:
    """Name of the kernel instantiation csrc/api.hip::fmri_igemm_ep routes these arguments to -- asked of the library
    itself (fmri_igemm_route: host code only, works without a GPU), the single statement of the routing rules."""
    buf = ctypes.create_string_buffer(160)
    code = lib.load().fmri_igemm_route(N, Hi, Wi, Ci, Ho, Wo, CoStore, Co, k, stride, pad, mode, act, 1 if out_f32 else 0,
                                       splits, tile, int(w_elems), 1 if has_bias else 0, int(stat_rows_cap),
                                       int(stat_group_n), 1 if want_bn_bwd else 0, 1 if want_act_y else 0,
                                       1 if want_affine else 0, buf, len(buf))
    lib.check(code, "fmri_igemm_route")
    return buf.value.decode()


def run_igemm(x, pw: PackedWeight, out, bias, N, Hi, Wi, Ci, Ho, Wo, CoStore, Co, k, stride, pad, mode, act, out_f32,
              splits, slab_stride, tile, flops=0.0, stats=None, bn_bwd=None, act_y=None, affine=None) -> int:
    """``stats`` = (partial-row tensor [groups][rows_cap][2][CoStore] fp32, rows_cap, images per group or 0): ask the
    kernel for the BatchNorm statistics of its output (fmri_igemm_ep).  ``bn_bwd`` = dict(x, gamma, beta, relu, groups =
    [(first image of x, BNSaved), ...]): the BatchNorm-BACKWARD form of that epilogue (masked cotangent + sum g,
    sum g*xhat rows).  Returns the number of rows written per group (0: the kernel behind this geometry has no such
    epilogue -- the output is then the plain contraction)."""
    w = pw.get()
    if lib.PROFILE is not None:
        # algorithmic bytes: the input once, the output once, the weights once
        lib.note(kernel=igemm_route(N, Hi, Wi, Ci, Ho, Wo, CoStore, Co, k, stride, pad, mode, act, out_f32, splits, tile,
                                    w.numel(), bias is not None, stats[1] if stats is not None else 0,
                                    stats[2] if stats is not None else 0, bn_bwd is not None,
                                    act_y is not None and stats is None and affine is None,
                                    affine is not None and stats is None),
                 flops=flops, bytes=2.0 * N * Hi * Wi * Ci + (4.0 if out_f32 else 2.0) * N * Ho * Wo * CoStore * splits
                 + 2.0 * w.numel())
    ep, done = None, ctypes.c_int(0)
    if stats is None and affine is not None:
        # eval-mode BatchNorm of the consumer folded into the epilogue (fmri_epilogue.aff_*); bit EP_AFFINE_APPLIED
        e = lib.Epilogue()
        e.aff_scale, e.aff_shift, e.aff_relu = _P(affine[0]), _P(affine[1]), 1 if affine[2] else 0
        ep = ctypes.byref(e)
    elif stats is None and act_y is not None:
        # ReLU backward of the layer below in the epilogue (fmri_epilogue.act_y); bit EP_ACT_APPLIED of the result
        e = lib.Epilogue()
        e.act_y = _P(act_y)
        ep = ctypes.byref(e)
    if stats is not None:
        e = lib.Epilogue(stats[0].data_ptr(), int(stats[1]), int(stats[2]))
        if bn_bwd is not None:
            e.bn_x, e.bn_gamma, e.bn_beta = _P(bn_bwd["x"]), _P(bn_bwd["gamma"]), _P(bn_bwd["beta"])
            e.bn_relu = 1 if bn_bwd.get("relu", True) else 0
            for i, (img0, sv) in enumerate(bn_bwd["groups"]):
                e.bn_mean[i], e.bn_rstd[i], e.bn_x_img0[i] = _P(sv.mean), _P(sv.rstd), int(img0)
        ep = ctypes.byref(e)
    lib.call("fmri_igemm_ep", _P(x), _P(w), _P(out), _P(bias), _P(zero_page(x.device)), N, Hi, Wi, Ci, Ho, Wo,
             CoStore, Co, k, stride, pad, mode, act, 1 if out_f32 else 0, splits, slab_stride, tile, w.numel(), ep,
             ctypes.byref(done))
    return int(done.value)


# BatchNorm statistics out of the producing contraction's epilogue (fmri_igemm_ep): on/off
_EPI_STATS = os.environ.get("FMRI_EPI_STATS") != "off"
# eval-mode BatchNorm folded into the producing convolution's epilogue (fmri_epilogue.aff_*): on/off
_EPI_AFFINE = os.environ.get("FMRI_EPI_AFFINE") != "off"
# ... and the BatchNorm-BACKWARD form (ReLU mask + sum g, sum g*xhat out of the data gradient's epilogue): off by
# default.  Measured on the B = 256 Stage-I step it removes 0.34 ms of reduction kernels but the epilogues' reads of the
# saved forward tile (8 bytes per lane, latency exposed once per parity class) cost the data-gradient kernels 0.59 ms.
_EPI_BWD = os.environ.get("FMRI_EPI_BWD") == "on"

# ReLU backward of a bias + ReLU layer in the epilogue of the data gradient above it (fmri_epilogue.act_y), bias
# gradient out of the narrow weight-gradient kernel's spare column: on/off
_EPI_ACT = os.environ.get("FMRI_EPI_ACT") != "off"

# window-resident wgrad kernel: on/off, resident-block target (2 per CU), largest split count still written as slabs
_WW_ON = os.environ.get("FMRI_WGRAD_WIN") != "off"
_WN_ON = os.environ.get("FMRI_WGRAD_NARROW") != "off"
# block budget per tile group (FMRI_WW_BLOCKS overrides): one 8-wave block per CU (256) when the kernel has the GPU to
# itself; 160 when it runs on the side stream beside the main stream's kernels -- its blocks hold a CU (144 KB of LDS) for
# their whole K range, and a main-stream kernel that finds no free CU waits for one: leaving ~96 CUs to the main stream
# measured 6.39-6.41 ms per step against 6.48-6.50 (two boxes, two repeats each; 224 / 192: no gain, 128: 6.40-6.42,
# <= 112: 6.8+; tools/probes/ab_ww_blocks.sh).  Deterministic mode keeps 256 in every launch mode, so that its results
# stay bit-identical between one-stream, two-stream and recorded steps.
_WW_BLOCKS = int(os.environ.get("FMRI_WW_BLOCKS", "0"))
_WW_SIDE_BLOCKS = 160
_WW_SLABS = int(os.environ.get("FMRI_WW_SLABS", "24"))


# ------------------------------------------------------------------------------------------------
# deterministic-reduction mode
# ------------------------------------------------------------------------------------------------
# Default: the weight-gradient kernels with many K splits, the narrow 5x5 weight gradient, the latent discriminator's bias
# gradients and the scalar loss sums meet in fp32 atomics, so two runs of the same step differ in the last bits of those
# sums (and RMSprop's normalised first updates can turn a last-bit difference of a near-zero gradient into a different
# step).  ``set_deterministic(True)`` (or FMRI_DETERMINISTIC=1) selects the fixed-order form of each: per-split slabs
# summed in slab order for every fmri_wgrad route, one slab per block for the narrow kernel, the bias gradients of the
# fused MLP backward as column sums of its (bit-reproducible) cotangent tiles, one-block launches of the scalar sums
# (fmri_set_deterministic) -- two runs of a step are then bit-identical, whatever the launch mode (eager, two streams,
# HIP graph).  tests/test_zz_selfcheck_gpu.py compares the launch modes under it.
_DET = {"on": os.environ.get("FMRI_DETERMINISTIC") == "1"}


def set_deterministic(on: bool) -> bool:
    """Switch the deterministic-reduction mode; returns the previous setting."""
    was = _DET["on"]
    _DET["on"] = bool(on)
    lib.load().fmri_set_deterministic(1 if on else 0)
    return was


def deterministic() -> bool:
    return _DET["on"]


# ------------------------------------------------------------------------------------------------
# side stream for weight gradients
# ------------------------------------------------------------------------------------------------
# A layer's weight gradient depends only on (input activation, output cotangent) and nothing downstream of the backward
# pass depends on it before the gradient buffers are all-reduced / consumed by the optimizer.  Issued on a second HIP
# stream it runs beside the HBM-bound BatchNorm-backward kernels and the data-gradient GEMM of the same layer (MFMA-bound
# next to bandwidth-bound work, and the tail of one GEMM grid filled by the next).  Under HIP-graph capture the fork /
# join events become parallel branches of the graph.  FMRI_SIDE_STREAM=off keeps everything on one stream.
_SIDE = {"on": os.environ.get("FMRI_SIDE_STREAM") != "off", "streams": {}, "pending": []}


def _side_stream(device):
    if not _SIDE["on"]:
        return None
    key = (device.type, device.index)
    st = _SIDE["streams"].get(key)
    if st is None:
        st = _SIDE["streams"][key] = torch.cuda.Stream(device)
        # everything the fork / join below needs, looked up once: the switch to the side stream and back is two
        # _cuda_setStream calls (the ``torch.cuda.stream`` context manager costs ~25 us of Python per use, 55 times per
        # step), the fork and join events are two reused Event objects (a wait captures the event's state when it is
        # issued, so re-recording the same event later is well defined)
        st._fmri_ids = (st.stream_id, st.device_index, st.device_type)
        st._fmri_fork = torch.cuda.Event()
        st._fmri_join = torch.cuda.Event()
    return st


_set_stream = torch._C._cuda_setStream


def side_run(device, fn, *keep):
    """Run ``fn`` (kernel launches only) on the side stream, ordered after everything issued so far on the current
    stream.  ``keep``: tensors ``fn`` reads -- referenced until ``join_side`` so that the caching allocator cannot hand
    their memory to later current-stream work while the side stream still reads it."""
    st = _side_stream(device)
    if st is None:
        fn()
        return
    cur = torch.cuda.current_stream(device)
    st._fmri_fork.record(cur)
    st.wait_event(st._fmri_fork)
    sid, didx, dtype = st._fmri_ids
    _set_stream(stream_id=sid, device_index=didx, device_type=dtype)
    try:
        fn()
    finally:
        _set_stream(stream_id=cur.stream_id, device_index=cur.device_index, device_type=cur.device_type)
    _SIDE["pending"].append((st, keep))


def join_side(device=None):
    """Make the current stream wait for all side-stream work issued since the last join."""
    if not _SIDE["pending"]:
        return
    seen = []
    for st, _ in _SIDE["pending"]:
        if st not in seen:
            seen.append(st)
    for st in seen:
        st._fmri_join.record(st)
        torch.cuda.current_stream(st.device).wait_event(st._fmri_join)
    _SIDE["pending"].clear()


def run_wgrad(P, Q, N, Yc, Xc, A, Hq, Wq, Bc, k, stride, pad, flip=0, flops=0.0, hold=None, gate=None):
    """Returns the packed fp32 gradient [apad][ldo].  ``hold``: see ``_grad_buffer``.  ``flops``: algorithmic FLOPs of the layer's weight gradient (for
    the profiling hook of lib.call; 0 = 2 * rows * A * Bc * k^2 of the padded operands)."""
    ba = tile_for(A)
    apad = ceil_to(A, ba)
    ldo = ceil_to(k * k * Bc, 128)
    if lib.PROFILE is not None:
        fl = flops or 2.0 * N * Yc * Xc * A * Bc * k * k
        nb = 2.0 * N * Yc * Xc * A + 2.0 * N * Hq * Wq * Bc + 4.0 * apad * ldo
        note = lambda kern: lib.note(kernel=kern, flops=fl, bytes=nb)
    else:
        note = lambda kern: None
    if (stride == 2 and not flip and ba == 128 and Bc % 32 == 0 and Yc * Xc > 1 and k == 5 and pad == 2 and _WW_ON
            and 2 * N * Yc * Xc * A < 2 ** 31 and 2 * N * Hq * Wq * Bc < 2 ** 31):       # 32-bit buffer offsets
        # window-resident kernel (csrc/wgrad_win.hip): blocks = 32-channel column blocks x 128-row blocks x 4 parity
        # planes x splits over the 8x8 pixel tiles; every split stores its own slab, unpack_grad sums them
        groups = (Bc // 32) * (apad // 128)
        budget = _WW_BLOCKS or (_WW_SIDE_BLOCKS if (_SIDE["on"] and not _DET["on"]) else 256)
        splits = max(4, budget // groups)                               # block budget per group over the 4 planes
        nslabs = lib.load().fmri_wgrad_slabs(N, Yc, Xc, k, pad, splits)
        slabs = nslabs <= _WW_SLABS or _DET["on"]                       # few splits: per-split slabs, else atomics
        out = _grad_buffer(hold, (nslabs, apad, ldo) if slabs else (apad, ldo), not slabs, P.device)
        note("fmri::wgrad_win_kernel")
        lib.call("fmri_wgrad_if", _P(gate), _P(P), _P(Q), _P(out), _P(zero_page(P.device)), N, Yc, Xc, A, Hq, Wq, Bc, k, stride,
                 pad, flip, apad, ba, ldo, splits, 2 if slabs else 1)
        return out, ldo
    if (stride == 1 and k == 5 and pad == 2 and A == 32 and Bc == 8 and Yc == Hq and Xc == Wq and _WN_ON
            and N * ((Yc + 7) // 8) * ((Xc + 7) // 8) >= 32768):
        # wave-private window kernel (csrc/wgrad_narrow.hip): atomic accumulation into zeroed 32 x ldo slab(s).  It
        # needs >= ~16 tiles per wave to amortise its block reduction: measured 1.9x on the 3B-image discriminator
        # layer, a loss on the B-image decoder layer
        nslabs = 4                   # blocks add into slab (block index % 4): a quarter of the same-address atomics
        if _DET["on"]:               # one slab per block: every element is added once, onto zero
            nslabs = lib.load().fmri_wgrad_narrow_blocks(N, Yc, Xc)
        out = _grad_buffer(hold, (nslabs, apad, ldo), True, P.device)
        note("fmri::wgrad_narrow_kernel")
        lib.call("fmri_wgrad_if", _P(gate), _P(P), _P(Q), _P(out), _P(zero_page(P.device)), N, Yc, Xc, A, Hq, Wq, Bc, k, stride, pad,
                 flip, apad, ba, ldo, nslabs, 3)
        out._fmri_colsum = True            # column 200 of every slab row a holds sum_m P[m][a]
        return out, ldo
    tiles = (ldo // 128) * (apad // ba)
    steps = (N * Yc * Xc + 63) // 64
    splits = 1
    if tiles < 512 and steps >= 16:
        splits = min(steps // 8, (1024 + tiles - 1) // tiles)
        splits = max(splits, 1)
    mode = 1 if splits > 1 else 0
    if splits > 1 and _DET["on"]:
        mode = 4                     # per-split slabs (plain stores), summed in slab order by unpack_grad
        # zero-filled: the library re-derives the split count from the K steps (ceil(steps / ceil(steps / splits)), which
        # can be smaller than ``splits`` -- e.g. 18 of 19 at the 100-px encoder conv.0 with 4 images) and stores only that
        # many slabs, while unpack_grad / fmri_apply_batch sum all of them
        out = _grad_buffer(hold, (splits, apad, ldo), True, P.device)
    elif splits > 1:
        out = _grad_buffer(hold, (apad, ldo), True, P.device)
    else:
        out = _grad_buffer(hold, (apad, ldo), False, P.device)
    note("fmri::wgrad_kernel")
    lib.call("fmri_wgrad_if", _P(gate), _P(P), _P(Q), _P(out), _P(zero_page(P.device)), N, Yc, Xc, A, Hq, Wq, Bc, k, stride, pad,
             flip, apad, ba, ldo, splits, mode)
    return out, ldo


def unpack_grad(packed, grad_view, sp: PackSpec, ld: int, scale: float):
    """packed: [rows][ld] or [nslabs][rows][ld] (per-split partial results, summed here)."""
    nslabs = packed.shape[0] if packed.dim() == 3 else 1
    lib.call("fmri_unpack_grad", _P(packed), _P(grad_view), sp.sa, sp.sta, sp.sb, sp.stb, sp.A, sp.TA, sp.B, sp.KW,
             sp.py, sp.px, sp.step, sp.TH, sp.TW, ld, float(scale), 1, nslabs, packed.shape[-2] * packed.shape[-1])


def _grad_buffer(hold, shape, zeroed: bool, device):
    """Output buffer of a weight-gradient launch.  ``hold`` is None: a fresh tensor per call (zeroed when the kernel
    adds into it).  Otherwise (deferred gradients, ``begin_grads``): the layer's persistent buffer -- its address sits
    in the device table of fmri_apply_batch, which also writes the zeros back after it has consumed the sums."""
    if hold is None or hold.get("busy"):
        out = (torch.zeros if zeroed else torch.empty)(shape, dtype=torch.float32, device=device)
        out._fmri_clear = False
        return out
    key = (tuple(shape), zeroed)
    out = hold.get(key)
    if out is None:
        if len(hold) > 6:                # (a batch size that keeps changing: do not collect a buffer per shape)
            for k in [k for k in hold if k != "busy"]:
                del hold[k]
        out = hold[key] = torch.zeros(shape, dtype=torch.float32, device=device)
        out._fmri_clear = zeroed
        out._fmri_hold = hold
    hold["busy"] = True
    return out


def emit_grad(group, packed, grad_view, sp: PackSpec, ld: int, scale: float):
    """What a weight-gradient launch does with its packed result: added to the reference-layout gradient now
    (fmri_unpack_grad), or -- between ``begin_grads(group, defer=True)`` and ``apply_group`` -- queued for the
    sub-network's one fmri_apply_batch launch."""
    if getattr(group, "defer_grads", False):
        group.pending.append((packed, grad_view, sp, ld, scale))
    else:
        unpack_grad(packed, grad_view, sp, ld, scale)


def _hold_of(layer):
    """The layer's persistent gradient buffers while its group defers gradients, else None."""
    if not getattr(layer.group, "defer_grads", False):
        return None
    h = layer.__dict__.get("_ghold")
    if h is None:
        h = layer._ghold = {}
    return h


def _gate_of(layer):
    """The device flag the layer's weight-gradient launch is conditioned on (``begin_grads(..., gate=)``), or None."""
    g = layer.group
    return getattr(g, "grad_gate", None) if getattr(g, "defer_grads", False) else None


_FUSED_APPLY = os.environ.get("FMRI_FUSED_APPLY") != "off"
_GATE_SKIP = os.environ.get("FMRI_GATE_SKIP") != "off"


def begin_grads(group, defer: bool, gate: Optional[torch.Tensor] = None):
    """Start of a backward pass over ``group`` (replaces ``zero_grad``).  ``defer``: the caller will hand the whole
    group to ``apply_group`` right after the pass and nobody reads reference-layout gradients in between -- weight
    gradients then stay in their GEMM layout until that one launch, and only the 1-D parameters' gradient segments
    (which the backward pass accumulates in place) are cleared here instead of the whole buffer."""
    defer = bool(defer and _FUSED_APPLY)
    group.drop_pending()
    group.defer_grads = defer
    group.grads_consumed = False
    # ``gate``: device int the group's update is conditioned on (the equilibrium gate's train_dis / train_dec): its
    # weight-gradient GEMMs are launched with fmri_wgrad_if and do nothing in a step that does not train the group --
    # the reference does not run that loss.backward() at all (train_vgan_stage1.py:420-431)
    group.grad_gate = gate if (defer and _GATE_SKIP) else None
    plan = getattr(group, "_flat_clear", None) if defer else None
    if plan is not None and plan["flat"] is not None:
        lib.call("fmri_apply_batch", _P(plan["flat"]), plan["flat_n"], plan["flat_tiles"], 2, None, 0.0, 0.0, 1.0,
                 None, 0.0, None, 0)
        group._cleared = plan["sig"]
    else:
        group.grad.zero_()
        group._cleared = "all"


def _plan_apply(group, state, update=True, pend=None):
    """Device table of fmri_apply_batch for the pending gradients of ``group`` (cached while the same buffers come
    back), or None when some tensor cannot go through it."""
    L = lib.load()
    pend = group.pending if pend is None else pend
    key = tuple((p[0].data_ptr(), tuple(p[0].shape), p[1].data_ptr(), p[3], p[4]) for p in pend) + \
        (state.data_ptr(), update)
    plans = group.__dict__.setdefault("_apply_plans", {})
    plan = plans.get(key)
    if plan is not None:
        return plan
    nbytes = L.fmri_apply_entry_bytes()
    g0, n_all = group.grad.data_ptr(), group.grad.numel()
    rows, flat_rows, covered, fused, tiles = [], [], [], [], 0
    by_master = {}
    for pw in getattr(group, "packed", []):
        if len(pw.specs) == 1:
            by_master.setdefault(pw.master.data_ptr(), []).append(pw)
    for packed, gv, sp, ld, scale in pend:
        off = (gv.data_ptr() - g0) // 4
        if not (0 <= off and off + gv.numel() <= n_all) or not gv.is_contiguous():
            return None
        if any(off < o + n and o < off + gv.numel() for o, n in covered):
            return None                                   # two gradients of one tensor in one pass: separate launches
        nsl = packed.shape[0] if packed.dim() == 3 else 1
        pk, kpad = None, 0
        for pw in by_master.get(group.data.data_ptr() + 4 * off, []) if update else ():
            if pw.specs[0] == sp and pw.rows_pad >= sp.rows:
                pk, kpad = pw.buf.data_ptr() + 2 * pw.offsets[0], pw.kpads[0]
                fused.append(pw)
                break
        host = ctypes.create_string_buffer(nbytes)
        n = L.fmri_apply_entry_fill(host, packed.data_ptr(), group.data.data_ptr() + 4 * off,
                                    state.data_ptr() + (4 * off if update else 0), gv.data_ptr(), pk, sp.sa, sp.sta, sp.sb, sp.stb, sp.A, sp.TA, sp.B, sp.KW, sp.py,
                                    sp.px, sp.step, sp.TH, sp.TW, ld, kpad, nsl, packed.shape[-2] * packed.shape[-1],
                                    1 if getattr(packed, "_fmri_clear", False) else 0, float(scale), 0, tiles)
        if n < 0:
            lib.check(n, "fmri_apply_entry_fill")
        if n == 0:
            return None
        rows.append(host.raw)
        tiles += n
        covered.append((off, gv.numel()))
    # everything else of the buffer: 1-D parameters (and tensors without a weight-gradient GEMM), updated from the
    # reference-layout gradient the backward pass accumulated in place
    covered.sort()
    segs, at = [], 0
    for o, n in covered + [(n_all, 0)]:
        if o > at:
            segs.append((at, o - at))
        at = max(at, o + n)
    ftiles = 0
    for o, n in segs:
        for tile0, dst in ((tiles, rows), (ftiles, flat_rows)):
            host = ctypes.create_string_buffer(nbytes)
            k = L.fmri_apply_entry_fill(host, None, group.data.data_ptr() + 4 * o,
                                        state.data_ptr() + (4 * o if update else 0), g0 + 4 * o,
                                        None, 0, 0, 0, 0, 1, 1, 1, 1, 0, 0, 1, 1, 1, 0, 0, 1, 0, 0, 1.0, n, tile0)
            if k <= 0:
                lib.check(k if k < 0 else -1, "fmri_apply_entry_fill")
            dst.append(host.raw)
        tiles += k
        ftiles += k
    up = lambda rr: torch.frombuffer(bytearray(b"".join(rr)), dtype=torch.uint8).to(group.device) if rr else None
    plan = dict(key=key, table=up(rows), n=len(rows), tiles=tiles, flat=up(flat_rows), flat_n=len(flat_rows),
                flat_tiles=ftiles, sig=tuple(segs), covered=covered, fused=fused)
    if len(plans) >= 8:              # (buffers that keep changing, e.g. a varying batch size: do not collect tables)
        keep = getattr(group, "_flat_clear", None)
        plans.clear()
        if keep is not None:
            plans[keep["key"]] = keep
    plans[key] = plan
    if getattr(group, "_flat_clear", None) is None:
        # what begin_grads clears from now on: the first table's flat segments -- every 1-D parameter is in them (and,
        # when the table was built for a part of the group, tensors of the rest: cleared for nothing, a few MB)
        group._flat_clear = plan
    return plan


def _segments_within(segs, cleared) -> bool:
    """Every flat segment the table updates from the in-place gradient buffer was cleared at the start of the pass
    (``cleared``: "all", or the segments begin_grads cleared -- a superset when its table was built for a part of the
    group, as in the data-parallel encoder)."""
    if cleared == "all":
        return True
    return all(any(co <= o and o + n <= co + cn for co, cn in cleared) for o, n in segs)


def flush_pending(group, keep=()):
    """Deferred gradients -> the reference-layout gradient buffer (what the separate launches would have left there).
    ``keep``: queue entries whose gradients materialize_grads has already put there."""
    pend, group.pending = getattr(group, "pending", []), []
    group.defer_grads = False
    if getattr(group, "_cleared", "all") != "all":
        # only the 1-D segments were cleared at the start of this pass: clear what the tensors' ranges still hold
        g0 = group.grad.data_ptr()
        holes = sorted(list(group._cleared) + [((p[1].data_ptr() - g0) // 4, p[1].numel()) for p in keep])
        at = 0
        for o, n in holes + [(group.grad.numel(), 0)]:
            if o > at:
                group.grad[at:o].zero_()
            at = max(at, o + n)
        group._cleared = "all"
    for packed, gv, sp, ld, scale in pend:
        unpack_grad(packed, gv, sp, ld, scale)
        if getattr(packed, "_fmri_clear", False):
            packed.zero_()
        if getattr(packed, "_fmri_hold", None) is not None:
            packed._fmri_hold["busy"] = False


def materialize_grads(group):
    """Deferred gradients -> reference-layout gradient buffer in ONE launch (fmri_apply_batch mode 0: slab sums mapped
    and stored, no read-modify-write, no memset in front) -- what a data-parallel step does before the gradient
    all-reduce.  May be called for a part of the group's tensors (the ones queued so far)."""
    if not getattr(group, "defer_grads", False) or not group.pending:
        return
    st = group.__dict__.get("_no_state")
    if st is None:
        st = group._no_state = torch.zeros(1, dtype=torch.float32, device=group.device)
    plan = _plan_apply(group, st, update=False)
    if plan is None:
        # (keep what an earlier partial call has already put into the reference layout -- it may be mid-reduction)
        flush_pending(group, keep=getattr(group, "materialized", []))
        return
    lib.note(bytes=8.0 * group.numel)
    lib.call("fmri_apply_batch", _P(plan["table"]), plan["n"], plan["tiles"], 0, None, 0.0, 0.0, 1.0, None, 0.0, None, 0)
    for p in group.pending:
        h = getattr(p[0], "_fmri_hold", None)
        if h is not None:
            h["busy"] = False
    group.materialized = getattr(group, "materialized", []) + group.pending
    group.pending = []            # (defer_grads stays on: the rest of the pass may queue more)


def apply_group(group, state, lr_dev, alpha, eps, flag, gdev, clamp=0.0) -> bool:
    """RMSprop update of a whole sub-network from its deferred gradients + the fp16 GEMM copies of the new weights:
    fmri_apply_batch (one launch) and the pack launch(es) of the orientations it does not write.  Returns False -- after
    bringing the reference-layout gradient buffer up to date -- when the group has to take the separate launches
    (nothing deferred, or a tensor the table cannot describe); the caller then runs the optimizer and the re-pack."""
    if not getattr(group, "defer_grads", False):
        return False
    done = getattr(group, "materialized", [])
    group.materialized = []
    plan, mode = None, 1
    if group.pending and not done:
        plan = _plan_apply(group, state)
    elif done and not group.pending:
        # data parallel: every gradient already sits in the reference layout (materialize_grads, all-reduced since); the
        # same table drives the update from there (mode 3)
        mode = 3
        plan = _plan_apply(group, state, pend=done)
    if plan is None or not _segments_within(plan["sig"], group._cleared):
        flush_pending(group, keep=done)
        return False
    lib.note(bytes=22.0 * group.numel)
    gate = getattr(group, "grad_gate", None)
    gated = 1 if (gate is not None and flag is not None and gate.data_ptr() == flag.data_ptr()) else 0
    lib.call("fmri_apply_batch", _P(plan["table"]), plan["n"], plan["tiles"], mode, _P(lr_dev), alpha, eps, 1.0,
             _P(gdev), clamp, _P(flag), gated)
    for p in group.pending:
        h = getattr(p[0], "_fmri_hold", None)
        if h is not None:
            h["busy"] = False
    group.pending = []
    group.defer_grads = False
    group.grads_consumed = mode == 1      # the reference-layout gradient buffer was never written in this pass
    group.version += 1
    for pw in plan["fused"]:
        pw.version = group.version
    _repack_group(group, skip=plan["fused"])
    return True


# ------------------------------------------------------------------------------------------------
# convolution layers
# ------------------------------------------------------------------------------------------------
class ConvLayer:
    """nn.Conv2d(k, stride, pad) (kind='conv', weight [Cout][Cin][k][k]) or
    nn.ConvTranspose2d(k, 2, pad, output_padding) (kind='deconv', weight [Cin][Cout][k][k])."""

    def __init__(self, group, wkey: str, bkey: Optional[str], kind: str, cin: int, cout: int, k: int, stride: int,
                 pad: int, out_pad: int = 0):
        self.group, self.kind = group, kind
        self.w = group.views[wkey]
        self.wg = group.grads[wkey]
        self.b = group.views[bkey] if bkey else None
        self.bg = group.grads[bkey] if bkey else None
        self.cin, self.cout, self.k, self.stride, self.pad, self.out_pad = cin, cout, k, stride, pad, out_pad
        self.cinp, self.coutp = pad8(cin), pad8(cout)
        kk = k * k
        self.t_out, self.t_in = tile_for(cout), tile_for(cin)
        if kind == "conv":
            # forward: rows co, reduce (tap, ci)
            self.pw_f = _single(self.w, group, PackSpec(sa=cin * kk, sta=0, A=cout, TA=1, sb=kk, stb=1, B=cin, KW=k,
                                                        TH=k, TW=k), self.t_out)
            if stride == 2:
                self.pw_d = _tconv(self.w, group, k, pad, sa=kk, A=cin, sb=cin * kk, B=cout, tile=self.t_in)
            else:
                self.pw_d = _single(self.w, group, PackSpec(sa=kk, sta=0, A=cin, TA=1, sb=cin * kk, stb=1, B=cout,
                                                            KW=k, TH=k, TW=k), self.t_in)
            self.gspec = PackSpec(sa=cin * kk, sta=0, A=cout, TA=1, sb=kk, stb=1, B=cin, KW=k, TH=k, TW=k)
            if stride == 2:
                self.pw_d.taps_of = self.pw_f       # class blocks = per-tap transposes of the forward copy (_repack_group)
        else:
            if stride != 2:
                raise ValueError("deconv layers are stride 2")
            self.pw_f = _tconv(self.w, group, k, pad, sa=kk, A=cout, sb=cout * kk, B=cin, tile=self.t_out)
            self.pw_d = _single(self.w, group, PackSpec(sa=cout * kk, sta=0, A=cin, TA=1, sb=kk, stb=1, B=cout, KW=k,
                                                        TH=k, TW=k), self.t_in)
            self.gspec = PackSpec(sa=cout * kk, sta=0, A=cin, TA=1, sb=kk, stb=1, B=cout, KW=k, TH=k, TW=k)
            self.pw_f.taps_of = self.pw_d

    def out_hw(self, hi: int, wi: int) -> Tuple[int, int]:
        if self.kind == "conv":
            f = lambda v: (v + 2 * self.pad - self.k) // self.stride + 1
        else:
            f = lambda v: (v - 1) * self.stride - 2 * self.pad + self.k + self.out_pad
        return f(hi), f(wi)

    def forward(self, x: torch.Tensor, act: int = ACT_NONE, out: Optional[torch.Tensor] = None,
                bn_groups: int = 0, affine=None) -> torch.Tensor:
        """``bn_groups`` > 0: the output goes into a train-mode BatchNorm whose batch is each of ``bn_groups`` equal
        image ranges; the kernel's epilogue is asked for the batch statistics (fmri_igemm_ep) and ``take_stats(g)``
        hands group g's accumulator to ``BatchNorm.forward`` (None if this geometry's kernel has no such epilogue).
        ``affine`` = (scale, shift, relu) of an EVAL-mode BatchNorm behind this layer (``BatchNorm.eval_affine``): asked
        of the kernel's epilogue; ``self.aff_applied`` tells whether the output already is the BatchNorm's."""
        N, Hi, Wi, C = x.shape
        assert C == self.cinp and x.dtype == torch.float16 and x.is_contiguous()
        Ho, Wo = self.out_hw(Hi, Wi)
        if out is None:
            out = torch.empty(N, Ho, Wo, self.coutp, dtype=torch.float16, device=x.device)
        mode = MODE_CONV if self.kind == "conv" else MODE_TCONV2
        stats = None
        self._stat_rows = 0
        if bn_groups > 0 and _EPI_STATS and N % bn_groups == 0:
            cap = (N // bn_groups) * Ho * Wo // 128 + 8
            part = getattr(self, "_stat_part", None)
            if part is None or part.shape[0] < bn_groups or part.shape[1] < cap:
                part = self._stat_part = torch.empty(bn_groups, cap, 2, self.coutp, dtype=torch.float32,
                                                     device=x.device)
            stats = (part, part.shape[1], N // bn_groups if bn_groups > 1 else 0)
        if affine is not None and (stats is not None or self.b is not None or act != ACT_NONE or not _EPI_AFFINE):
            affine = None
        r = run_igemm(x, self.pw_f, out, self.b, N, Hi, Wi, self.cinp, Ho, Wo, self.coutp, self.cout,
                      self.k, self.stride, self.pad, mode, act, False, 1, 0, self.t_out,
                      self._flops(N, Hi, Wi, Ho, Wo), stats=stats, affine=affine)
        self._stat_rows = (r & ~(lib.EP_ACT_APPLIED | lib.EP_AFFINE_APPLIED)) if stats is not None else 0
        self.aff_applied = bool(r & lib.EP_AFFINE_APPLIED)
        return out

    def take_stats(self, g: int = 0) -> Optional[torch.Tensor]:
        """Rows [P][2][coutp] with the per-block batch statistics of group ``g`` of the last forward, or None."""
        return self._stat_part[g, :self._stat_rows] if getattr(self, "_stat_rows", 0) > 0 else None

    def _flops(self, N, Hi, Wi, Ho, Wo):
        """Algorithmic FLOPs of one pass (fwd == dgrad == wgrad): 2 * pixels * cin * cout * k^2, pixels = conv
        output pixels (conv) or deconv input pixels (deconv)."""
        pix = Ho * Wo if self.kind == "conv" else Hi * Wi
        return 2.0 * N * pix * self.cin * self.cout * self.k * self.k

    def dgrad(self, dy: torch.Tensor, hi: int, wi: int, out: Optional[torch.Tensor] = None,
              bn_bwd: Optional[dict] = None, relu_y: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Cotangent w.r.t. the layer input (same scale as dy).

        ``relu_y``: the saved ReLU output this layer consumed (geometry of the result): ask the kernel's epilogue for the
        ReLU backward as well (fmri_epilogue.act_y).  ``self.act_applied`` tells whether it did -- if not, the result is
        the plain data gradient and ``act_backward`` has to follow.

        ``bn_bwd`` = dict(bn=<BatchNorm whose (ReLU'd) output is this layer's input>, x=<its saved forward input>,
        groups=[(first image of x, BNSaved), ...]): the rows of ``dy`` are len(groups) equal blocks (cotangent streams /
        decoder calls); the kernel's epilogue applies the ReLU mask and emits that BatchNorm's backward statistics
        (fmri_epilogue.bn_x).  ``take_bwd_stats()`` then returns (rows [G][cap][2][C], valid rows) for
        ``BatchNorm.backward(..., stat=...)``, or None when this geometry's kernel has no such epilogue (the output is
        then the plain data gradient)."""
        N, Ho, Wo, C = dy.shape
        assert C == self.coutp and dy.is_contiguous()
        if out is None:
            out = torch.empty(N, hi, wi, self.cinp, dtype=torch.float16, device=dy.device)
        fl = self._flops(N, hi, wi, Ho, Wo)
        stats, bb = None, None
        self._bwd_rows = 0
        if bn_bwd is not None and _EPI_STATS and self.cinp == self.cin:
            bn, G = bn_bwd["bn"], len(bn_bwd["groups"])
            if bn.C == self.cinp and not bn.perm and N % G == 0 and G <= 4:
                cap = (N // G) * hi * wi // 128 + 8
                part = getattr(self, "_bwd_part", None)
                if part is None or part.shape[0] < G or part.shape[1] < cap:
                    part = self._bwd_part = torch.empty(max(G, 2), cap, 2, self.cinp, dtype=torch.float32,
                                                        device=dy.device)
                gamma, beta, _, _ = bn._params()
                stats = (part, part.shape[1], N // G if G > 1 else 0)
                bb = dict(x=bn_bwd["x"], gamma=gamma, beta=beta, relu=bn_bwd.get("relu", True), groups=bn_bwd["groups"])
        if self.kind == "conv":
            mode = MODE_TCONV2 if self.stride == 2 else MODE_CONV_FLIP
            r = run_igemm(dy, self.pw_d, out, None, N, Ho, Wo, self.coutp, hi, wi, self.cinp, self.cin, self.k,
                          self.stride, self.pad, mode, ACT_NONE, False, 1, 0, self.t_in, fl, stats=stats, bn_bwd=bb,
                          act_y=relu_y if _EPI_ACT else None)
        else:
            r = run_igemm(dy, self.pw_d, out, None, N, Ho, Wo, self.coutp, hi, wi, self.cinp, self.cin, self.k, 2,
                          self.pad, MODE_CONV, ACT_NONE, False, 1, 0, self.t_in, fl, stats=stats, bn_bwd=bb)
        self._bwd_rows = (r & ~(lib.EP_ACT_APPLIED | lib.EP_AFFINE_APPLIED)) if stats is not None else 0
        self.act_applied = bool(r & lib.EP_ACT_APPLIED)
        return out

    def take_bwd_stats(self):
        """(rows tensor [G][cap][2][cinp], valid rows per group) of the last ``dgrad(..., bn_bwd=...)``, or None."""
        return (self._bwd_part, self._bwd_rows) if getattr(self, "_bwd_rows", 0) > 0 else None

    def wgrad(self, x: torch.Tensor, dy: torch.Tensor, scale: float, bias_too: bool = False):
        """weight.grad += (1/scale) * dW(x, dy)  (on the side stream: ops.join_side() before the gradient is read).
        ``bias_too`` (kind='conv', stride 1, cin < cout): bias.grad += (1/scale) * sum_pixels(dy) as well."""
        side_run(x.device, lambda: self._wgrad(x, dy, scale, bias_too), x, dy)

    def _wgrad(self, x: torch.Tensor, dy: torch.Tensor, scale: float, bias_too: bool = False):
        N, Hi, Wi, _ = x.shape
        _, Ho, Wo, _ = dy.shape
        if self.kind == "conv" and self.stride == 1 and self.cinp > self.coutp:
            if bias_too:
                raise NotImplementedError("ConvLayer.wgrad(bias_too=True) on the role-exchanged branch (cin > cout): "
                                          "reduce the bias gradient with act_backward(colsum=...) as DecoderNet does")
            # exchange the roles (dW[co][ci][k] = sum_m' X[m'][ci] * dY[m' + pad - k][co]) so that the gathered
            # operand is the narrow one: rows ci, columns (tap, co)
            packed, ldo = run_wgrad(x, dy, N, Hi, Wi, self.cinp, Ho, Wo, self.coutp, self.k, 1, self.pad, flip=1,
                                    flops=self._flops(N, Hi, Wi, Ho, Wo), hold=_hold_of(self), gate=_gate_of(self))
            kk = self.k * self.k
            spec = PackSpec(sa=kk, sta=0, A=self.cin, TA=1, sb=self.cin * kk, stb=1, B=self.cout, KW=self.k,
                            TH=self.k, TW=self.k)
            emit_grad(self.group, packed, self.wg, spec, ldo, 1.0 / scale)
            return
        if self.kind == "conv":
            packed, ldo = run_wgrad(dy, x, N, Ho, Wo, self.coutp, Hi, Wi, self.cinp, self.k, self.stride, self.pad,
                                    flops=self._flops(N, Hi, Wi, Ho, Wo), hold=_hold_of(self), gate=_gate_of(self))
        else:
            packed, ldo = run_wgrad(x, dy, N, Hi, Wi, self.cinp, Ho, Wo, self.coutp, self.k, 2, self.pad,
                                    flops=self._flops(N, Hi, Wi, Ho, Wo), hold=_hold_of(self), gate=_gate_of(self))
        emit_grad(self.group, packed, self.wg, self.gspec, ldo, 1.0 / scale)
        if bias_too:
            # bias.grad += (1/scale) * sum over pixels of dy: the narrow kernel's spare column, else a reduction
            if getattr(packed, "_fmri_colsum", False):
                # slabs [nslabs][apad][ldo]: element (a, column k*k*cinp) of every slab
                colsum_acc(packed[0, 0, self.k * self.k * self.cinp:], packed.shape[0], self.cout,
                           packed.shape[1] * packed.shape[2], packed.shape[2], 1.0 / scale, self.bg)
            else:
                d2 = dy.reshape(-1, self.coutp)
                colsum_acc(d2, d2.shape[0], self.cout, self.coutp, 1, 1.0 / scale, self.bg)


# ------------------------------------------------------------------------------------------------
# dense layers
# ------------------------------------------------------------------------------------------------
class DenseLayer:
    """nn.Linear(K -> Nout), weight [Nout][K].

    in_perm=(C, HW):  the K inputs are a flattened conv map stored (H,W,C) in the engine but (C,H,W) in the
                      reference weight (models/vae_gan.py:89,181).
    out_perm=(C, HW): the Nout outputs are reshaped to a conv map (models/vae_gan.py:127); the engine emits
                      them in (H,W,C) order.
    """

    def __init__(self, group, wkey, bkey, k_in: int, n_out: int, in_perm=None, out_perm=None):
        self.group = group
        if isinstance(wkey, str):
            self.w, self.wg = group.views[wkey], group.grads[wkey]
            self.b = group.views[bkey] if bkey else None
            self.bg = group.grads[bkey] if bkey else None
        else:                       # explicit (weight, grad) / (bias, grad) tensor pairs (fused heads)
            self.w, self.wg = wkey
            self.b, self.bg = bkey if bkey else (None, None)
        self.k_in, self.n_out = k_in, n_out
        self.kp, self.np_ = pad8(k_in), pad8(n_out)
        # 64-wide output tiles for the dense layers: they run on a few hundred rows, so the launch is filled by split-K
        # slabs -- twice the tiles halve the slabs (and their fp32 write + re-read): -15...-20 % on the two
        # 16384-wide layers, nothing lost on the others (tools/probes/dense_tiles.py, profiles/r04_dense_tiles.log)
        self.t_out, self.t_in = min(64, tile_for(n_out)), min(64, tile_for(k_in))
        K, Nn = k_in, n_out
        if in_perm:
            C, HW = in_perm
            assert C * HW == K and C % 8 == 0
            f = PackSpec(sa=K, sta=0, A=Nn, TA=1, sb=HW, stb=1, B=C, KW=HW, TH=1, TW=HW)
            d = PackSpec(sa=HW, sta=1, A=C, TA=HW, sb=K, stb=0, B=Nn)
            g = f
        elif out_perm:
            C, HW = out_perm
            assert C * HW == Nn and C % 8 == 0
            f = PackSpec(sa=HW * K, sta=K, A=C, TA=HW, sb=1, stb=0, B=K)
            d = PackSpec(sa=1, sta=0, A=K, TA=1, sb=HW * K, stb=K, B=C, KW=HW, TH=1, TW=HW)
            g = f
        else:
            f = PackSpec(sa=K, sta=0, A=Nn, TA=1, sb=1, stb=0, B=K)
            d = PackSpec(sa=1, sta=0, A=K, TA=1, sb=K, stb=0, B=Nn)
            g = f
        self.pw_f = _single(self.w, group, f, self.t_out)
        self.pw_d = _single(self.w, group, d, self.t_in)
        # the data-gradient orientation [k][n] is the transpose of the forward one [n][k] (the flatten permutation is the
        # same column / row order in both): _repack_group makes it from the fp16 copy
        if f.kcols == d.rows and d.kcols >= f.rows:
            self.pw_d.transpose_of = self.pw_f
        self.gspec = g

    def _gemm(self, x, pw, M, Ci, Co, CoStore, tile, bias, act, want16, want32):
        ksteps = ceil_to(Ci, 64) // 64
        blocks = ((M + 127) // 128) * (ceil_to(Co, tile) // tile)
        splits = _choose_splits(blocks, ksteps)
        dev = x.device
        if splits == 1 and not want32:
            out = torch.empty(M, CoStore, dtype=torch.float16, device=dev)
            run_igemm(x, pw, out, bias, M, 1, 1, Ci, 1, 1, CoStore, Co, 1, 1, 0, MODE_CONV, act, False, 1, 0, tile,
                      2.0 * M * Ci * Co)
            return out, None
        slabs = torch.empty(splits, M, CoStore, dtype=torch.float32, device=dev)
        run_igemm(x, pw, slabs, None, M, 1, 1, Ci, 1, 1, CoStore, Co, 1, 1, 0, MODE_CONV, ACT_NONE, True, splits,
                  M * CoStore, tile)
        o16 = torch.empty(M, CoStore, dtype=torch.float16, device=dev) if want16 else None
        o32 = torch.empty(M, Co, dtype=torch.float32, device=dev) if want32 else None
        lib.call("fmri_reduce_slabs", _P(slabs), splits, M * CoStore, M, Co, CoStore, _P(bias), act, _P(o32), Co,
                 _P(o16), CoStore)
        return o16, o32

    def forward(self, x: torch.Tensor, act: int = ACT_NONE, want16: bool = True, want32: bool = False):
        """x [M, Kp] fp16 -> (out16 [M, Np] or None, out32 [M, Nout] or None); bias (if any) and act fused."""
        M, Kp = x.shape
        assert Kp == self.kp and x.is_contiguous()
        return self._gemm(x, self.pw_f, M, self.kp, self.n_out, self.np_, self.t_out, self.b, act, want16, want32)

    def dgrad(self, dy: torch.Tensor, want32: bool = False):
        """dy [M, Np] -> dx [M, Kp] (fp16, same scale) and/or fp32 [M, K]."""
        M, Np = dy.shape
        assert Np == self.np_ and dy.is_contiguous()
        return self._gemm(dy, self.pw_d, M, self.np_, self.k_in, self.kp, self.t_in, None, ACT_NONE, not want32,
                          want32)

    def wgrad(self, x: torch.Tensor, dy: torch.Tensor, scale: float):
        side_run(x.device, lambda: self._wgrad(x, dy, scale), x, dy)

    def _wgrad(self, x: torch.Tensor, dy: torch.Tensor, scale: float):
        M = x.shape[0]
        packed, ldo = run_wgrad(dy, x, M, 1, 1, self.np_, 1, 1, self.kp, 1, 1, 0, flops=2.0 * M * self.k_in * self.n_out,
                                hold=_hold_of(self), gate=_gate_of(self))
        emit_grad(self.group, packed, self.wg, self.gspec, ldo, 1.0 / scale)

    def bias_grad(self, dy: torch.Tensor, scale: float):
        if self.bg is not None:
            colsum_acc(dy, dy.shape[0], self.n_out, dy.shape[1], 1, 1.0 / scale, self.bg)


# ------------------------------------------------------------------------------------------------
# batch norm
# ------------------------------------------------------------------------------------------------
# BatchNorm calls over at most this many rows (the BatchNorm1d layers behind the dense layers) run as ONE launch per
# direction (csrc/norm.hip bn_cols_*); 0 switches the path off (A/B timing)
_BN_COLS_ROWS = int(os.environ.get("FMRI_BN_COLS_ROWS", "2048"))


class BNSaved:
    __slots__ = ("mean", "rstd", "scale", "shift", "count", "sums")


class BatchNorm:
    """Train-mode BatchNorm2d/1d(momentum=0.9) (+ReLU) over fp16 rows [M, C].

    ``perm=(C0, HW)``: the C = C0*HW features are stored in engine order (HW, C0) while the reference
    parameter / running-stat vectors are in (C0, HW) order (decoder.fc.1, models/vae_gan.py:108,127).
    ``reducer``: optional callable all-reducing a fp32 tensor in place and returning the world size
    (SyncBN for data-parallel runs).
    """

    def __init__(self, group, prefix: str, C: int, perm=None):
        self.group, self.prefix, self.C, self.perm = group, prefix, C, perm
        self.gamma, self.beta = group.views[prefix + "weight"], group.views[prefix + "bias"]
        self.ggamma, self.gbeta = group.grads[prefix + "weight"], group.grads[prefix + "bias"]
        self.rm, self.rv = group.bufs[prefix + "running_mean"], group.bufs[prefix + "running_var"]
        self.nbt = group.bufs[prefix + "num_batches_tracked"]
        self.reducer = None
        self.eval_mode = False
        self._v = -1
        if perm:
            dev = self.gamma.device
            self.gamma_e = torch.empty(C, dtype=torch.float32, device=dev)
            self.beta_e = torch.empty(C, dtype=torch.float32, device=dev)
            self.rm_e = torch.empty(C, dtype=torch.float32, device=dev)
            self.rv_e = torch.empty(C, dtype=torch.float32, device=dev)

    def _params(self):
        if not self.perm:
            return self.gamma, self.beta, self.rm, self.rv
        if self._v != self.group.version:
            c0, hw = self.perm
            for s, d in ((self.gamma, self.gamma_e), (self.beta, self.beta_e)):
                lib.call("fmri_permute_chw", _P(s), _P(d), c0, hw, 1, 1.0, 0)
            self._v = self.group.version
        return self.gamma_e, self.beta_e, self.rm_e, self.rv_e

    # Running statistics of a (C,H,W)-permuted BatchNorm1d: by default the engine-order copies are refreshed from / written
    # back to the reference-order buffers around every call (4 tiny launches per call: what a module whose buffers alias
    # them needs).  ``enable_lazy_running`` (fused step classes) makes the engine-order copies the live ones: they are
    # loaded when the buffers were written from outside (FlatGroup.buf_version) and written back when somebody reads the
    # buffers (FlatGroup.flush_hooks, i.e. state_dict()).
    def enable_lazy_running(self):
        if self.perm and not getattr(self, "_lazy", False):
            self._lazy, self._run_v, self._run_dirty = True, -1, False
            self.group.flush_hooks.append(self.flush_running)

    def _running_in(self):
        if self.perm:
            if getattr(self, "_lazy", False):
                if self._run_v == self.group.buf_version:
                    return
                self._run_v, self._run_dirty = self.group.buf_version, False
            c0, hw = self.perm
            lib.call("fmri_permute_chw", _P(self.rm), _P(self.rm_e), c0, hw, 1, 1.0, 0)
            lib.call("fmri_permute_chw", _P(self.rv), _P(self.rv_e), c0, hw, 1, 1.0, 0)

    def _running_out(self):
        if self.perm:
            if getattr(self, "_lazy", False):
                self._run_dirty = True
                return
            c0, hw = self.perm
            lib.call("fmri_permute_chw", _P(self.rm_e), _P(self.rm), c0, hw, 0, 1.0, 0)
            lib.call("fmri_permute_chw", _P(self.rv_e), _P(self.rv), c0, hw, 0, 1.0, 0)

    def flush_running(self):
        """Lazy mode: write the engine-order running_mean / running_var back to the reference-order buffers.  Done whenever
        the engine-order copies are the loaded, current ones -- a replayed HIP graph updates them without passing through
        ``_running_out``, so there is no reliable dirty flag."""
        if self.perm and getattr(self, "_lazy", False) and self._run_v == self.group.buf_version:
            c0, hw = self.perm
            lib.call("fmri_permute_chw", _P(self.rm_e), _P(self.rm), c0, hw, 0, 1.0, 0)
            lib.call("fmri_permute_chw", _P(self.rv_e), _P(self.rv), c0, hw, 0, 1.0, 0)
            self._run_dirty = False

    def forward_groups(self, raws, relu: bool, updates: int, outs, stat_accs, order, in_scales=None):
        """Train-mode forward of several calls of this BatchNorm whose inputs are ready together (the decoder's call groups
        of one layer), running-statistics updates in ``order``.  Data parallel with SyncBN: the statistics of all calls are
        exchanged in ONE all-reduce ([G][2][C]) instead of one per call.  ``in_scales``: per call, the device scalar its
        rows are stored range-scaled by (``forward``), or None.  Returns the BNSaved of each call."""
        G = len(raws)
        svs = [None] * G
        ins = in_scales if in_scales is not None else [None] * G
        if self.reducer is None or self.eval_mode:
            for gi in order:
                _, svs[gi] = self.forward(raws[gi], relu, updates, out=outs[gi], stat_acc=stat_accs[gi], in_scale=ins[gi])
            return svs
        C = self.C
        sums_all = torch.empty(G, 2, C, dtype=torch.float32, device=raws[0].device)
        for gi in range(G):
            self._forward_sums(raws[gi], stat_accs[gi], sums_all[gi])
        world = self.reducer(sums_all)
        for gi in order:
            _, svs[gi] = self.forward(raws[gi], relu, updates, out=outs[gi], exchanged=(sums_all[gi], world),
                                      in_scale=ins[gi])
        return svs

    def _forward_sums(self, raw: torch.Tensor, stat_acc, sums: torch.Tensor):
        """sums [2][C] <- (sum x, sum x^2) of ``raw`` (from the producer's statistics rows where given)."""
        C = self.C
        x2 = raw.reshape(-1, C)
        M = x2.shape[0]
        if stat_acc is not None:
            scratch = torch.empty(lib.load().fmri_bn_fold_scratch_floats(C), dtype=torch.float32, device=raw.device)
            lib.call("fmri_bn_fold", _P(stat_acc), stat_acc.shape[0], C, _P(scratch), _P(sums))
        else:
            ws = _reduce_ws(M, C, raw.device)
            lib.note(bytes=2.0 * M * C)
            lib.call("fmri_bn_stats", _P(x2), M, C, _P(sums), _P(ws), ws.numel())

    def forward(self, raw: torch.Tensor, relu: bool = True, updates: int = 1, out: Optional[torch.Tensor] = None,
                stat_acc: Optional[torch.Tensor] = None, exchanged=None, in_scale: Optional[torch.Tensor] = None):
        """``stat_acc``: the batch statistics of ``raw`` as rows [P][2][C] written by the producing kernel's epilogue
        (``ConvLayer.take_stats``): the statistics pass over ``raw`` is skipped.  ``exchanged`` = (sums [2][C] already
        summed over the ranks, world size): ``forward_groups``' second phase.  ``in_scale`` (device fp32 scalar s, a power
        of two): ``raw`` holds s * x (a latent batch stored range-scaled, ``latent_ranged``) -- the result is BatchNorm(x),
        the saved statistics are those of the stored rows, the running statistics the true-scale ones."""
        if self.eval_mode:
            return self.forward_eval(raw, relu, out, in_scale=in_scale), None
        C = self.C
        x2 = raw.reshape(-1, C)
        M = x2.shape[0]
        dev = raw.device
        gamma, beta, rm, rv = self._params()
        sums = torch.empty(2, C, dtype=torch.float32, device=dev) if exchanged is None else exchanged[0]
        ws = _reduce_ws(M, C, dev) if (stat_acc is None and exchanged is None) else None
        count = float(M)
        sv = BNSaved()
        buf = torch.empty(4, C, dtype=torch.float32, device=dev)
        sv.mean, sv.rstd, sv.scale, sv.shift = buf[0], buf[1], buf[2], buf[3]
        if updates > 0:
            self._running_in()
        fin = (_P(gamma), _P(beta), 1e-5, 0.9, updates, _P(rm) if updates > 0 else None,
               _P(rv) if updates > 0 else None, _P(sv.mean), _P(sv.rstd), _P(sv.scale), _P(sv.shift),
               _P(self.nbt) if updates > 0 else None)
        if exchanged is None and stat_acc is None and self.reducer is None and M <= _BN_COLS_ROWS:
            # few rows (the BatchNorm1d layers behind the dense layers): statistics, finalize and apply in ONE launch
            if out is None:
                out = torch.empty_like(raw)
            lib.note(bytes=6.0 * M * C)
            lib.call("fmri_bn_cols_fwd_s", _P(x2), _P(out), M, C, count, _P(gamma), _P(beta), 1e-5, 0.9, updates,
                     _P(rm) if updates > 0 else None, _P(rv) if updates > 0 else None, _P(sv.mean), _P(sv.rstd),
                     _P(sv.scale), _P(sv.shift), _P(sums), _P(self.nbt) if updates > 0 else None, 1 if relu else 0,
                     _P(in_scale))
            sv.count, sv.sums = count, sums
            if updates > 0:
                self._running_out()
            return out, sv
        if exchanged is not None:
            count *= exchanged[1]
            lib.call("fmri_bn_finalize_s", _P(sums), C, count, *fin, _P(in_scale))
        elif in_scale is not None:
            # (many rows of a range-scaled input: statistics, exchange, finalize with the input scale)
            assert stat_acc is None
            lib.note(bytes=2.0 * M * C)
            lib.call("fmri_bn_stats", _P(x2), M, C, _P(sums), _P(ws), ws.numel())
            if self.reducer is not None:
                count *= self.reducer(sums)
            lib.call("fmri_bn_finalize_s", _P(sums), C, count, *fin, _P(in_scale))
        elif stat_acc is not None:
            assert stat_acc.shape[-1] == C and stat_acc.is_contiguous()
            rows = stat_acc.shape[0]
            scratch = torch.empty(lib.load().fmri_bn_fold_scratch_floats(C), dtype=torch.float32, device=dev)
            if self.reducer is None:
                lib.call("fmri_bn_fold_finalize", _P(stat_acc), rows, C, _P(scratch), _P(sums), count, *fin)
            else:
                lib.call("fmri_bn_fold", _P(stat_acc), rows, C, _P(scratch), _P(sums))
                count *= self.reducer(sums)
                lib.call("fmri_bn_finalize", _P(sums), C, count, *fin)
        elif self.reducer is None:
            # no statistics exchange: the fold of the partial sums finalizes (one launch less on the critical path)
            lib.note(bytes=2.0 * M * C)
            lib.call("fmri_bn_stats_finalize", _P(x2), M, C, _P(sums), _P(ws), ws.numel(), count, *fin)
        else:
            lib.note(bytes=2.0 * M * C)
            lib.call("fmri_bn_stats", _P(x2), M, C, _P(sums), _P(ws), ws.numel())
            count *= self.reducer(sums)
            lib.call("fmri_bn_finalize", _P(sums), C, count, *fin)
        sv.count = count
        sv.sums = sums                  # [2][C] sum x, sum x^2 of the batch: update_running_again() re-applies them
        if updates > 0:
            self._running_out()
        if out is None:
            out = torch.empty_like(raw)
        lib.note(bytes=4.0 * M * C)
        lib.call("fmri_bn_apply", _P(x2), _P(out), M, C, _P(sv.scale), _P(sv.shift), 1 if relu else 0)
        return out, sv

    def update_running_again(self, sv: BNSaved, updates: int = 1):
        """One more momentum update of running_mean / running_var / num_batches_tracked with the batch statistics of an
        earlier train-mode ``forward`` (``sv``): what a second forward call on the SAME input does to the module's buffers
        (the reference runs the discriminator's conv stack twice per step on identical inputs, models/vae_gan.py:284-285)
        without recomputing anything else."""
        C = self.C
        gamma, beta, rm, rv = self._params()
        self._running_in()
        junk = torch.empty(4, C, dtype=torch.float32, device=rm.device)
        lib.call("fmri_bn_finalize", _P(sv.sums), C, sv.count, _P(gamma), _P(beta), 1e-5, 0.9, int(updates), _P(rm), _P(rv),
                 _P(junk[0]), _P(junk[1]), _P(junk[2]), _P(junk[3]), _P(self.nbt))
        self._running_out()

    def eval_affine(self, relu: bool = True):
        """(scale, shift, relu) of the eval-mode map y = relu(scale * x + shift) (running statistics): the ``affine``
        argument of ``ConvLayer.forward``."""
        gamma, beta, rm, rv = self._params()
        self._running_in()
        scale = (gamma * torch.rsqrt(rv + 1e-5)).contiguous()
        shift = (beta - rm * scale).contiguous()
        if self.C % 8:                  # the kernels read these vectors in 16-byte groups up to the PADDED channel count
            pad = 8 - self.C % 8
            scale = torch.nn.functional.pad(scale, (0, pad))
            shift = torch.nn.functional.pad(shift, (0, pad))
        return scale, shift, relu

    def forward_eval(self, raw: torch.Tensor, relu: bool = True, out: Optional[torch.Tensor] = None,
                     in_scale: Optional[torch.Tensor] = None):
        """Eval-mode BN (running statistics, models/vae_gan.py:288-297 path): y = relu(gamma*(x-rm)/sqrt(rv+eps)+beta).
        ``in_scale``: ``raw`` holds s * x (see ``forward``)."""
        C = self.C
        x2 = raw.reshape(-1, C)
        gamma, beta, rm, rv = self._params()
        self._running_in()
        scale = gamma * torch.rsqrt(rv + 1e-5)
        shift = beta - rm * scale
        if in_scale is not None:
            scale = scale / in_scale
        if out is None:
            out = torch.empty_like(raw)
        lib.call("fmri_bn_apply", _P(x2), _P(out), x2.shape[0], C, _P(scale), _P(shift), 1 if relu else 0)
        return out

    def _fold_bwd(self, stat, group0: int, groups: int, sums: torch.Tensor, param_scale, param_group: int):
        """Backward statistics rows of a data gradient's epilogue (``ConvLayer.take_bwd_stats``) -> sums [groups][2][C]
        (+ gamma / beta gradients from group ``param_group``)."""
        part, rows = stat
        C = self.C
        cap = part.shape[1]
        scratch = torch.empty(groups * lib.load().fmri_bn_fold_scratch_floats(C), dtype=torch.float32,
                              device=part.device)
        pg = param_scale is not None
        lib.call("fmri_bn_bwd_fold", _P(part[group0]), rows, cap, C, groups, _P(scratch), _P(sums),
                 _P(self.gbeta) if pg else None, _P(self.ggamma) if pg else None, (1.0 / param_scale) if pg else 0.0,
                 int(param_group))

    def backward(self, raw: torch.Tensor, dy: torch.Tensor, sv: BNSaved, relu: bool = True,
                 param_scale: Optional[float] = None, out: Optional[torch.Tensor] = None, stat=None,
                 stat_group: int = 0, sums: Optional[torch.Tensor] = None, phase: int = 3):
        """dx through (ReLU o BN) with batch statistics; if ``param_scale`` is given, gamma/beta grads are
        accumulated as (1/param_scale) * sums.  ``stat``: the reduction already done by the epilogue of the data
        gradient that produced ``dy`` (group ``stat_group`` of ``ConvLayer.take_bwd_stats()``; dy is then ReLU-masked).
        ``phase`` 1 = the reduction only (into ``sums``, a caller-owned [2][C] view), 2 = the apply pass only (``sums``
        already exchanged), 3 = both with this call's own SyncBN exchange in between: a caller with several calls that are
        ready together (the decoder's call groups of one layer) runs all phase-1 calls, ONE exchange, all phase-2 calls."""
        C = self.C
        x2 = raw.reshape(-1, C)
        g2 = dy.reshape(-1, C)
        M = x2.shape[0]
        gamma, beta, _, _ = self._params()
        if sums is None:
            sums = torch.empty(2, C, dtype=torch.float32, device=raw.device)
        # gamma/beta gradients come from the LOCAL sums (the SUM all-reduce of the gradients adds the other ranks); the
        # fold kernel of the reduction accumulates them, the permuted (C,H,W)-ordered BN1d needs the scatter kernel
        direct = param_scale is not None and not self.perm
        if phase == 3 and stat is None and self.reducer is None and M <= _BN_COLS_ROWS:
            # few rows: reduction, parameter gradients and dx in ONE launch
            if out is None:
                out = torch.empty_like(dy)
            lib.note(bytes=10.0 * M * C)
            lib.call("fmri_bn_cols_bwd", _P(x2), _P(g2), _P(out), M, C, 1, sv.count, _P(sv.mean), _P(sv.rstd), _P(gamma),
                     _P(beta), 1 if relu else 0, _P(sums), _P(self.gbeta) if direct else None,
                     _P(self.ggamma) if direct else None, (1.0 / param_scale) if direct else 0.0, 0)
            if param_scale is not None and self.perm:
                self.accumulate_param_grads(sums, param_scale)
            return out, sums
        if phase & 1:
            if stat is not None:
                assert not self.perm
                self._fold_bwd(stat, stat_group, 1, sums, param_scale, 0)
            else:
                ws = _reduce_ws(M, C, raw.device)
                lib.note(bytes=4.0 * M * C)
                lib.call("fmri_bn_bwd_reduce", _P(x2), _P(g2), M, C, _P(sv.mean), _P(sv.rstd), _P(gamma), _P(beta),
                         1 if relu else 0, _P(sums), _P(ws), ws.numel(), _P(self.gbeta) if direct else None,
                         _P(self.ggamma) if direct else None, (1.0 / param_scale) if direct else 0.0)
            if param_scale is not None and self.perm:
                self.accumulate_param_grads(sums, param_scale)
        if phase == 3 and self.reducer is not None:
            self.reducer(sums)
        if not phase & 2:
            return None, sums
        if out is None:
            out = torch.empty_like(dy)
        lib.note(bytes=6.0 * M * C)
        lib.call("fmri_bn_bwd_apply", _P(x2), _P(g2), _P(out), M, C, sv.count, _P(sv.mean), _P(sv.rstd), _P(gamma),
                 _P(beta), 1 if relu else 0, _P(sums))
        return out, sums

    def backward2(self, raw: torch.Tensor, dy2: torch.Tensor, sv: BNSaved, relu: bool = True,
                  param_scale: Optional[float] = None, out: Optional[torch.Tensor] = None, param_stream: int = 0,
                  stat=None, stat_group: int = 0, sums: Optional[torch.Tensor] = None, phase: int = 3):
        """``backward`` for TWO cotangent streams stacked along the rows, ``dy2 = [A rows | B rows]`` (each as many rows
        as ``raw``): the forward tensor, xhat and the ReLU mask are read / computed once for both.  gamma / beta
        gradients (``param_scale``) are taken from ONE stream (``param_stream``: 0 = A, 1 = B).  ``sums`` ([4][C]) /
        ``phase``: as in ``backward``."""
        C = self.C
        x2 = raw.reshape(-1, C)
        g2 = dy2.reshape(-1, C)
        M = x2.shape[0]
        assert g2.shape[0] == 2 * M
        gamma, beta, _, _ = self._params()
        if sums is None:
            sums = torch.empty(4, C, dtype=torch.float32, device=raw.device)
        pg = param_scale is not None and not self.perm     # (C,H,W)-permuted BN1d: scattered below
        if phase == 3 and stat is None and self.reducer is None and M <= _BN_COLS_ROWS:
            if out is None:
                out = torch.empty_like(dy2)
            lib.note(bytes=16.0 * M * C)
            lib.call("fmri_bn_cols_bwd", _P(x2), _P(g2), _P(out), M, C, 2, sv.count, _P(sv.mean), _P(sv.rstd), _P(gamma),
                     _P(beta), 1 if relu else 0, _P(sums), _P(self.gbeta) if pg else None,
                     _P(self.ggamma) if pg else None, (1.0 / param_scale) if pg else 0.0, int(param_stream))
            if param_scale is not None and self.perm:
                self.accumulate_param_grads(sums[2 * int(param_stream):2 * int(param_stream) + 2], param_scale)
            return out, sums
        if phase & 1:
            if stat is not None:
                # groups stat_group, stat_group + 1 of the producing data gradient's epilogue rows (dy2 is ReLU-masked)
                assert not self.perm
                self._fold_bwd(stat, stat_group, 2, sums, param_scale, int(param_stream))
            else:
                ws = torch.empty(2 * lib.load().fmri_bn_ws_floats(M, C), dtype=torch.float32, device=raw.device)
                lib.note(bytes=6.0 * M * C)
                lib.call("fmri_bn_bwd_reduce2", _P(x2), _P(g2), M, C, _P(sv.mean), _P(sv.rstd), _P(gamma), _P(beta),
                         1 if relu else 0, _P(sums), _P(ws), ws.numel(), _P(self.gbeta) if pg else None,
                         _P(self.ggamma) if pg else None, (1.0 / param_scale) if pg else 0.0, int(param_stream))
            if param_scale is not None and self.perm:
                self.accumulate_param_grads(sums[2 * int(param_stream):2 * int(param_stream) + 2], param_scale)
        if phase == 3 and self.reducer is not None:
            self.reducer(sums)
        if not phase & 2:
            return None, sums
        if out is None:
            out = torch.empty_like(dy2)
        lib.note(bytes=10.0 * M * C)
        lib.call("fmri_bn_bwd_apply2", _P(x2), _P(g2), _P(out), M, C, sv.count, _P(sv.mean), _P(sv.rstd), _P(gamma),
                 _P(beta), 1 if relu else 0, _P(sums))
        return out, sums

    def accumulate_param_grads(self, sums: torch.Tensor, scale: float):
        inv = 1.0 / scale
        if self.perm:
            c0, hw = self.perm
            lib.call("fmri_permute_chw", _P(sums[1]), _P(self.ggamma), c0, hw, 0, inv, 1)
            lib.call("fmri_permute_chw", _P(sums[0]), _P(self.gbeta), c0, hw, 0, inv, 1)
        else:
            self.ggamma.add_(sums[1], alpha=inv)
            self.gbeta.add_(sums[0], alpha=inv)


# ------------------------------------------------------------------------------------------------
# misc helpers
# ------------------------------------------------------------------------------------------------
def images_to_nhwc(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """NCHW fp32 -> NHWC fp16 (channels padded to 8)."""
    require_gpu(x)
    N, C, H, W = x.shape
    x = x.contiguous().float()
    cp = pad8(C)
    if out is None:
        out = torch.empty(N, H, W, cp, dtype=torch.float16, device=x.device)
    lib.call("fmri_nchw_to_nhwc", _P(x), _P(out), N, C, H * W, cp)
    return out


def nhwc_to_images(x16: torch.Tensor, C: int, scale: float = 1.0) -> torch.Tensor:
    N, H, W, cp = x16.shape
    out = torch.empty(N, C, H, W, dtype=torch.float32, device=x16.device)
    lib.call("fmri_nhwc_to_nchw", _P(x16), _P(out), N, C, H * W, cp, float(scale))
    return out


def rows_to_f16(x: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    require_gpu(x)
    M, Cc = x.shape
    x = x.contiguous().float()
    out = torch.empty(M, pad8(Cc), dtype=torch.float16, device=x.device)
    lib.call("fmri_rows_f32_to_f16", _P(x), _P(out), M, Cc, pad8(Cc), float(scale))
    return out


# largest |z| a latent batch is stored with at scale 1 (``latent_ranged``): above it the fp16 rows carry 2^-k * z
LATENT_CAP = 256.0


def latent_ranged(head32: Optional[torch.Tensor], eps: Optional[torch.Tensor], rows: int, Z: int, z16: torch.Tensor,
                  zmax: torch.Tensor, zscale: torch.Tensor, kl_total: Optional[torch.Tensor] = None, sample: bool = True,
                  z32: Optional[torch.Tensor] = None, max_reduce=None):
    """Range-safe reparameterisation (fmri_latent_fwd_ranged): z = eps * exp(0.5 logvar) + mu (or mu) in fp32, then
    ``z16`` [rows, zp] = fp16(s * z) with s = the largest power of two <= 1 that brings max |z| of the batch under
    LATENT_CAP, written to the device scalar ``zscale`` -- the ``zscale`` argument of ``DecoderNet.forward``.  ``zmax``:
    device scalar, zero on entry.  ``head32`` None: ``z32`` is a caller-provided fp32 latent [rows, Z].  ``max_reduce``
    (data parallel with SyncBN: the decoder's batch is the global one): callable all-reducing ``zmax`` with MAX."""
    zp = z16.shape[1]
    if head32 is not None:
        if z32 is None:
            z32 = torch.empty(rows, Z, dtype=torch.float32, device=z16.device)
        phase = 3 if max_reduce is None else 1
        lib.call("fmri_latent_fwd_ranged", _P(head32), _P(eps), rows, Z, zp, _P(z16), None, _P(kl_total),
                 1 if sample else 0, _P(z32), _P(zmax), _P(zscale), LATENT_CAP, phase)
        if max_reduce is None:
            return z32
    else:
        lib.call("fmri_rows_absmax", _P(z32), rows * Z, _P(zmax))
    if max_reduce is not None:
        max_reduce(zmax)
    lib.call("fmri_latent_fwd_ranged", None, None, rows, Z, zp, _P(z16), None, None, 0, _P(z32), _P(zmax), _P(zscale),
             LATENT_CAP, 2)
    return z32


def rows_to_f16_ranged(z: torch.Tensor):
    """fp32 latent rows [M, Z] -> (fp16 [M, pad8(Z)] stored at scale s, s as a device scalar) -- ``rows_to_f16`` for a
    latent of unknown range (the module API's ``Decoder(z)``)."""
    require_gpu(z)
    M, Z = z.shape
    z = z.contiguous().float()
    z16 = torch.empty(M, pad8(Z), dtype=torch.float16, device=z.device)
    st = torch.zeros(2, dtype=torch.float32, device=z.device)
    latent_ranged(None, None, M, Z, z16, st[0:1], st[1:2], z32=z)
    return z16, st[1:2]


def _reduce_ws(M: int, C: int, device) -> torch.Tensor:
    """Per-block partial-sum workspace of the BN / column-sum reductions."""
    return torch.empty(lib.load().fmri_bn_ws_floats(M, C), dtype=torch.float32, device=device)


def act_backward(y: torch.Tensor, dy: torch.Tensor, act: int, colsum: Optional[torch.Tensor] = None,
                 out: Optional[torch.Tensor] = None, dbias: Optional[torch.Tensor] = None,
                 dbias_scale: float = 1.0) -> torch.Tensor:
    """dpre = dy * act'(y).  ``colsum`` (fp32, >= 2*C floats): its first C entries receive sum_rows(dpre); ``dbias``
    (fp32, needs ``colsum``): dbias[:n] += dbias_scale * sum_rows(dpre)[:n], n = min(C, len(dbias)), in the same launches."""
    C = y.shape[-1]
    M = y.numel() // C
    if out is None:
        out = torch.empty_like(dy)
    lib.note(bytes=6.0 * M * C)
    if colsum is not None:
        assert colsum.numel() >= 2 * C
        ws = _reduce_ws(M, C, y.device)
        lib.call("fmri_act_bwd", _P(y), _P(dy), _P(out), M, C, act, _P(colsum), _P(ws), ws.numel(), _P(dbias),
                 min(C, dbias.numel()) if dbias is not None else 0, float(dbias_scale))
    else:
        lib.call("fmri_act_bwd", _P(y), _P(dy), _P(out), M, C, act, None, None, 0, None, 0, 0.0)
    return out


def colsum_acc(src: torch.Tensor, M: int, C: int, ld_row: int, ld_col: int, scale: float, dst: torch.Tensor):
    """dst[c] += scale * sum_{m < M} src[m*ld_row + c*ld_col] (bias gradients).  A few hundred rows (dense layers, the
    narrow weight-gradient kernel's slabs): one small launch (csrc/layout.hip); many contiguous fp16 rows (a conv
    cotangent): the BatchNorm statistics reduction with the accumulation in its fold (csrc/norm.hip)."""
    if M >= 2048 and src.dtype == torch.float16 and ld_col == 1 and ld_row % 8 == 0 and src.is_contiguous():
        ws = _reduce_ws(M, ld_row, src.device)
        sums = torch.empty(2, ld_row, dtype=torch.float32, device=src.device)
        lib.note(bytes=2.0 * M * ld_row)
        lib.call("fmri_colsum_rows", _P(src), M, ld_row, _P(sums), _P(ws), ws.numel(), _P(dst), C, float(scale))
        return
    lib.call("fmri_colsum_acc", _P(src), 1 if src.dtype == torch.float16 else 0, M, C, ld_row, ld_col, float(scale),
             _P(dst))


def axpby(x: torch.Tensor, y: Optional[torch.Tensor], a: float, b: float, out: Optional[torch.Tensor] = None,
          a_dev: Optional[torch.Tensor] = None, b_dev: Optional[torch.Tensor] = None):
    """out = a * (*a_dev) * x + b * (*b_dev) * y  (fp16 tensors; a_dev / b_dev: optional device fp32 scalars)."""
    if out is None:
        out = torch.empty_like(x)
    lib.call("fmri_axpby2_f16", _P(x), _P(y), _P(out), x.numel(), float(a), float(b), _P(a_dev), _P(b_dev))
    return out


def ingest_u8(images_u8: torch.Tensor, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5), flip: Optional[torch.Tensor] = None,
              shift: Optional[torch.Tensor] = None, want16: bool = True, want32: bool = False):
    """uint8 [N,H,W,C] (C = 1 or 3, device) -> (fp16 NHWC8 engine input or None, fp32 NCHW or None): per-image
    horizontal flip (int32 [N]) and integer shift (int32 [N,2] = rows, cols; edge replicated), /255, grey -> RGB,
    (v - mean) / std in one pass (csrc/ingest.hip)."""
    require_gpu(images_u8)
    assert images_u8.dtype == torch.uint8 and images_u8.dim() == 4 and images_u8.is_contiguous()
    N, H, W, C = images_u8.shape
    dev = images_u8.device
    o16 = torch.empty(N, H, W, 8, dtype=torch.float16, device=dev) if want16 else None
    o32 = torch.empty(N, 3, H, W, dtype=torch.float32, device=dev) if want32 else None
    if flip is not None:
        flip = flip.to(device=dev, dtype=torch.int32).contiguous()
    if shift is not None:
        shift = shift.to(device=dev, dtype=torch.int32).contiguous()
    lib.call("fmri_ingest_u8", _P(images_u8), N, H, W, C, _P(flip), _P(shift), float(mean[0]), float(mean[1]),
             float(mean[2]), float(std[0]), float(std[1]), float(std[2]), _P(o16), _P(o32))
    return o16, o32


# ------------------------------------------------------------------------------------------------
# head of the image pipeline: CenterCrop + Resize on the device, pinned double-buffered staging
# ------------------------------------------------------------------------------------------------
_RESIZE_TABLES = {}


def resize_tables(crop: int, size: int, device):
    """Device coefficient tables of one Pillow BILINEAR pass ``crop`` -> ``size`` (fmri_resize_coeffs), cached.
    Returns (bounds int32 [size][2], coef int32 [size][ksize], ksize, largest tap count); ksize == 0: identity pass."""
    key = (crop, size, str(device))
    t = _RESIZE_TABLES.get(key)
    if t is None:
        import numpy as np
        L = lib.load()
        cap = 2 * int(-(-max(crop, size) // size)) + 1
        b = np.zeros((size, 2), np.int32)
        k = np.zeros((size, cap), np.int32)
        ks = L.fmri_resize_coeffs(crop, size, b.ctypes.data, k.ctypes.data, cap)
        if ks < 0:
            lib.check(ks, "fmri_resize_coeffs")
        if ks == 0:
            t = (None, None, 0, 1)
        else:
            t = (torch.from_numpy(b).to(device), torch.from_numpy(np.ascontiguousarray(k[:, :ks].reshape(size, ks))).to(device),
                 ks, int(b[:, 1].max()))
        _RESIZE_TABLES[key] = t
    return t


def crop_resize_u8(pool: torch.Tensor, offsets: torch.Tensor, dims: torch.Tensor, crop: int, size: int,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """CenterCrop((crop, crop)) + Resize((size, size)) of a ragged batch of decoded images, bit-exact with the
    torchvision 0.5.0 / PIL transforms of train_vgan_stage1.py:162-165 (csrc/ingest.hip).  ``pool``: uint8 device
    buffer with the images back to back (HWC); ``offsets`` int64 [N] byte offsets; ``dims`` int32 [N][3] = (H, W, C),
    C = 1 or 3.  Returns uint8 [N][size][size][3] (grey replicated): the input of ``ingest_u8``."""
    require_gpu(pool)
    assert pool.dtype == torch.uint8 and offsets.dtype == torch.int64 and dims.dtype == torch.int32
    N = offsets.numel()
    assert dims.shape == (N, 3) and offsets.is_cuda and dims.is_cuda
    b, k, ks, vmax = resize_tables(crop, size, pool.device)
    if out is None:
        out = torch.empty(N, size, size, 3, dtype=torch.uint8, device=pool.device)
    lib.call("fmri_crop_resize_u8", _P(pool), _P(offsets), _P(dims.contiguous()), N, crop, size, _P(b), _P(k), ks,
             _P(b), _P(k), ks, vmax, _P(out))
    return out


class HostStager:
    """Pinned-memory double-buffered host -> device staging of decoded image batches (replaces the transform chain the
    reference runs inside its DataLoader workers, train/train_vgan_stage1.py:162-170,195): the loader threads only
    decode; ``submit`` packs a ragged batch into one pinned buffer, copies it on a SIDE stream and runs crop + resize +
    ingest (flip / shift / ToTensor / GreyToColor / Normalize) there, so that batch i + 1 is staged while the training
    step of batch i runs on the main stream.  ``depth`` buffers rotate; a buffer is re-used only after the device
    work that read it has completed (event), so ``submit`` never overwrites bytes still being copied.

        stager = HostStager("cuda:0", crop=375, size=64)
        t = stager.submit(images)                 # list of uint8 HWC numpy arrays (C = 1 or 3), any sizes
        x16, x32 = stager.take(t)                 # main stream waits for the side stream's event, no host sync
    """

    def __init__(self, device, crop: int, size: int, depth: int = 2, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5),
                 capacity: int = 1 << 26):
        self.device = torch.device(device)
        self.crop, self.size, self.mean, self.std = crop, size, mean, std
        self.depth = depth
        self.stream = torch.cuda.Stream(self.device)
        self.pinned = [torch.empty(capacity, dtype=torch.uint8).pin_memory() for _ in range(depth)]
        self.dev = [torch.empty(capacity, dtype=torch.uint8, device=self.device) for _ in range(depth)]
        self.free_evt = [None] * depth            # recorded when the side stream is done with slot i
        # pinned per-slot metadata (byte offsets, (H, W, C) per image), grown on demand: no pin_memory() per submit
        self.meta = [None] * depth
        self.slot = 0
        self.tickets = {}
        self.next_ticket = 0

    def submit(self, images, flip=None, shift=None, want16: bool = True, want32: bool = False) -> int:
        import numpy as np
        i = self.slot
        self.slot = (self.slot + 1) % self.depth
        if self.free_evt[i] is not None:
            self.free_evt[i].synchronize()        # the copy / kernels that read this slot have finished
        offs, dims, pos = [], [], 0
        host = self.pinned[i].numpy()
        for img in images:
            a = np.ascontiguousarray(img)
            if a.ndim == 2:
                a = a[:, :, None]
            assert a.dtype == np.uint8 and a.shape[2] in (1, 3)
            nb = a.size
            if pos + nb > host.size:
                raise ValueError("HostStager: batch exceeds the staging capacity")
            host[pos:pos + nb] = a.reshape(-1)
            offs.append(pos)
            dims.append(a.shape)
            pos += (nb + 15) // 16 * 16
        N = len(offs)
        slot_meta = self.meta[i]
        if slot_meta is None or slot_meta[0].numel() < N:
            cap = max(N, 256)
            slot_meta = self.meta[i] = (torch.empty(cap, dtype=torch.int64).pin_memory(),
                                        torch.empty(cap, 3, dtype=torch.int32).pin_memory())
        meta, dm = slot_meta[0][:N], slot_meta[1][:N]
        meta.copy_(torch.tensor(offs, dtype=torch.int64))
        dm.copy_(torch.tensor(dims, dtype=torch.int32))
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)              # allocations made below are ordered after the caller's frees
        with torch.cuda.stream(self.stream):
            self.dev[i][:pos].copy_(self.pinned[i][:pos], non_blocking=True)
            offs_d = meta.to(self.device, non_blocking=True)
            dims_d = dm.to(self.device, non_blocking=True)
            u8 = crop_resize_u8(self.dev[i], offs_d, dims_d, self.crop, self.size)
            x16, x32 = ingest_u8(u8, self.mean, self.std, flip, shift, want16, want32)
            # caller-allocated per-image draws: read here on the staging stream -- tell the allocator, or their storage
            # could be handed to later work of the caller's stream while these kernels still read it
            for t in (flip, shift):
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(self.stream)
            done = torch.cuda.Event()
            done.record(self.stream)
        self.free_evt[i] = done
        t = self.next_ticket
        self.next_ticket += 1
        self.tickets[t] = (x16, x32, done, (meta, dm, offs_d, dims_d, u8))
        return t

    def take(self, ticket: int):
        """(fp16 NHWC8 engine input or None, fp32 NCHW module input or None) of a submitted batch; the CURRENT stream
        waits for the staging work (device-side wait: the host does not block)."""
        x16, x32, done, keep = self.tickets.pop(ticket)
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(done)
        for t in (x16, x32) + keep[2:]:
            if t is not None:
                t.record_stream(cur)
        return x16, x32
