"""Engine-vs-engine comparisons (HIP engine against itself in another launch mode), collected AFTER every
HIP-vs-oracle / HIP-vs-golden test (marker ``selfcheck``, tests/conftest.py).

They run in the engine's deterministic-reduction mode (``fmri_hip.ops.set_deterministic``: per-split slabs summed in
slab order for every weight gradient, one-block launches of the scalar loss sums, no fp32 atomics anywhere on the
step), where two executions of the same step are bit-identical -- so every comparison below is EXACT: a launch mode that
back-propagates the wrong batch, skips an update, reads stale fp16 weights or races with the side stream cannot hide
inside a tolerance.  One test keeps the default (atomic) reductions and the element-count criterion of round 3.
"""
import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.selfcheck]
DEV = "cuda:0"
LOSS_KEYS = ("loss_encoder", "loss_decoder", "loss_discriminator", "nle", "kl", "mse", "bce_orig", "bce_pred", "bce_samp")


def _assert_same_bits(sa, sb, what=""):
    """Every tensor of two state dicts (or dicts of tensors) equal bit for bit."""
    assert sa.keys() == sb.keys()
    for k in sa:
        a, b = sa[k], sb[k]
        if not torch.equal(a, b):
            d = (a.double() - b.double()).abs()
            raise AssertionError(f"{what}: {k} differs in {int((d > 0).sum())} of {a.numel()} elements, "
                                 f"max |diff| {float(d.max()):.3e}")


def _assert_same_logs(la, lb, what=""):
    for k in la:
        assert la[k] == lb[k], (what, k, la[k], lb[k])


def _same_update(sa, sb, what=""):
    """DEFAULT (atomic) reductions: two runs of the same step leave the same parameters up to the run-to-run spread of
    the fp32 atomics in the weight-gradient sums.  RMSprop's first update is lr*g/(sqrt(0.1 g^2)+1e-8): +-3.16e-4 for
    every element whose gradient is well above 1e-8, proportional to g below that -- so the spread shows as up to ~1e-1
    of a step on elements with near-zero gradients and as single elements whose gradient changes sign (a full 6.3e-4).
    A wrong or missing update moves (nearly) EVERY element of a tensor by a step.  Hence: elements may differ by a
    quarter step (8e-5; or 2e-5 of the tensor's largest entry, for the running statistics), and at most max(4, 1 %)
    of a tensor's elements by more."""
    for k in sa:
        a, b = sa[k].float().cpu().reshape(-1), sb[k].float().cpu().reshape(-1)
        lim = max(2e-5 * float(b.abs().max()), 8e-5)
        bad = int(((a - b).abs() > lim).sum())
        assert bad <= max(4, a.numel() // 100), (what, k, bad, a.numel(), float((a - b).abs().max()))


def _stage1_pair(B, seed=0, hp=None):
    """Two Stage-I engines with the same parameters and optimizer state + one seeded batch."""
    from oracle import vaegan_oracle as O
    from fmri_hip.params import ArchConfig
    from fmri_hip.steps import Stage1Step
    data = O.synth_batch(B, O.ArchCfg.px64(), seed=1234, steps=1)
    x, e, zp = data["x"].to(DEV), data["noise"][0, 0].to(DEV), data["noise"][0, 1].to(DEV)
    a = Stage1Step(ArchConfig.px64(), DEV, hp=hp)
    a.load_recipe(seed, True)
    b = Stage1Step(ArchConfig.px64(), DEV, hp=hp)
    b.load_state_dict(a.state_dict())
    return a, b, (x, e, zp), data


def _sync_optimizers(a, b):
    for n in ("opt_enc", "opt_dec", "opt_dis"):
        getattr(b, n).s1.copy_(getattr(a, n).s1)


def _finish():
    from fmri_hip import ops
    ops.join_side()
    torch.cuda.synchronize()


@pytest.mark.parametrize("B", [8, 256])
def test_two_runs_of_a_step_are_bit_identical(deterministic, B):
    """The property the other tests of this file stand on: in deterministic mode two engines fed the same batch leave
    the same bits after three steps -- at B = 8 (generic split-K weight gradients as slabs) and at BASELINE configs[1]'s
    B = 256 (window kernel with many K pieces as slabs, the narrow 5x5 kernel with one slab per block)."""
    a, b, args, _ = _stage1_pair(B)
    for _ in range(3):
        a.step(*args)
        b.step(*args)
    _finish()
    _assert_same_logs(a.logs(), b.logs(), f"B={B}")
    _assert_same_bits(a.state_dict(), b.state_dict(), f"two runs, B={B}")


def test_default_mode_runs_agree_to_the_atomic_spread():
    """The DEFAULT reductions (fp32 atomics in the many-split weight gradients): one step of two engines on the same
    batch agrees to the element-count criterion of ``_same_update`` and to 1e-5 on the logged losses."""
    a, b, args, _ = _stage1_pair(8)
    a.step(*args)
    b.step(*args)
    _finish()
    la, lb = a.logs(), b.logs()
    for k in LOSS_KEYS:
        assert abs(la[k] - lb[k]) <= 1e-5 * abs(lb[k]), (k, la[k], lb[k])
    _same_update(a.state_dict(), b.state_dict(), "default mode, two runs")


def test_fused_step_equals_separate_calls(deterministic):
    """``Stage1Step.step`` (weight gradients, discriminator / decoder optimizer updates and weight repacks queued on the
    side stream under the rest of the backward pass) against the same step issued as forward / gate / backward / apply
    on one stream: the same bits after two steps."""
    from fmri_hip import ops
    a, b, args, _ = _stage1_pair(8)
    side_was = ops._SIDE["on"]
    try:
        for _ in range(2):
            ops._SIDE["on"] = True
            a.step(*args)
            _finish()
            ops._SIDE["on"] = False
            b.forward(*args)
            b.gate(8)
            b.backward()
            b.apply()
            _finish()
    finally:
        ops._SIDE["on"] = side_was
    _assert_same_logs(a.logs(), b.logs(), "fused vs separate")
    _assert_same_bits(a.state_dict(), b.state_dict(), "fused vs separate")


@pytest.mark.parametrize("B", [8, 256])
def test_one_launch_update_equals_separate_launches(deterministic, B):
    """fmri_apply_batch (``Stage1Step.step``: weight gradients stay in their GEMM layout until ONE launch per sub-network
    sums their slabs, maps them to the reference layout, runs RMSprop and writes the fp16 GEMM copy) against
    fmri_unpack_grad + fmri_rmsprop_dev + fmri_pack_weight* (``backward()`` / ``apply()``): after three steps the same
    bits in the parameters, in the RMSprop state and in EVERY fp16 GEMM copy of both orientations -- at B = 8 and at
    BASELINE configs[1]'s B = 256 (other weight-gradient kernels, other slab counts)."""
    from fmri_hip import ops
    from fmri_hip.nets import refresh_net
    from fmri_hip.steps import GanHyper
    # a margin nothing reaches: the gate trains all three sub-networks in EVERY step, so every gradient buffer of the
    # one-launch path is consumed and refilled three times (with the default margin the B = 256 discriminator is switched
    # off after the first step, and a buffer that kept part of an earlier step's sums would go unnoticed)
    a, b, args, _ = _stage1_pair(B, hp=GanHyper(margin=10.0))
    fused = 0
    real = ops.apply_group
    def counting(*aa, **kw):
        nonlocal fused
        r = real(*aa, **kw)
        fused += 1 if r else 0
        return r
    ops.apply_group = counting
    try:
        for _ in range(3):
            a.step(*args)
            b.forward(*args)
            b.gate(B)
            b.backward()
            b.apply()
            _finish()
            la = a.logs()
            assert la["train_dis"] and la["train_dec"], "the gate switched a sub-network off: the test lost its point"
    finally:
        ops.apply_group = real
    assert fused == 9, f"the one-launch path ran {fused} times in 3 steps x 3 sub-networks"
    _assert_same_logs(a.logs(), b.logs(), "one launch vs separate")
    _assert_same_bits(a.state_dict(), b.state_dict(), "one launch vs separate")
    for n in ("opt_enc", "opt_dec", "opt_dis"):
        assert torch.equal(getattr(a, n).s1, getattr(b, n).s1), f"{n}: RMSprop state differs"
    for na, nb in ((a.enc, b.enc), (a.dec, b.dec), (a.dis, b.dis)):
        refresh_net(na)
        refresh_net(nb)
        assert len(na.group.packed) == len(nb.group.packed)
        for i, (pa, pb) in enumerate(zip(na.group.packed, nb.group.packed)):
            assert torch.equal(pa.buf, pb.buf), f"{type(na).__name__}: fp16 GEMM copy {i} differs"
    # and the gradients of a following backward() are the reference-layout ones again
    a.forward(*args); a.gate(B); a.backward()
    b.forward(*args); b.gate(B); b.backward()
    _finish()
    for na, nb in ((a.enc, b.enc), (a.dec, b.dec), (a.dis, b.dis)):
        assert torch.equal(na.group.grad, nb.group.grad), f"{type(na).__name__}: gradients after a fused step differ"


def test_hybrid_recorded_forward_step_equals_eager_step(deterministic):
    """``Stage1Step.capture_forward``: forward + gate replayed from a HIP graph, backward / updates issued eagerly on two
    streams -- against the plain ``step`` of a second engine started from the same parameters and RMSprop state: the
    same bits after each of three steps (every replay must see the weights the previous one's early updates produced)."""
    a, b, args, _ = _stage1_pair(8)
    run = a.capture_forward(*args, warmup=1)
    b.load_state_dict(a.state_dict())
    _sync_optimizers(a, b)
    for it in range(3):
        run()
        b.step(*args)
        _finish()
        _assert_same_logs(a.logs(), b.logs(), f"hybrid vs eager, step {it}")
        _assert_same_bits(a.state_dict(), b.state_dict(), f"hybrid vs eager, step {it}")


def test_recorded_forward_survives_an_eager_step_in_between(deterministic):
    """capture_forward(): an eager step() at another batch size between two run() calls (the last, partial batch of an
    epoch) must not leave run() back-propagating the eager batch (it rebinds the recorded forward's tensors).  After the
    eager B = 4 step ONE run() is compared with ONE eager step of an engine that made the same three steps eagerly: a
    run() that back-propagated the eager batch's tensors would differ in every element."""
    a, b, args, _ = _stage1_pair(8)
    x, e, zp = args
    run = a.capture_forward(*args, warmup=1)
    b.load_state_dict(a.state_dict())
    _sync_optimizers(a, b)
    for what in ("run", "eager", "run"):
        if what == "run":
            run()
            b.step(*args)
        else:
            a.step(x[:4], e[:4], zp[:4])
            b.step(x[:4], e[:4], zp[:4])
        _finish()
        _assert_same_logs(a.logs(), b.logs(), f"after {what}")
        _assert_same_bits(a.state_dict(), b.state_dict(), f"after {what}")


def test_recorded_step_follows_hyper_parameter_schedule(deterministic):
    """lr, lambda, equilibrium and margin live in device memory: two replays of ONE captured step with the epoch-end
    updates of train_vgan_stage1.py:448-458 applied in between equal two eagerly issued steps with the same schedule
    (bit for bit); the gate of the second replay is the ORACLE's gate under the new equilibrium / margin and the size of
    its update is the oracle's under the new learning rate."""
    from oracle import vaegan_oracle as O
    from fmri_hip import ops
    a, b, args, data = _stage1_pair(8)
    cfg_o = O.ArchCfg.px64()
    # "epoch end": lr halved, lambda x 100, equilibrium far above every bce mean -> train_dis = False, train_dec = True
    sched = dict(lr=0.5e-4, margin=0.01, equilibrium=10.0, lambda_mse=1e-4)
    run = a.capture(*args, warmup=1)                  # one real (warm-up) step, then the recording (executes nothing)
    b.load_state_dict(a.state_dict())
    _sync_optimizers(a, b)
    flags, before = [], None
    for it in range(2):
        if it == 1:
            a.set_hyper(**sched)
            b.set_hyper(**sched)
            before = {k: v.clone() for k, v in a.state_dict().items()}
        run()
        side_was = ops._SIDE["on"]
        ops._SIDE["on"] = False                       # capture() records a one-stream step
        try:
            b.step(*args)
        finally:
            ops._SIDE["on"] = side_was
        _finish()
        la, lb = a.logs(), b.logs()
        flags.append((la["train_dis"], la["train_dec"]))
        _assert_same_logs(la, lb, f"replay vs eager, step {it}")
        _assert_same_bits(a.state_dict(), b.state_dict(), f"replay vs eager, step {it}")
    assert flags[1] == (False, True), flags
    after = a.state_dict()
    # oracle: warm-up step, default step, scheduled step
    P = O.fill_state(O.vaegan_spec(cfg_o), 0, True)
    opts = {n: O.OptState(kind="rmsprop", lr=1e-4) for n in ("encoder", "decoder", "discriminator")}
    hp = O.GanHyper()
    oargs = (data["x"], data["noise"][0, 0], data["noise"][0, 1], cfg_o)
    O.stage1_step(P, opts, *oargs, hp=hp)
    ref = O.stage1_step(P, opts, *oargs, hp=hp)
    assert flags[0] == (ref["logs"]["train_dis"], ref["logs"]["train_dec"]), flags
    hp = O.GanHyper(lr=sched["lr"], lambda_mse=sched["lambda_mse"], margin=sched["margin"],
                    equilibrium=sched["equilibrium"])
    for o in opts.values():
        o.lr = sched["lr"]
    P2 = {k: v.clone() for k, v in P.items()}
    ref = O.stage1_step(P, opts, *oargs, hp=hp)
    assert flags[1] == (ref["logs"]["train_dis"], ref["logs"]["train_dec"]), flags
    for k in ("encoder.fc.0.weight", "decoder.conv.0.conv.weight", "discriminator.conv.2.conv.weight"):
        du_e = (after[k].float().cpu() - before[k].float().cpu()).norm().item()
        du_o = (P[k] - P2[k]).norm().item()
        print(k, du_e, du_o)
        if k.startswith("discriminator."):
            assert du_e == 0.0 and du_o == 0.0, k          # gated off by the new equilibrium
        else:
            assert abs(du_e - du_o) < 0.1 * du_o, (k, du_e, du_o)    # half the step of lr = 1e-4


def test_decoder_fc_running_statistics_lazy_shadow_round_trips(deterministic):
    """decoder.fc.1 (the (C,H,W)-permuted BatchNorm1d) keeps its running statistics in engine order inside the fused steps
    and writes them back only when the state dict is read: state_dict() before any step returns what was loaded, after
    steps (eager and replayed from a HIP graph) exactly what an eagerly synchronised BatchNorm holds, and
    load_state_dict() in between reaches the next step."""
    from fmri_hip import ops
    a, b, args, _ = _stage1_pair(4)
    keys = ("decoder.fc.1.running_mean", "decoder.fc.1.running_var")
    sd0 = a.state_dict()
    assert float(sd0[keys[0]].abs().max()) == 0.0 and float((sd0[keys[1]] - 1.0).abs().max()) == 0.0   # as loaded
    b.dec.fc_bn._lazy = False                         # reference behaviour: synchronised around every call
    a.step(*args)
    b.step(*args)
    _finish()
    _assert_same_bits(a.state_dict(), b.state_dict(), "lazy vs synchronised, eager step")
    # an outside write of the buffers reaches the next step
    sd = a.state_dict()
    sd[keys[0]] = torch.full_like(sd[keys[0]], 3.0)
    a.load_state_dict(sd)
    b.load_state_dict(sd)
    _sync_optimizers(a, b)
    run = a.capture(*args, warmup=1)
    run()
    side_was, ops._SIDE["on"] = ops._SIDE["on"], False
    try:
        b.step(*args)
        b.step(*args)
    finally:
        ops._SIDE["on"] = side_was
    _finish()
    sa, sb = a.state_dict(), b.state_dict()
    assert float((sa[keys[0]] - 3.0).abs().max()) > 0.0          # the statistics moved on from the written value
    _assert_same_bits(sa, sb, "lazy vs synchronised, replayed steps")


@pytest.mark.parametrize("stage", [1, 3])
def test_wae_step_recorded_into_a_hip_graph_equals_eager_steps(deterministic, stage):
    """WaeStep.capture: the whole step (Adam with its step count on the device, the fused latent-discriminator kernels)
    replayed from a HIP graph equals the eagerly issued steps bit for bit -- 2 warm-up + 3 replayed steps against 5
    eager ones."""
    from fmri_hip.params import ArchConfig
    from fmri_hip.wae_steps import WaeStep
    cfg, V, B = ArchConfig.px64(), 512, 8
    rs = np.random.RandomState(11)
    x = torch.tanh(torch.from_numpy(rs.standard_normal((B, 3, 64, 64)).astype(np.float32))).to(DEV)
    zf = torch.from_numpy(rs.standard_normal((B, cfg.latent_dim)).astype(np.float32)).to(DEV)
    fm = torch.from_numpy(rs.standard_normal((B, V)).astype(np.float32)).to(DEV)
    args = (x, zf) if stage == 1 else (x, None, fm)

    def make():
        st = WaeStep(cfg, DEV, stage, V if stage > 1 else 0)
        st.load_recipe(5, False if stage == 1 else None)
        return st
    a, b = make(), make()
    s0 = {k: v.clone() for k, v in a.state_dict().items()}
    for _ in range(5):
        a.step(*args)                 # two streams (weight gradients on the side stream)
    run = b.capture(*args)            # two eager warm-up steps inside; records a one-stream step
    for _ in range(3):
        run()
    _finish()
    _assert_same_logs(a.logs(), b.logs(), f"wae stage {stage}")
    sa, sb = a.state_dict(), b.state_dict()
    _assert_same_bits(sa, sb, f"wae stage {stage}, graph vs eager")
    moved = [k for k in sa if sa[k].dtype.is_floating_point and sa[k].numel() >= 1024 and "running" not in k
             and not torch.equal(sa[k], s0[k])]
    assert moved, "no parameter moved in five steps"


@pytest.mark.parametrize("kind", ["stage2", "stage3", "wae2", "dual1"])
def test_other_steps_two_runs_are_bit_identical(deterministic, kind):
    """The determinism property for the steps BASELINE configs[2..4] run: Stage II / III (cognitive encoder, teacher,
    frozen sub-networks, gradient clamp), WAE Stage II (Adam, fused latent-discriminator kernels) and the Dual WAE +
    VAE/GAN step -- two engines, the same batch, three steps, the same bits."""
    from oracle import vaegan_oracle as O
    from fmri_hip.params import ArchConfig
    from fmri_hip.steps import CognitiveStep
    from fmri_hip.wae_steps import DualStage1Step, WaeStep
    cfg, cfg_o, B, V = ArchConfig.px64(), O.ArchCfg.px64(), 8, 512
    data = O.synth_batch(B, cfg_o, n_voxels=V, seed=4321, steps=1)
    x, fm = data["x"].to(DEV), data["fmri"].to(DEV)
    nz = [t.to(DEV) for t in data["noise"][0]]

    def make():
        if kind in ("stage2", "stage3"):
            st = CognitiveStep(cfg, V, DEV, 2 if kind == "stage2" else 3)
            st.load_recipe(3, True)
            return st, (lambda: st.step(fm, x, nz[0], nz[1], nz[2]))
        if kind == "wae2":
            st = WaeStep(cfg, DEV, 2, V)
            st.load_recipe(5, None)
            return st, (lambda: st.step(x, None, fm))
        st = DualStage1Step(cfg, DEV)
        st.load_recipe(8, True)
        return st, (lambda: st.step(x, nz[0], nz[1], nz[2]))
    (a, run_a), (b, run_b) = make(), make()
    for _ in range(3):
        run_a()
        run_b()
    _finish()
    _assert_same_logs(a.logs(), b.logs(), kind)
    _assert_same_bits(a.state_dict(), b.state_dict(), f"two runs of {kind}")


@pytest.mark.parametrize("stage", [2, 3])
def test_cognitive_step_one_launch_update_equals_separate_launches(deterministic, stage):
    """Stage II / III: ``CognitiveStep.step`` (deferred weight gradients, fmri_apply_batch with the scripts' gradient
    clamp) against forward / gate / backward / apply with the separate launches: the same bits after three steps."""
    from oracle import vaegan_oracle as O
    from fmri_hip.params import ArchConfig
    from fmri_hip.steps import CognitiveStep
    cfg, cfg_o, B, V = ArchConfig.px64(), O.ArchCfg.px64(), 8, 512
    data = O.synth_batch(B, cfg_o, n_voxels=V, seed=4321, steps=1)
    x, fm = data["x"].to(DEV), data["fmri"].to(DEV)
    nz = [t.to(DEV) for t in data["noise"][0]]
    from fmri_hip import ops
    a, b = CognitiveStep(cfg, V, DEV, stage), CognitiveStep(cfg, V, DEV, stage)
    a.load_recipe(3, True)
    b.load_state_dict(a.state_dict())
    fused, real = 0, ops.apply_group

    def counting(*aa, **kw):
        nonlocal fused
        r = real(*aa, **kw)
        fused += 1 if r else 0
        return r
    ops.apply_group = counting
    try:
        for _ in range(3):
            a.step(fm, x, nz[0], nz[1], nz[2])
            b.forward(fm, x, nz[0], nz[1], nz[2])
            b.gate(B)
            b.backward()
            b.apply()
        _finish()
    finally:
        ops.apply_group = real
    assert fused == 6, f"the one-launch path ran {fused} times in 3 steps x 2 trained sub-networks"
    _assert_same_logs(a.logs(), b.logs(), f"stage {stage}")
    _assert_same_bits(a.state_dict(), b.state_dict(), f"stage {stage}: one launch vs separate")


def test_gated_weight_gradients_change_nothing_but_the_work(deterministic):
    """``gate_skip`` (fmri_wgrad_if: no weight-gradient GEMMs for a sub-network the equilibrium gate does not train in a
    step, as the reference skips that ``loss.backward()``) against an engine that always runs them and only conditions the
    update: the same bits after six steps in which the gate switches the discriminator off and on."""
    from oracle import vaegan_oracle as O
    from fmri_hip.params import ArchConfig
    from fmri_hip.steps import GanHyper, Stage1Step
    B = 8
    data = O.synth_batch(B, O.ArchCfg.px64(), seed=1234, steps=1)
    x, e, zp = data["x"].to(DEV), data["noise"][0, 0].to(DEV), data["noise"][0, 1].to(DEV)
    hp = dict(equilibrium=0.72, margin=0.02)          # bce of a fresh discriminator ~ 0.69: below equilibrium - margin
    a = Stage1Step(ArchConfig.px64(), DEV, hp=GanHyper(**hp), gate_skip=True)
    a.load_recipe(0, True)
    b = Stage1Step(ArchConfig.px64(), DEV, hp=GanHyper(**hp), gate_skip=False)
    b.load_state_dict(a.state_dict())
    seen = set()
    for it in range(6):
        if it == 3:                                   # re-arm: both sub-networks trained again
            a.set_hyper(equilibrium=0.68, margin=0.35)
            b.set_hyper(equilibrium=0.68, margin=0.35)
        a.step(x, e, zp)
        b.step(x, e, zp)
        _finish()
        la, lb = a.logs(), b.logs()
        seen.add((la["train_dis"], la["train_dec"]))
        _assert_same_logs(la, lb, f"gated vs ungated, step {it}")
    _assert_same_bits(a.state_dict(), b.state_dict(), "gated vs ungated")
    for n in ("opt_enc", "opt_dec", "opt_dis"):
        assert torch.equal(getattr(a, n).s1, getattr(b, n).s1), f"{n}: RMSprop state differs"
    assert len(seen) >= 2 and any(not (d and c) for d, c in seen), f"the gate never switched a sub-network off: {seen}"


def test_gradients_of_a_fused_step_are_not_silently_stale():
    """``named_grads()`` after ``step()`` raises (the weight gradients went from the GEMM layout into the update and were
    never stored in the reference layout); after ``backward()`` it returns them."""
    a, _, args, _ = _stage1_pair(8)
    a.step(*args)
    _finish()
    with pytest.raises(RuntimeError, match="consumed by its fused update"):
        a.named_grads()
    a.forward(*args)
    a.gate(8)
    a.backward()
    _finish()
    g = a.named_grads()
    assert all(torch.isfinite(v).all() for v in g.values()) and float(g["decoder.conv.0.conv.weight"].abs().max()) > 0
    a.apply()


def test_one_launch_update_in_the_default_reduction_mode():
    """The same comparison with the DEFAULT reductions at BASELINE configs[1]'s batch: the weight-gradient kernels ADD into
    persistent buffers (fp32 atomics), which fmri_apply_batch has to hand back zeroed every step.  Three steps of
    After every ``step()`` each such buffer is zero again in ALL its columns (the narrow kernel keeps a bias gradient in a
    spare column behind the taps); the first step agrees with forward / gate / backward / apply to the run-to-run spread
    of the atomics."""
    B = 256
    a, b, args, _ = _stage1_pair(B)
    for it in range(3):
        a.step(*args)
        b.forward(*args)
        b.gate(B)
        b.backward()
        b.apply()
        _finish()
        if it == 0:
            _same_update(a.state_dict(), b.state_dict(), "default mode, one launch vs separate, step 0")
            b.load_state_dict(a.state_dict())          # re-synchronise: later steps compare one update each
            _sync_optimizers(a, b)
    # the direct statement: every buffer a weight-gradient kernel adds into is back to zero after the step
    held = 0
    for net in (a.enc, a.dec, a.dis):
        for layer in vars(net).values():
            for lay in (layer if isinstance(layer, (list, tuple)) else [layer]):
                for obj in [lay] + [v for v in vars(lay).values()] if hasattr(lay, "__dict__") else []:
                    hold = getattr(obj, "_ghold", None)
                    if not isinstance(hold, dict):
                        continue
                    for key, buf in hold.items():
                        if key != "busy" and getattr(buf, "_fmri_clear", False):
                            held += 1
                            assert float(buf.abs().max()) == 0.0, f"{type(obj).__name__}: accumulation buffer not cleared"
    assert held >= 3, f"only {held} accumulation buffers found: the walk over the layers missed them"



@pytest.mark.parametrize("B", [4, 5, 7])
def test_deterministic_px100_unfused_backward_sums_only_written_slabs(deterministic, B):
    """Deterministic weight gradients at the as-shipped 100-px geometry through the NON-fused forward / gate / backward
    path (fresh slab buffers per launch, no persistent zero-initialised holds): the library stores ceil(steps /
    ceil(steps / splits)) slabs, which is fewer than the ``splits`` the caller allocates at these sizes (18 of 19 at
    encoder.conv.0 with 4 images) -- the trailing slabs must be zero, not allocator leftovers.  The allocator is primed
    with NaN-filled blocks of the slab sizes first, so a summed stale slab cannot pass by luck; two runs must give the
    same bits, finite, and agree with the default (atomic) reduction mode to its rounding."""
    from oracle import vaegan_oracle as O
    from fmri_hip import ops
    from fmri_hip.params import ArchConfig
    from fmri_hip.steps import Stage1Step
    cfg, cfg_o = ArchConfig.px100(), O.ArchCfg.px100()
    data = O.synth_batch(B, cfg_o, seed=99, steps=1)
    x, e, zp = data["x"].to(DEV), data["noise"][0, 0].to(DEV), data["noise"][0, 1].to(DEV)

    def grads(det):
        was = ops.set_deterministic(det)
        try:
            st = Stage1Step(cfg, DEV)
            st.load_recipe(3, True)
            # poison the caching allocator's free lists: blocks of many sizes filled with NaN, then released
            junk = [torch.full((n,), float("nan"), device=DEV) for n in (1 << 12, 1 << 16, 1 << 18, 1 << 20, 1 << 22, 1 << 24)
                    for _ in range(3)]
            del junk
            st.forward(x, e, zp)
            st.gate(B)
            st.backward()
            ops.join_side()
            torch.cuda.synchronize()
            return {k: v.clone() for k, v in st.named_grads().items()}
        finally:
            ops.set_deterministic(was)

    ga, gb, gd = grads(True), grads(True), grads(False)
    for k in ga:
        assert torch.isfinite(ga[k]).all(), f"{k}: non-finite deterministic gradient (an unwritten slab was summed)"
    _assert_same_bits(ga, gb, f"two deterministic px100 backward passes at B = {B}")
    for k in ga:
        ref = gd[k].float()
        err = (ga[k].float() - ref).abs().max().item()
        assert err <= 2e-3 * ref.abs().max().item() + 1e-7, (k, err, ref.abs().max().item())
