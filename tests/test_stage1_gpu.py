"""GPU parity of the fused Stage-I VAE/GAN step (HIP engine, through the C ABI) against
  (a) the CPU oracle on the same seeded inputs, and
  (b) the committed golden vectors that were produced by the real reference.

Tolerances: logged losses 1e-3 relative (the bar BASELINE.json's north_star states); forward tensors
3e-3 of their RMS (fp16 storage); gradients 2e-2 of the tensor's norm (fp16 operands, fp32 accumulate).
"""
import os

import numpy as np
import pytest
import torch

import gradcheck

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
LOSS_RTOL = 1e-3
FW_TOL = 1e-2


def _rel(a, b):
    return abs(a - b) / max(abs(b), 1e-12)


def _tensor_err(got, ref):
    got, ref = got.detach().float().cpu().reshape(-1), ref.detach().float().cpu().reshape(-1)
    return ((got - ref).norm() / (ref.norm() + 1e-20)).item()


def _same_update(sa, sb, what=""):
    """Two runs of the same step leave the same parameters, up to the run-to-run spread of the fp32 atomics in the
    weight-gradient sums.  RMSprop's first update is lr*g/(sqrt(0.1 g^2)+1e-8): +-3.16e-4 for every element whose
    gradient is well above 1e-8, PROPORTIONAL to g below that -- so the spread shows as (a) up to ~1e-1 of a step on
    the elements with near-zero gradients (measured: 5 % of encoder.conv.1 moved by <= 3.6e-5 in one run of four) and
    (b) single elements whose gradient changes sign (a full 6.3e-4; the 3-element bias of the decoder's last conv does
    this regularly).  A wrong or missing update moves (nearly) EVERY element of a tensor by a step.  Hence: elements may
    differ by a quarter step (8e-5; or 2e-5 of the tensor's largest entry, for the running statistics), and at most
    max(4, 1 %) of a tensor's elements by more."""
    for k in sa:
        a, b = sa[k].float().cpu().reshape(-1), sb[k].float().cpu().reshape(-1)
        lim = max(2e-5 * float(b.abs().max()), 8e-5)
        bad = int(((a - b).abs() > lim).sum())
        assert bad <= max(4, a.numel() // 100), (what, k, bad, a.numel(), float((a - b).abs().max()))


def _engine_relu_masks(st):
    """The ReLU masks of the engine's last forward (stored activations > 0) in the call order of the oracle's ReLUs:
    encoder (3 conv blocks, fc), decoder(z), decoder(z_p) (fc, 3 deconv blocks each), discriminator 'REC' pass
    (conv0, blocks 1-2) and 'GAN' pass (conv0, blocks 1-3, fc)."""
    fw = st.fw
    B = fw["B"]
    nchw = lambda t: (t.float() > 0).permute(0, 3, 1, 2).float().cpu()
    rows = lambda t: (t.float() > 0).float().cpu()
    e, d, c = fw["ectx"], fw["dctx"], fw["sctx"]
    m = [nchw(e["acts"][i]) for i in (1, 2, 3)] + [rows(e["hfc"])]
    for g in range(2):
        sl = slice(g * B, (g + 1) * B)
        m.append(nchw(d["acts"][0][sl]).reshape(B, -1))          # (H,W,C) engine order -> the reference's (C,H,W)
        m += [nchw(d["acts"][i][sl]) for i in (1, 2, 3)]
    rec = [nchw(c["acts"][i]) for i in (0, 1, 2)]
    return m + rec + [t.clone() for t in rec] + [nchw(c["acts"][3]), rows(c["hfc"])]


def _run_engine(cfg_e, B, seed, perturb, steps, noise, x, masks_out=None):
    from fmri_hip.steps import Stage1Step
    st = Stage1Step(cfg_e, DEV)
    st.load_recipe(seed, perturb)
    outs = []
    for s in range(steps):
        st.forward(x.to(DEV), noise[s, 0].to(DEV), noise[s, 1].to(DEV))
        if s == 0 and masks_out is not None:
            masks_out.extend(_engine_relu_masks(st))
        st.gate(B)
        st.backward()
        rec = dict(logs=None, outputs={k: v.cpu() for k, v in st.outputs().items()},
                   grads={k: v.detach().cpu().clone() for k, v in st.named_grads().items()})
        st.apply()
        rec["logs"] = st.logs()
        rec["state"] = {k: v.cpu() for k, v in st.state_dict().items()}
        outs.append(rec)
    return outs


@pytest.mark.parametrize("arch,B,seed,perturb", [("px64", 4, 0, True), ("px100", 4, 3, True)])
def test_stage1_step_matches_oracle(arch, B, seed, perturb):
    from oracle import vaegan_oracle as O
    from fmri_hip.params import ArchConfig
    cfg_o = getattr(O.ArchCfg, arch)()
    cfg_e = getattr(ArchConfig, arch)()
    steps = 2
    data = O.synth_batch(B, cfg_o, seed=1234, steps=steps)
    masks = []
    eng = _run_engine(cfg_e, B, seed, perturb, steps, data["noise"], data["x"], masks_out=masks)
    P = O.fill_state(O.vaegan_spec(cfg_o), seed, perturb)
    opts = {n: O.OptState(kind="rmsprop", lr=1e-4) for n in ("encoder", "decoder", "discriminator")}
    report = []
    fw_fail = False
    for s in range(steps):
        ref = O.stage1_step(P, opts, data["x"], data["noise"][s, 0], data["noise"][s, 1], cfg_o, keep_grads=True)
        e = eng[s]
        for k in ("loss_encoder", "loss_decoder", "loss_discriminator", "nle", "kl", "mse", "bce_orig", "bce_pred",
                  "bce_samp"):
            r = _rel(e["logs"][k], ref["logs"][k])
            report.append((s, k, e["logs"][k], ref["logs"][k], r))
        assert e["logs"]["train_dis"] == ref["logs"]["train_dis"] and e["logs"]["train_dec"] == ref["logs"]["train_dec"]
        for k in ("x_tilde", "x_p", "disc_class", "disc_layer", "mus", "log_variances"):
            err = _tensor_err(e["outputs"][k], ref["fw"][k])
            report.append((s, "fw:" + k, err, 0, err))
            fw_fail = fw_fail or (s == 0 and err > FW_TOL)
        if s == 0:
            # gradients: tight against the oracle run with the ENGINE's ReLU masks and its 16-bit storage model (what is
            # left is rounding), direction / length against the plain fp32 oracle (tests/gradcheck.py)
            P16 = O.fill_state(O.vaegan_spec(cfg_o), seed, perturb)
            o16 = {n: O.OptState(kind="rmsprop", lr=1e-4) for n in ("encoder", "decoder", "discriminator")}
            with gradcheck.storage16(O):
                O.RELU_MASKS = list(masks)
                try:
                    ref16 = O.stage1_step(P16, o16, data["x"], data["noise"][s, 0], data["noise"][s, 1], cfg_o,
                                          keep_grads=True)
                finally:
                    left, O.RELU_MASKS = len(O.RELU_MASKS), None
            assert left == 0, left
            gradcheck.check(e["grads"], ref["grads"], ref16["grads"], f"stage1 {arch}")
    for row in report:
        print(row)
    assert not fw_fail, "forward tensors off"
    for s, k, got, want, r in report:
        if isinstance(k, str) and not k.startswith(("fw:", "grad")) and s == 0:
            assert r < LOSS_RTOL, (s, k, got, want, r)
    # Losses of the NEXT forward (on the updated weights).  RMSprop's first update is lr*sign(g)*3.16 for
    # every weight, so the ~10% gradient noise flips the sign of ~3% of the updates; at this tiny batch that
    # moves mu/logvar by 15-20% and the KL term by a few % while the large sums stay within ~1e-3
    # (measured px64 B=4: enc 1.0e-3, dec 7e-4, dis 7e-4, nle 9e-4, mse 9e-4, kl 2.7e-2).  DESIGN.md 5.
    for s, k, got, want, r in report:
        if isinstance(k, str) and not k.startswith(("fw:", "grad")) and s == 1:
            assert r < 5e-2, (s, k, got, want, r)


def test_stage1_matches_reference_golden(golden_dir):
    """First-step losses against the numbers the real reference produced (tests/golden/stage1_b4.npz)."""
    from oracle import vaegan_oracle as O
    from fmri_hip.params import ArchConfig
    g = np.load(os.path.join(golden_dir, "stage1_b4.npz"))
    B, seed, perturb = int(g["meta/B"]), int(g["meta/seed"]), bool(g["meta/perturb"])
    data = O.synth_batch(B, O.ArchCfg.px64(), seed=1234, steps=2)
    eng = _run_engine(ArchConfig.px64(), B, seed, perturb, 1, data["noise"], data["x"])
    for k in ("loss_encoder", "loss_decoder", "loss_discriminator", "nle", "kl", "mse"):
        want = float(g[f"step0/logs/{k}"])
        got = eng[0]["logs"][k]
        assert _rel(got, want) < LOSS_RTOL, (k, got, want)
    # post-step parameter fingerprints (norm of every updated tensor)
    keys = [str(k) for k in g["step0/state_keys"]]
    summ = g["step0/state_sum"]
    for i, k in enumerate(keys):
        if "num_batches" in k:
            assert float(eng[0]["state"][k]) == summ[i][1]
            continue
        got = eng[0]["state"][k].double().norm().item()
        assert _rel(got, summ[i][0]) < 2e-3, (k, got, summ[i][0])


def test_stage1_full_batch_first_step_matches_oracle():
    """BASELINE configs[1] (B = 256, the batch bench.py times): every logged loss of the first step against the CPU
    oracle run live on the same seeded inputs (~10-20 s of CPU), gates equal, and the BatchNorm running statistics
    after the step."""
    from oracle import vaegan_oracle as O
    from fmri_hip.params import ArchConfig
    B, seed = 256, 0
    cfg_o, cfg_e = O.ArchCfg.px64(), ArchConfig.px64()
    data = O.synth_batch(B, cfg_o, seed=1234, steps=1)
    eng = _run_engine(cfg_e, B, seed, True, 1, data["noise"], data["x"])[0]
    P = O.fill_state(O.vaegan_spec(cfg_o), seed, True)
    opts = {n: O.OptState(kind="rmsprop", lr=1e-4) for n in ("encoder", "decoder", "discriminator")}
    ref = O.stage1_step(P, opts, data["x"], data["noise"][0, 0], data["noise"][0, 1], cfg_o)
    for k in ("loss_encoder", "loss_decoder", "loss_discriminator", "nle", "kl", "mse", "bce_orig", "bce_pred",
              "bce_samp"):
        r = _rel(eng["logs"][k], ref["logs"][k])
        print(k, eng["logs"][k], ref["logs"][k], r)
        assert r < LOSS_RTOL, (k, eng["logs"][k], ref["logs"][k], r)
    assert eng["logs"]["train_dis"] == ref["logs"]["train_dis"] and eng["logs"]["train_dec"] == ref["logs"]["train_dec"]
    for k in ("x_tilde", "disc_class", "mus", "log_variances"):
        err = _tensor_err(eng["outputs"][k], ref["fw"][k])
        assert err < FW_TOL, (k, err)
    for k, v in P.items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            err = _tensor_err(eng["state"][k], v)
            assert err < 5e-3, (k, err)


def test_fused_step_equals_separate_calls():
    """``Stage1Step.step`` (weight gradients, discriminator / decoder optimizer updates and weight repacks queued on the
    side stream under the rest of the backward pass) against the same step issued as forward / gate / backward / apply
    on one stream: same losses and the same parameters after the step.  (One step only: the fp32 atomics of the
    weight-gradient kernels make two runs of the SAME code differ by ~1e-6 after one step, and RMSprop's sign-like
    first updates grow that to 1e-2 within three steps -- measured with tools/debug_fused.py.)"""
    from oracle import vaegan_oracle as O
    from fmri_hip import ops
    from fmri_hip.params import ArchConfig
    from fmri_hip.steps import Stage1Step
    B = 8
    data = O.synth_batch(B, O.ArchCfg.px64(), seed=1234, steps=1)
    x = data["x"].to(DEV)
    e, zp = data["noise"][0, 0].to(DEV), data["noise"][0, 1].to(DEV)
    res = []
    side_was = ops._SIDE["on"]
    try:
        for fused in (True, False):
            st = Stage1Step(ArchConfig.px64(), DEV)
            st.load_recipe(0, True)
            ops._SIDE["on"] = fused
            if fused:
                st.step(x, e, zp)
            else:
                st.forward(x, e, zp)
                st.gate(B)
                st.backward()
                st.apply()
            ops.join_side()
            torch.cuda.synchronize()
            res.append((st.logs(), {k: v.float().cpu() for k, v in st.state_dict().items()}))
    finally:
        ops._SIDE["on"] = side_was
    (la, sa), (lb, sb) = res
    for k in ("loss_encoder", "loss_decoder", "loss_discriminator", "kl"):
        assert _rel(la[k], lb[k]) < 1e-5, (k, la[k], lb[k])
    _same_update(sa, sb, "fused vs separate")


def test_hybrid_recorded_forward_step_equals_eager_step():
    """``Stage1Step.capture_forward``: forward + gate replayed from a HIP graph, backward / updates issued eagerly on two
    streams -- against the plain ``step`` of a second engine started from the same parameters and RMSprop state: same
    losses, same parameters after the step (to the run-to-run spread of the fp32 atomics), twice in a row (the second
    replay must see the weights the first one's early updates produced)."""
    from oracle import vaegan_oracle as O
    from fmri_hip import ops
    from fmri_hip.params import ArchConfig
    from fmri_hip.steps import Stage1Step
    B = 8
    data = O.synth_batch(B, O.ArchCfg.px64(), seed=1234, steps=1)
    x = data["x"].to(DEV)
    e, zp = data["noise"][0, 0].to(DEV), data["noise"][0, 1].to(DEV)
    a = Stage1Step(ArchConfig.px64(), DEV)
    a.load_recipe(0, True)
    run = a.capture_forward(x, e, zp, warmup=1)
    b = Stage1Step(ArchConfig.px64(), DEV)
    b.load_state_dict(a.state_dict())
    for oa, ob in ((a.opt_enc, b.opt_enc), (a.opt_dec, b.opt_dec), (a.opt_dis, b.opt_dis)):
        ob.s1.copy_(oa.s1)
    for it in range(2):
        run()
        b.step(x, e, zp)
        ops.join_side()
        torch.cuda.synchronize()
        la, lb = a.logs(), b.logs()
        for k in ("loss_encoder", "loss_decoder", "loss_discriminator", "kl"):
            # second step: the first step's +-3.16e-4 sign-like updates already differ on a few near-zero-gradient
            # elements between two runs; the small KL term feels that most
            assert _rel(la[k], lb[k]) < (1e-5 if it == 0 else (2e-2 if k == "kl" else 5e-3)), (it, k, la[k], lb[k])
        sa, sb = a.state_dict(), b.state_dict()
        if it == 0:
            _same_update(sa, sb, "hybrid vs eager")
        else:
            for k in sa:
                assert _tensor_err(sa[k], sb[k]) < 2e-2, (it, k, _tensor_err(sa[k], sb[k]))


@pytest.mark.parametrize("mode,beta", [("beta-vae", 4.0), ("dcgan", 1.0), ("vae", 1.0)])
def test_stage1_modes_match_oracle_and_golden(golden_dir, mode, beta):
    """The other loss compositions of train_vgan_stage1.py:359-388 on the fused step: first-step losses against the
    oracle and the reference's golden numbers (1e-3), same gate flags, gradients of the trained sub-networks, next-step
    losses within the sign-like-update bound."""
    from oracle import vaegan_oracle as O
    from fmri_hip.params import ArchConfig
    from fmri_hip.steps import Stage1Step, GanHyper
    name = {"beta-vae": "stage1_betavae_b4", "dcgan": "stage1_dcgan_b4", "vae": "stage1_vae_b4"}[mode]
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    B, seed, perturb, steps = int(g["meta/B"]), int(g["meta/seed"]), bool(g["meta/perturb"]), int(g["meta/steps"])
    assert str(g["meta/mode"]) == mode and float(g["meta/beta"]) == beta
    cfg_o = O.ArchCfg.px64()
    data = O.synth_batch(B, cfg_o, seed=1234, steps=steps)
    st = Stage1Step(ArchConfig.px64(), DEV, hp=GanHyper(beta=beta), mode=mode)
    st.load_recipe(seed, perturb)
    P = O.fill_state(O.vaegan_spec(cfg_o), seed, perturb)
    opts = {n: O.OptState(kind="rmsprop", lr=1e-4) for n in ("encoder", "decoder", "discriminator")}
    x = data["x"].to(DEV)
    keys = ("loss_encoder", "loss_decoder", "loss_discriminator", "nle", "kl", "mse", "bce_orig", "bce_pred", "bce_samp")
    for s in range(steps):
        st.forward(x, data["noise"][s, 0].to(DEV), data["noise"][s, 1].to(DEV))
        st.gate(B)
        st.backward()
        grads = {k: v.detach().cpu().clone() for k, v in st.named_grads().items()}
        st.apply()
        logs = st.logs()
        ref = O.stage1_step(P, opts, data["x"], data["noise"][s, 0], data["noise"][s, 1], cfg_o, keep_grads=True,
                            mode=mode, beta=beta)
        assert logs["train_dis"] == ref["logs"]["train_dis"] and logs["train_dec"] == ref["logs"]["train_dec"], s
        assert logs["train_dis"] == bool(g[f"step{s}/logs/train_dis"]) and logs["train_dec"] == bool(
            g[f"step{s}/logs/train_dec"])
        for k in keys:
            r = _rel(logs[k], ref["logs"][k])
            print(mode, s, k, logs[k], ref["logs"][k], r)
            if s == 0:
                assert r < LOSS_RTOL, (k, logs[k], ref["logs"][k])
                assert _rel(logs[k], float(g[f"step0/logs/{k}"])) < LOSS_RTOL, (k, "golden")
            else:
                assert r < 5e-2, (s, k, logs[k], ref["logs"][k])
        if s == 0:
            skip = [k for k in ref["grads"] if mode == "dcgan" and k.startswith("encoder.")]
            gradcheck.check(grads, ref["grads"], ref["grads"], f"stage1 {mode}", skip=skip, tol16=None)
    # post-step parameter fingerprints of the reference (sub-networks that were not trained stay put)
    sd = {k: v.cpu() for k, v in st.state_dict().items()}
    skeys = [str(k) for k in g[f"step{steps - 1}/state_keys"]]
    summ = g[f"step{steps - 1}/state_sum"]
    for i, k in enumerate(skeys):
        if "num_batches" in k:
            assert float(sd[k]) == summ[i][1], k
        elif "running" not in k:
            assert _rel(sd[k].double().norm().item(), summ[i][0]) < 2e-3, (k, sd[k].double().norm().item(), summ[i][0])


def test_recorded_step_follows_hyper_parameter_schedule():
    """lr, lambda, equilibrium and margin live in device memory: two replays of ONE captured step with the epoch-end
    updates of train_vgan_stage1.py:448-458 applied in between equal two eagerly issued steps with the same schedule;
    the gate of the second replay is the oracle's gate under the new equilibrium / margin and the size of its update
    is the oracle's under the new learning rate."""
    from oracle import vaegan_oracle as O
    from fmri_hip import ops
    from fmri_hip.params import ArchConfig
    from fmri_hip.steps import Stage1Step
    B = 8
    cfg_o = O.ArchCfg.px64()
    data = O.synth_batch(B, cfg_o, seed=1234, steps=1)
    x, e, zp = data["x"].to(DEV), data["noise"][0, 0].to(DEV), data["noise"][0, 1].to(DEV)
    # "epoch end": lr halved, lambda x 100, equilibrium far above every bce mean -> train_dis = False, train_dec = True
    sched = dict(lr=0.5e-4, margin=0.01, equilibrium=10.0, lambda_mse=1e-4)
    a = Stage1Step(ArchConfig.px64(), DEV)
    a.load_recipe(0, True)
    run = a.capture(x, e, zp, warmup=1)               # one real (warm-up) step, then the recording (executes nothing)
    b = Stage1Step(ArchConfig.px64(), DEV)
    b.load_state_dict(a.state_dict())
    for oa, ob in ((a.opt_enc, b.opt_enc), (a.opt_dec, b.opt_dec), (a.opt_dis, b.opt_dis)):
        ob.s1.copy_(oa.s1)
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
            b.step(x, e, zp)
        finally:
            ops._SIDE["on"] = side_was
        torch.cuda.synchronize()
        la, lb = a.logs(), b.logs()
        flags.append((la["train_dis"], la["train_dec"]))
        assert (la["train_dis"], la["train_dec"]) == (lb["train_dis"], lb["train_dec"]), it
        for k in ("loss_encoder", "loss_decoder", "loss_discriminator", "kl"):
            # loss_decoder = lambda * mse - (1 - lambda) * loss_discriminator cancels to ~1e-3 under this schedule: its
            # error is measured on the scale of its terms, not of the remainder
            floor = 1e-2 * abs(lb["loss_discriminator"]) if k == "loss_decoder" else 0.0
            err = abs(la[k] - lb[k]) / max(abs(lb[k]), floor, 1e-12)
            assert err < (1e-5 if it == 0 else 5e-3), (it, k, la[k], lb[k])
        if it == 0:
            _same_update(a.state_dict(), b.state_dict(), "replay vs eager")
    assert flags[1] == (False, True), flags
    after = a.state_dict()
    # oracle: warm-up step, default step, scheduled step
    P = O.fill_state(O.vaegan_spec(cfg_o), 0, True)
    opts = {n: O.OptState(kind="rmsprop", lr=1e-4) for n in ("encoder", "decoder", "discriminator")}
    hp = O.GanHyper()
    args = (data["x"], data["noise"][0, 0], data["noise"][0, 1], cfg_o)
    O.stage1_step(P, opts, *args, hp=hp)
    ref = O.stage1_step(P, opts, *args, hp=hp)
    assert flags[0] == (ref["logs"]["train_dis"], ref["logs"]["train_dec"]), flags
    hp = O.GanHyper(lr=sched["lr"], lambda_mse=sched["lambda_mse"], margin=sched["margin"],
                    equilibrium=sched["equilibrium"])
    for o in opts.values():
        o.lr = sched["lr"]
    P2 = {k: v.clone() for k, v in P.items()}
    ref = O.stage1_step(P, opts, *args, hp=hp)
    assert flags[1] == (ref["logs"]["train_dis"], ref["logs"]["train_dec"]), flags
    for k in ("encoder.fc.0.weight", "decoder.conv.0.conv.weight", "discriminator.conv.2.conv.weight"):
        du_e = (after[k].float().cpu() - before[k].float().cpu()).norm().item()
        du_o = (P[k] - P2[k]).norm().item()
        print(k, du_e, du_o)
        if k.startswith("discriminator."):
            assert du_e == 0.0 and du_o == 0.0, k          # gated off by the new equilibrium
        else:
            assert abs(du_e - du_o) < 0.1 * du_o, (k, du_e, du_o)    # half the step of lr = 1e-4


def test_recorded_forward_survives_an_eager_step_in_between():
    """capture_forward(): an eager step() at another batch size between two run() calls (the last, partial batch of an
    epoch) must not leave run() back-propagating the eager batch (it rebinds the recorded forward's tensors)."""
    from oracle import vaegan_oracle as O
    from fmri_hip import ops
    from fmri_hip.params import ArchConfig
    from fmri_hip.steps import Stage1Step
    B = 8
    cfg_o = O.ArchCfg.px64()
    data = O.synth_batch(B, cfg_o, seed=1234, steps=1)
    x, e, zp = data["x"].to(DEV), data["noise"][0, 0].to(DEV), data["noise"][0, 1].to(DEV)
    a = Stage1Step(ArchConfig.px64(), DEV)
    a.load_recipe(0, True)
    run = a.capture_forward(x, e, zp, warmup=1)
    b = Stage1Step(ArchConfig.px64(), DEV)
    b.load_state_dict(a.state_dict())
    for oa, ob in ((a.opt_enc, b.opt_enc), (a.opt_dec, b.opt_dec), (a.opt_dis, b.opt_dis)):
        ob.s1.copy_(oa.s1)
    seq = [("run", None), ("eager", 4), ("run", None)]
    for what, n in seq:
        if what == "run":
            run()
            b.step(x, e, zp)
        else:
            a.step(x[:n], e[:n], zp[:n])
            b.step(x[:n], e[:n], zp[:n])
        ops.join_side()
        torch.cuda.synchronize()
    la, lb = a.logs(), b.logs()
    for k in ("loss_encoder", "loss_decoder", "loss_discriminator"):
        assert _rel(la[k], lb[k]) < 2e-2, (k, la[k], lb[k])
    sa, sb = a.state_dict(), b.state_dict()
    for k in sa:
        if sa[k].dtype == torch.float32 and "running" not in k and sa[k].numel() > 1000:
            assert _tensor_err(sa[k], sb[k]) < 2e-2, (k, _tensor_err(sa[k], sb[k]))


def test_stage1_b32_matches_reference_golden(golden_dir):
    """BASELINE configs[0] (Stage-I, batch 32 -- the reference's own CPU-runnable case): the first-step losses of the HIP
    step against the numbers the real reference produced (tests/golden/stage1_b32.npz)."""
    from oracle import vaegan_oracle as O
    from fmri_hip.params import ArchConfig
    g = np.load(os.path.join(golden_dir, "stage1_b32.npz"))
    B, seed, perturb = int(g["meta/B"]), int(g["meta/seed"]), bool(g["meta/perturb"])
    assert B == 32
    data = O.synth_batch(B, O.ArchCfg.px64(), seed=1234, steps=1)
    eng = _run_engine(ArchConfig.px64(), B, seed, perturb, 1, data["noise"], data["x"])[0]
    for k in ("loss_encoder", "loss_decoder", "loss_discriminator", "nle", "kl", "mse", "bce_orig", "bce_pred",
              "bce_samp"):
        want = float(g[f"step0/logs/{k}"])
        print(k, eng["logs"][k], want, _rel(eng["logs"][k], want))
        assert _rel(eng["logs"][k], want) < LOSS_RTOL, (k, eng["logs"][k], want)
    assert eng["logs"]["train_dis"] == bool(g["step0/logs/train_dis"])
    assert eng["logs"]["train_dec"] == bool(g["step0/logs/train_dec"])
    keys = [str(k) for k in g["step0/state_keys"]]
    summ = g["step0/state_sum"]
    for i, k in enumerate(keys):
        if "num_batches" in k:
            assert float(eng["state"][k]) == summ[i][1]
        elif "running" not in k:
            assert _rel(eng["state"][k].double().norm().item(), summ[i][0]) < 2e-3, k


def test_stage1_full_batch_two_steps_match_oracle():
    """BASELINE configs[1] (B = 256): losses of the first AND of the second step (i.e. "after one step", on the weights
    the engine itself updated) against the CPU oracle run live, with each ratio reported.  The second step sees the
    sign-like RMSprop updates of step one (+-3.16e-4 per weight, ~1 % of them with the other sign because of ReLU-mask
    flips): at B = 256 that leaves the large sums within ~1e-3 and moves the small KL term by a few 1e-3."""
    from oracle import vaegan_oracle as O
    from fmri_hip.params import ArchConfig
    B, seed = 256, 0
    cfg_o, cfg_e = O.ArchCfg.px64(), ArchConfig.px64()
    data = O.synth_batch(B, cfg_o, seed=1234, steps=2)
    eng = _run_engine(cfg_e, B, seed, True, 2, data["noise"], data["x"])
    P = O.fill_state(O.vaegan_spec(cfg_o), seed, True)
    opts = {n: O.OptState(kind="rmsprop", lr=1e-4) for n in ("encoder", "decoder", "discriminator")}
    worst = {}
    log = os.environ.get("FMRI_GRADLOG")
    for s in range(2):
        ref = O.stage1_step(P, opts, data["x"], data["noise"][s, 0], data["noise"][s, 1], cfg_o)
        assert eng[s]["logs"]["train_dis"] == ref["logs"]["train_dis"], s
        assert eng[s]["logs"]["train_dec"] == ref["logs"]["train_dec"], s
        for k in ("loss_encoder", "loss_decoder", "loss_discriminator", "nle", "kl", "mse", "bce_orig", "bce_pred",
                  "bce_samp"):
            r = _rel(eng[s]["logs"][k], ref["logs"][k])
            line = f"B256 step {s} {k}: engine {eng[s]['logs'][k]:.6g} oracle {ref['logs'][k]:.6g} rel {r:.2e}"
            print(line)
            if log:
                with open(log, "a") as f:
                    f.write(line + "\n")
            worst[(s, k)] = r
            # step 0: the north-star bar (1e-3; measured <= 5e-5).  step 1 = "after one step" on the engine's own updated
            # weights: RMSprop's first update is +-3.16 lr per weight whatever the gradient's size, so the sign of every
            # near-zero gradient decides where its weight goes, and that sign hangs on single fp16 roundings.  Measured in
            # round 3 with nothing changed but which (numerically equivalent: 0.2503-0.2516 ulp mean error against fp64 for
            # every one of them, tools/probes/conv_err.py) convolution kernels run: 3.1e-4, 7.2e-4, 1.66e-3, 2.4e-3
            # (FMRI_C5W / FMRI_TC5W off-off, off-on, on-off, on-on; profiles/r03_fullbatch_routing.log); round 2: 1.29e-3.
            # The bound is the spread, not a precision claim
            assert r < (LOSS_RTOL if s == 0 else 3e-3), (s, k, eng[s]["logs"][k], ref["logs"][k])


def test_decoder_fc_running_statistics_lazy_shadow_round_trips():
    """decoder.fc.1 (the (C,H,W)-permuted BatchNorm1d) keeps its running statistics in engine order inside the fused steps
    and writes them back only when the state dict is read: state_dict() before any step returns what was loaded, after
    steps (eager and replayed from a HIP graph) what an eagerly synchronised BatchNorm holds, and load_state_dict() in
    between reaches the next step."""
    from oracle import vaegan_oracle as O
    from fmri_hip.params import ArchConfig
    from fmri_hip.steps import Stage1Step
    B = 4
    data = O.synth_batch(B, O.ArchCfg.px64(), seed=1234, steps=1)
    x, e, zp = data["x"].to(DEV), data["noise"][0, 0].to(DEV), data["noise"][0, 1].to(DEV)
    a = Stage1Step(ArchConfig.px64(), DEV)
    a.load_recipe(0, True)
    keys = ("decoder.fc.1.running_mean", "decoder.fc.1.running_var")
    sd0 = a.state_dict()
    assert float(sd0[keys[0]].abs().max()) == 0.0 and float((sd0[keys[1]] - 1.0).abs().max()) == 0.0   # as loaded
    b = Stage1Step(ArchConfig.px64(), DEV)
    b.load_state_dict(sd0)
    b.dec.fc_bn._lazy = False                         # reference behaviour: synchronised around every call
    a.step(x, e, zp)
    b.step(x, e, zp)
    sa, sb = a.state_dict(), b.state_dict()
    for k in keys:
        assert _tensor_err(sa[k], sb[k]) < 1e-5, k
    # an outside write of the buffers reaches the next step
    sd = a.state_dict()
    sd[keys[0]] = torch.full_like(sd[keys[0]], 3.0)
    a.load_state_dict(sd)
    b.load_state_dict(sd)
    run = a.capture(x, e, zp, warmup=1)
    run()
    from fmri_hip import ops
    side_was, ops._SIDE["on"] = ops._SIDE["on"], False
    try:
        b.step(x, e, zp)
        b.step(x, e, zp)
    finally:
        ops._SIDE["on"] = side_was
    torch.cuda.synchronize()
    sa, sb = a.state_dict(), b.state_dict()
    for k in keys:
        assert _tensor_err(sa[k], sb[k]) < 2e-2, (k, _tensor_err(sa[k], sb[k]))
