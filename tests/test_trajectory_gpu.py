"""Trajectory-level checks of the fused Stage-I step (the reference's unit of work is an epoch of steps,
train/train_vgan_stage1.py:311-445, not one step).

  * 150 steps at the benchmark's batch 256 on bench.py's rotating synthetic batches stay finite in every launch mode
    (two streams, one stream, recorded forward), weights and optimizer state included -- before round 5 roughly a third of
    such runs ended with NaN losses: the latent's sigma = exp(0.5 logvar) left fp16's range.
  * 25 steps at batch 32 next to the fp32 CPU oracle from the same weights and data, one oracle run for two engines: a
    free-running one (same equilibrium-gate decisions, losses inside a stated envelope for as long as two arithmetic
    models of a GAN can be expected to agree; 50-step tables in profiles/r05_trajectory_b32.log) and one that is
    re-loaded with the oracle's weights, BatchNorm buffers and RMSprop state every 6th step (the one-step parity at
    trained weights, not only at the initial recipe).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
NBATCH = 8
LOSSES = ("bce_orig", "bce_pred", "bce_samp", "kl", "mse", "nle")


def _bench_data(B, Z):
    """bench.py's rotating batches (rank 0)."""
    xs = [torch.from_numpy(np.random.RandomState(1234 + 97 * i).uniform(-1, 1, (B, 3, 64, 64)).astype(np.float32))
          for i in range(NBATCH)]
    nz = [torch.from_numpy(np.random.RandomState(1236 + 97 * i).standard_normal((2, B, Z)).astype(np.float32))
          for i in range(NBATCH)]
    return xs, nz


def _finite_state(st):
    for name, net, opt in (("encoder", st.enc, st.opt_enc), ("decoder", st.dec, st.opt_dec),
                           ("discriminator", st.dis, st.opt_dis)):
        assert torch.isfinite(net.group.data).all(), f"{name}: non-finite weights"
        assert torch.isfinite(opt.s1).all(), f"{name}: non-finite RMSprop state"
        for pw in net.group.packed:
            assert torch.isfinite(pw.buf.float()).all(), f"{name}: non-finite fp16 weight copy"
    for k, v in st.state_dict().items():
        if v.is_floating_point():
            assert torch.isfinite(v).all(), k


@pytest.mark.parametrize("mode", ["two_streams", "one_stream", "hybrid"])
def test_stage1_b256_150_steps_stay_finite(mode):
    from fmri_hip import ops
    from fmri_hip.params import ArchConfig
    from fmri_hip.steps import Stage1Step
    cfg = ArchConfig.px64()
    B, steps = 256, 150
    xs, nz = _bench_data(B, cfg.latent_dim)
    xs = [x.to(DEV) for x in xs]
    nz = [n.to(DEV) for n in nz]
    side_was = ops._SIDE["on"]
    ops.join_side()
    ops._SIDE["on"] = mode != "one_stream"
    try:
        st = Stage1Step(cfg, DEV, gate_skip=False)           # bench.py's headline setting: every GEMM in every step
        st.load_recipe(0, False)
        run = lambda i: st.step(xs[i % NBATCH], nz[i % NBATCH][0], nz[i % NBATCH][1])
        n0 = 0
        if mode == "hybrid":
            for i in range(20):
                run(i)
            n0 = 20
            sb = [xs[0].clone(), nz[0][0].clone(), nz[0][1].clone()]
            replay = st.capture_forward(*sb)

            def run(i):
                j = i % NBATCH
                sb[0].copy_(xs[j]); sb[1].copy_(nz[j][0]); sb[2].copy_(nz[j][1])
                return replay()
        rec, lv = [], []
        for i in range(n0, n0 + steps):
            run(i)
            rec.append(st.scal[:10].clone())
            lv.append(st.fw["head32"][:, cfg.latent_dim:].max())
        ops.join_side()
        torch.cuda.synchronize()
        R = torch.stack(rec).cpu().numpy()
        bad = np.argwhere(~np.isfinite(R))
        assert bad.size == 0, f"first non-finite loss at step {bad[0][0]} (slot {bad[0][1]})"
        _finite_state(st)
        print(f"[{mode}] max logvar over the run {float(torch.stack(lv).max()):.2f}; last kl {R[-1, 3]:.1f} "
              f"mse {R[-1, 4]:.1f} nle {R[-1, 5]:.1f}")
        # the run is a training run, not a fixed point: the decoder is alive (x_tilde is not the all-zero image, whose
        # nle against U[-1, 1] images is 1/6 per pixel value)
        assert R[-1, 5] > 1.2 * B * 3 * 64 * 64 / 6.0
    finally:
        ops.join_side()
        ops._SIDE["on"] = side_was


def _load_opt_state(st, opts, O, cfg_o):
    """Oracle RMSprop state -> the engine's flat square-average buffers (reference layout, FlatGroup.offsets)."""
    for name, net, opt in (("encoder", st.enc, st.opt_enc), ("decoder", st.dec, st.opt_dec),
                           ("discriminator", st.dis, st.opt_dis)):
        g = net.group
        opt.s1.zero_()
        for k in g.pkeys:
            buf = opts[name].bufs.get(f"{name}.{k}")
            if buf is not None:
                o = g.offsets[k]
                opt.s1[o:o + buf.numel()].copy_(buf.reshape(-1).to(DEV))


def _sd_for_engine(P):
    return {k: (v.detach().reshape(()) if k.endswith("num_batches_tracked") else v.detach().clone()) for k, v in P.items()}


def test_stage1_b32_next_to_the_oracle_free_running_and_reloaded(deterministic):
    """(Deterministic reductions: a free-running comparison amplifies last-bit differences, and the test should give the same
    verdict on every run.)  ONE 25-step run of the fp32 CPU oracle at batch 32 on bench.py's data (~4.5 s per step on the GPU box's host),
    two engines beside it:

    free-running   same initial weights, same batches, never re-synchronised: the equilibrium-gate decisions are the
                   oracle's for (at least) the first 20 steps and the losses stay inside the envelope below up to the first
                   differing decision -- after which the two are different training runs;
    re-loaded      every 6th step a second engine takes the ORACLE's weights, BatchNorm buffers and RMSprop state and makes
                   that step: its losses agree to 1e-3 (the one-step parity at trained weights, not only at the initial
                   recipe), its gate decisions are the oracle's, and the losses of the NEXT forward on its own updated
                   weights stay inside the after-one-update bound of tests/test_stage1_gpu.py.
    """
    from fmri_hip.params import ArchConfig
    from fmri_hip.steps import Stage1Step
    from oracle import vaegan_oracle as O
    cfg_o, cfg_e = O.ArchCfg.px64(), ArchConfig.px64()
    B, steps, every = 32, 25, 6
    xs, nz = _bench_data(B, cfg_e.latent_dim)
    P = O.fill_state(O.vaegan_spec(cfg_o), 0, False)
    opts = {n: O.OptState(kind="rmsprop", lr=1e-4) for n in ("encoder", "decoder", "discriminator")}
    st = Stage1Step(cfg_e, DEV)
    st.load_recipe(0, False)
    st2 = Stage1Step(cfg_e, DEV)
    rows, pending, checked = [], None, 0
    for i in range(steps):
        j = i % NBATCH
        x, e, zp = xs[j].to(DEV), nz[j][0].to(DEV), nz[j][1].to(DEV)
        st.step(x, e, zp)
        torch.cuda.synchronize()
        got = st.logs()
        if i % every == 0:
            st2.load_state_dict(_sd_for_engine(P))
            _load_opt_state(st2, opts, O, cfg_o)
            st2.step(x, e, zp)
            torch.cuda.synchronize()
            got2 = st2.logs()
            jn = (i + 1) % NBATCH                  # forward + gate on its own updated weights, next batch, no update
            st2.forward(xs[jn].to(DEV), nz[jn][0].to(DEV), nz[jn][1].to(DEV))
            st2.gate(B)
            torch.cuda.synchronize()
            pending = (i, st2.logs())
        ref = O.stage1_step(P, opts, xs[j], nz[j][0], nz[j][1], cfg_o)["logs"]
        rel = {k: abs(got[k] - ref[k]) / max(abs(ref[k]), 1e-12) for k in LOSSES}
        rows.append((i, got["train_dis"] == ref["train_dis"] and got["train_dec"] == ref["train_dec"], rel, got, ref))
        assert all(np.isfinite(got[k]) for k in LOSSES), (i, got)
        if i % every == 0:
            for k in LOSSES + ("loss_encoder", "loss_decoder", "loss_discriminator"):
                r2 = abs(got2[k] - ref[k]) / max(abs(ref[k]), 1e-12)
                assert r2 < 1e-3, ("re-loaded", i, k, got2[k], ref[k], r2)
            assert got2["train_dis"] == ref["train_dis"] and got2["train_dec"] == ref["train_dec"], (i, got2, ref)
            checked += 1
        elif pending is not None and pending[0] == i - 1:
            after = pending[1]
            print(f"step {i - 1}: re-loaded engine, losses after its own update vs the oracle's: "
                  + " ".join(f"{k} {abs(after[k] - ref[k]) / max(abs(ref[k]), 1e-12):.1e}" for k in LOSSES))
            for k in LOSSES:
                r2 = abs(after[k] - ref[k]) / max(abs(ref[k]), 1e-12)
                assert r2 < 5e-2, ("after one update", i, k, after[k], ref[k], r2)   # batch 4..32 bound of test_stage1_gpu.py
            pending = None
    assert checked == (steps + every - 1) // every
    print("step gate_same " + " ".join(f"{k:>9s}" for k in LOSSES))
    for i, same, rel, got, ref in rows:
        print(f"{i:4d} {str(same):>9s} " + " ".join(f"{rel[k]:9.2e}" for k in LOSSES)
              + f"   kl {got['kl']:.1f}/{ref['kl']:.1f} bce_pred {got['bce_pred']:.2f}/{ref['bce_pred']:.2f}")
    first_gate_split = next((i for i, same, *_ in rows if not same), steps)
    # Envelope.  Step 0 is the one-step parity (tests/test_stage1_gpu.py: 1e-3).  From there two arithmetic models of the
    # same GAN drift apart at the rate the dynamics amplify a 16-bit rounding, and after the first differing gate
    # decision they are two different training runs; the bounds below hold UP TO that step and are 1.5-2x what the 50-step
    # runs in profiles/r05_trajectory_b32.log measured (same gate decisions for the first 45-50 steps; first six steps
    # <= 1.9e-2; nle <= 9.4e-2, mse <= 0.21, kl inside a factor 1.9 -- the largest kl ratios sit on the steps where the
    # ORACLE's own KL jumps 2-3x from one step to the next, the latent excursions of DESIGN 4a; the bce sums, which pass
    # through zero when the discriminator wins, inside a factor 3 or 0.1 nat per sample).
    assert all(rows[0][2][k] < 1e-3 for k in LOSSES), rows[0][2]
    assert first_gate_split >= 20, f"equilibrium gate decisions differ already at step {first_gate_split}"
    for i, same, rel, got, ref in rows[:first_gate_split]:
        for k in LOSSES:
            if i < 6:
                assert rel[k] < 4e-2, (i, k, got[k], ref[k])
            lim = {"nle": 0.15, "mse": 0.35}.get(k)
            if lim is not None:
                assert rel[k] < lim, (i, k, got[k], ref[k])
            else:
                ratio = max(got[k], 1e-12) / max(ref[k], 1e-12)
                near = k.startswith("bce") and abs(got[k] - ref[k]) < 0.1 * B
                assert near or 1 / 3.0 < ratio < 3.0, (i, k, got[k], ref[k])


@pytest.mark.parametrize("kind", ["stage2", "stage3", "stage2-vae", "dual1", "wae1", "stage1-px100", "stage1-betavae"])
def test_other_steps_free_running_next_to_the_oracle(deterministic, kind):
    """(Deterministic reductions, as above.)  Eight free-running steps at batch 8 of the other fused steps (Stage II / III of the cognitive VAE/GAN, Stage II in the
    scripts' `--mode vae`, the Dual WAE + VAE/GAN step, WAE Stage I with Adam) beside the fp32 oracle from the same recipe
    weights and the same per-step noise: every logged loss finite, the equilibrium-gate decisions the oracle's, the
    first-step losses at 1e-3 and the following ones inside the after-k-updates envelope the first-step tests of these
    steps use (5e-2 per update behind the forward; the sign-like first updates make small batches chaotic)."""
    from fmri_hip.params import ArchConfig
    from fmri_hip.steps import CognitiveStep, GanHyper, Stage1Step
    from fmri_hip.wae_steps import DualStage1Step, WaeStep
    from oracle import vaegan_oracle as O
    cfg_o, cfg_e = O.ArchCfg.px64(), ArchConfig.px64()
    if kind == "stage1-px100":                    # the as-shipped 100-px configuration (configs/models_config.py:3-21)
        cfg_o, cfg_e = O.ArchCfg.px100(), ArchConfig.px100()
    B, V, steps = 8, 512, 8
    data = O.synth_batch(B, cfg_o, n_voxels=V, seed=4321, steps=steps)
    x, fm = data["x"].to(DEV), data["fmri"].to(DEV)
    rms = lambda *names: {n: O.OptState(kind="rmsprop", lr=1e-4) for n in names}
    if kind in ("stage2", "stage3", "stage2-vae"):
        stage, mode = (3, "vae-gan") if kind == "stage3" else (2, "vae" if kind.endswith("vae") else "vae-gan")
        st = CognitiveStep(cfg_e, V, DEV, stage, mode=mode)
        st.load_recipe(3, True)
        teacher = O.fill_state(O.vaegan_spec(cfg_o), 3, True)
        P = dict(O.fill_state(O.cognitive_encoder_spec(cfg_o, V), 103, True))
        P.update({k: v for k, v in teacher.items() if k.startswith(("decoder.", "discriminator."))})
        if stage == 2:
            for k, v in teacher.items():
                P["teacher_net." + k] = P[k] if k.startswith(("decoder.", "discriminator.")) else v
        opts = rms("encoder", "decoder", "discriminator")
        ostep = O.stage2_step if stage == 2 else O.stage3_step
        keys = LOSSES

        def both(s):
            nz = data["noise"][s]
            st.step(fm, x, nz[0].to(DEV), nz[1].to(DEV), nz[2].to(DEV))
            ref = ostep(P, opts, data["fmri"], data["x"], nz, cfg_o, V, mode=mode)["logs"]
            if stage == 2:
                for k in teacher:
                    if k.startswith(("decoder.", "discriminator.")):
                        P["teacher_net." + k] = P[k]
            return ref
    elif kind.startswith("stage1"):
        mode, beta = ("beta-vae", 4.0) if kind.endswith("betavae") else ("vae-gan", 1.0)
        st = Stage1Step(cfg_e, DEV, mode=mode, hp=GanHyper(beta=beta))
        st.load_recipe(3, True)
        P = O.fill_state(O.vaegan_spec(cfg_o), 3, True)
        opts = rms("encoder", "decoder", "discriminator")
        keys = LOSSES

        def both(s):
            nz = data["noise"][s]
            st.step(x, nz[0].to(DEV), nz[1].to(DEV))
            return O.stage1_step(P, opts, data["x"], nz[0], nz[1], cfg_o, mode=mode, beta=beta)["logs"]
    elif kind == "dual1":
        st = DualStage1Step(cfg_e, DEV)
        st.load_recipe(8, True)
        P = O.fill_state(O.vaegan_spec(cfg_o), 8, True)
        P.update(O.fill_state(O.wae_discriminator_spec(cfg_o, pre="wae_discriminator."), 208, True))
        opts = rms("encoder", "decoder", "discriminator", "wae_discriminator")
        keys = LOSSES + ("loss_penalty",)

        def both(s):
            nz = data["noise"][s]
            st.step(x, nz[0].to(DEV), nz[1].to(DEV), nz[2].to(DEV))
            return O.dual_stage1_step(P, opts, data["x"], nz, cfg_o)["logs"]
    else:
        st = WaeStep(cfg_e, DEV, 1)
        st.load_recipe(5, False)
        P = O.fill_state(O.encoder_spec(cfg_o) + O.decoder_spec(cfg_o) + O.wae_discriminator_spec(cfg_o), 5, False)
        opts = {"encoder": O.OptState(kind="adam", lr=1e-4), "decoder": O.OptState(kind="adam", lr=1e-4),
                "discriminator": O.OptState(kind="adam", lr=0.5e-4)}
        keys = ("loss_reconstruction", "loss_penalty", "loss_discriminator_fake", "loss_discriminator_real")

        def both(s):
            st.step(x, data["noise"][s, 2].to(DEV))
            return O.wae_stage1_step(P, opts, data["x"], data["noise"][s, 2], cfg_o)["logs"]
    print(f"[{kind}] step " + " ".join(f"{k:>12s}" for k in keys))
    for s in range(steps):
        ref = both(s)
        torch.cuda.synchronize()
        got = st.logs()
        rel = {k: abs(got[k] - ref[k]) / max(abs(ref[k]), 1e-12) for k in keys}
        print(f"[{kind}] {s:4d} " + " ".join(f"{rel[k]:12.2e}" for k in keys))
        assert all(np.isfinite(got[k]) for k in keys), (s, got)
        if "train_dis" in ref and not (got["train_dis"] == ref["train_dis"] and got["train_dec"] == ref["train_dec"]):
            # a differing gate decision ends the comparison (two different runs from here on); it may only come late, and
            # only with a bce mean next to a gate threshold (the 100-px configuration at batch 8 gets there at step 7:
            # bce_orig / B = 0.27 against 0.37 around equilibrium - margin = 0.33)
            assert s >= 5, (kind, s, got, ref)
            print(f"[{kind}] gate decisions differ at step {s}: comparison ends")
            break
        for k in keys:
            pen = k == "loss_penalty"          # scored by the latent discriminator AFTER its (sign-like) update of the step
            lim = (5e-3 if pen else 1e-3) if s == 0 else min(5e-2 * (s + (1 if pen else 0)), 0.3)
            near = abs(got[k] - ref[k]) < 0.02 * B            # sums that pass through zero (bce terms)
            assert rel[k] < lim or near, (kind, s, k, got[k], ref[k], rel[k], lim)
