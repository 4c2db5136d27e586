"""GPU parity of the Stage-II / Stage-III cognitive steps (HIP engine) against the CPU oracle and the
reference-generated golden vectors (tests/golden/stage{2,3}_b4.npz).  Same tolerances as Stage I."""
import os

import numpy as np
import pytest
import torch

import gradcheck

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
LOSS_KEYS = ("loss_encoder", "loss_decoder", "loss_discriminator", "nle", "kl", "mse", "bce_orig", "bce_pred",
             "bce_samp")


def _rel(a, b):
    return abs(a - b) / max(abs(b), 1e-12)


def _terr(got, ref):
    got, ref = got.detach().float().cpu().reshape(-1), ref.detach().float().cpu().reshape(-1)
    return ((got - ref).norm() / (ref.norm() + 1e-20)).item()


def _oracle_state(O, cfg, V, seed, perturb, stage):
    teacher = O.fill_state(O.vaegan_spec(cfg), seed, perturb)
    P = dict(O.fill_state(O.cognitive_encoder_spec(cfg, V), seed + 100, perturb))
    P.update({k: v for k, v in teacher.items() if k.startswith(("decoder.", "discriminator."))})
    if stage == 2:
        for k, v in teacher.items():
            P["teacher_net." + k] = P[k] if k.startswith(("decoder.", "discriminator.")) else v
    return P, teacher


@pytest.mark.parametrize("mode", ["vae-gan", "vae"])
@pytest.mark.parametrize("stage", [2, 3])
def test_cognitive_step_matches_oracle_and_golden(golden_dir, stage, mode):
    """``mode='vae'``: the scripts' `--mode vae` (train_vgan_stage2.py:234-238,362-366; train_vgan_stage3.py:370-374) --
    no teacher net, pixel nle instead of the feature mse -- against the oracle and the reference-generated
    tests/golden/stage{2,3}_vae_b4.npz."""
    from oracle import vaegan_oracle as O
    from fmri_hip.params import ArchConfig
    from fmri_hip.steps import CognitiveStep
    g = np.load(os.path.join(golden_dir, f"stage{stage}_b4.npz" if mode == "vae-gan" else f"stage{stage}_vae_b4.npz"))
    B, V, seed, perturb = int(g["meta/B"]), int(g["meta/V"]), int(g["meta/seed"]), bool(g["meta/perturb"])
    cfg_o = O.ArchCfg.px64()
    steps = 2
    data = O.synth_batch(B, cfg_o, n_voxels=V, seed=1234, steps=steps)
    st = CognitiveStep(ArchConfig.px64(), V, DEV, stage, mode=mode)
    st.load_recipe(seed, perturb)
    P, teacher = _oracle_state(O, cfg_o, V, seed, perturb, stage)
    opts = {n: O.OptState(kind="rmsprop", lr=1e-4) for n in ("encoder", "decoder", "discriminator")}
    ostep_ = O.stage2_step if stage == 2 else O.stage3_step
    ostep = lambda *a, **k: ostep_(*a, mode=mode, **k)
    fm, im = data["fmri"].to(DEV), data["x"].to(DEV)
    for s in range(steps):
        nz = data["noise"][s]
        st.forward(fm, im, nz[0].to(DEV), nz[1].to(DEV), nz[2].to(DEV))
        st.gate(B)
        st.backward()
        outs = {k: v.cpu() for k, v in st.outputs().items()}
        grads = {k: v.cpu() for k, v in st.named_grads().items()}
        st.apply()
        logs = st.logs()
        ref = ostep(P, opts, data["fmri"], data["x"], nz, cfg_o, V, keep_grads=True)
        if stage == 2:
            for k in teacher:
                if k.startswith(("decoder.", "discriminator.")):
                    P["teacher_net." + k] = P[k]
        assert logs["train_dis"] == ref["logs"]["train_dis"] and logs["train_dec"] == ref["logs"]["train_dec"]
        for k in LOSS_KEYS:
            r = _rel(logs[k], ref["logs"][k])
            print(stage, s, k, logs[k], ref["logs"][k], r)
            if s == 0:
                assert r < 1e-3, (s, k, logs[k], ref["logs"][k])
                assert _rel(logs[k], float(g[f"step0/logs/{k}"])) < 1e-3, (k, "golden")
            else:   # after one update (sign-like RMSprop step, see test_stage1_gpu.py)
                assert r < 5e-2, (s, k, logs[k], ref["logs"][k])
        if s == 0:
            for k in ("gt_x", "x_tilde", "x_p", "disc_class", "disc_layer", "mus", "log_variances"):
                e = _terr(outs[k], ref["fw"][k])
                print(stage, "fw", k, e)
                assert e < 1e-2, (k, e)
            P16, _ = _oracle_state(O, cfg_o, V, seed, perturb, stage)
            o16 = {n: O.OptState(kind="rmsprop", lr=1e-4) for n in ("encoder", "decoder", "discriminator")}
            with gradcheck.storage16(O):
                ref16 = ostep(P16, o16, data["fmri"], data["x"], nz, cfg_o, V, keep_grads=True)
            gradcheck.check(grads, ref["grads"], ref16["grads"], f"stage{stage}", tol16=None)
    # BN running statistics / update counters follow the reference's call pattern (golden fingerprints)
    sd = {k: v.cpu() for k, v in st.state_dict().items()}
    keys = [str(k) for k in g["step1/state_keys"]]
    summ = g["step1/state_sum"]
    for i, k in enumerate(keys):
        if "num_batches" in k:
            assert float(sd[k]) == summ[i][1], k
        elif "running_mean" in k or "running_var" in k:
            assert _rel(sd[k].double().norm().item(), summ[i][0]) < 2e-2, k


def test_stage3_px128_bold5000_shape_matches_oracle():
    """BASELINE configs[5] shape (Stage III, 128x128 images, V = 3620 voxels) at a batch the CPU oracle finishes in
    seconds: first-step losses within 1e-3 of the oracle, same gate decision."""
    from oracle import vaegan_oracle as O
    from fmri_hip.params import ArchConfig
    from fmri_hip.steps import CognitiveStep
    B, V, seed = 4, 3620, 5
    cfg_o = O.ArchCfg.px128()
    data = O.synth_batch(B, cfg_o, n_voxels=V, seed=4321, steps=1)
    st = CognitiveStep(ArchConfig.px128(), V, DEV, 3)
    st.load_recipe(seed, True)
    P, _ = _oracle_state(O, cfg_o, V, seed, True, 3)
    opts = {n: O.OptState(kind="rmsprop", lr=1e-4) for n in ("encoder", "decoder", "discriminator")}
    nz = data["noise"][0]
    st.forward(data["fmri"].to(DEV), data["x"].to(DEV), nz[0].to(DEV), nz[1].to(DEV), nz[2].to(DEV))
    st.gate(B)
    st.backward()
    outs = {k: v.cpu() for k, v in st.outputs().items()}
    st.apply()
    logs = st.logs()
    ref = O.stage3_step(P, opts, data["fmri"], data["x"], nz, cfg_o, V, keep_grads=False)
    assert logs["train_dis"] == ref["logs"]["train_dis"] and logs["train_dec"] == ref["logs"]["train_dec"]
    for k in LOSS_KEYS:
        assert _rel(logs[k], ref["logs"][k]) < 1e-3, (k, logs[k], ref["logs"][k])
    for k in ("x_tilde", "disc_class", "disc_layer", "mus"):
        assert _terr(outs[k], ref["fw"][k]) < 1e-2, k


def test_checkpoint_bridge_stage1_to_stage2_to_stage3(tmp_path):
    """The on-disk format either side of the path: a Stage-I step's ``state_dict`` saved with
    ``torch.save`` is a reference-format ``VaeGan`` checkpoint (keys / shapes / fp32, (C,H,W)-ordered fc weights); the
    Stage-II step takes it as its teacher (train_vgan_stage2.py:212-217), trains, saves; the Stage-III step loads that
    file (train_vgan_stage3.py:241).  Every hop is bit-exact on the tensors it carries over."""
    from oracle import vaegan_oracle as O
    from fmri_hip.params import ArchConfig
    from fmri_hip.steps import CognitiveStep, Stage1Step
    cfg, cfg_o = ArchConfig.px64(), O.ArchCfg.px64()
    B, V = 4, 256
    data = O.synth_batch(B, cfg_o, n_voxels=V, seed=1234, steps=1)
    x, fmri = data["x"].to(DEV), data["fmri"].to(DEV)
    nz = [data["noise"][0, i].to(DEV) for i in range(3)]
    s1 = Stage1Step(cfg, DEV)
    s1.load_recipe(0, True)
    s1.step(x, nz[0], nz[1])
    f1 = str(tmp_path / "vgan_stage1.pth")
    torch.save({k: v.cpu() for k, v in s1.state_dict().items()}, f1)
    ck1 = torch.load(f1, map_location="cpu")
    spec = {k: tuple(s) for k, s, _ in O.vaegan_spec(cfg_o)}
    assert list(ck1.keys()) == list(spec.keys())                       # reference key order
    for k, v in ck1.items():
        assert tuple(v.shape) == spec[k] and (v.dtype == torch.float32 or "num_batches" in k), k
    # fresh Stage-I engine from the file: identical state
    s1b = Stage1Step(cfg, DEV)
    s1b.load_state_dict(ck1)
    for k, v in s1b.state_dict().items():
        assert torch.equal(v.cpu(), ck1[k]), k
    # Stage II: the file is the teacher
    s2 = CognitiveStep(cfg, V, DEV, stage=2)
    s2.load_recipe(5, True)
    s2.load_teacher(ck1)
    sd2 = {k: v.cpu() for k, v in s2.state_dict().items()}
    for k, v in ck1.items():
        pre, rest = k.split(".", 1)
        assert torch.equal(sd2["teacher_net." + k], v), k
        if pre in ("decoder", "discriminator"):
            assert torch.equal(sd2[k], v), k
    s2.step(fmri, x, nz[0], nz[1], nz[2])
    logs = s2.logs()
    assert all(np.isfinite(logs[k]) for k in ("loss_encoder", "loss_discriminator"))
    f2 = str(tmp_path / "vgan_stage2.pth")
    torch.save({k: v.cpu() for k, v in s2.state_dict().items()}, f2)
    ck2 = torch.load(f2, map_location="cpu")
    # Stage III loads the Stage-II file
    s3 = CognitiveStep(cfg, V, DEV, stage=3)
    s3.load_state_dict(ck2)
    sd3 = {k: v.cpu() for k, v in s3.state_dict().items()}
    for k, v in sd3.items():
        assert torch.equal(v, ck2[k]), k
    s3.step(fmri, x, nz[0], nz[1])
    assert np.isfinite(s3.logs()["loss_decoder"])


def test_stage3_px128_matches_reference_golden(golden_dir):
    """BASELINE configs[4] shape (128 px, V = 3620), Stage III of the VAE/GAN: first-step losses against the numbers the
    real reference produced (tests/golden/stage3_px128_b2.npz)."""
    from oracle import vaegan_oracle as O
    from fmri_hip.params import ArchConfig
    from fmri_hip.steps import CognitiveStep
    g = np.load(os.path.join(golden_dir, "stage3_px128_b2.npz"))
    B, V, seed, perturb = int(g["meta/B"]), int(g["meta/V"]), int(g["meta/seed"]), bool(g["meta/perturb"])
    cfg_o = O.ArchCfg.px128()
    data = O.synth_batch(B, cfg_o, n_voxels=V, seed=1234, steps=1)
    st = CognitiveStep(ArchConfig.px128(), V, DEV, 3)
    st.load_recipe(seed, perturb)
    nz = data["noise"][0]
    st.step(data["fmri"].to(DEV), data["x"].to(DEV), nz[0].to(DEV), nz[1].to(DEV), nz[2].to(DEV))
    logs = st.logs()
    for k in LOSS_KEYS:
        want = float(g[f"step0/logs/{k}"])
        print(k, logs[k], want, _rel(logs[k], want))
        assert _rel(logs[k], want) < 1e-3, (k, logs[k], want)
    assert logs["train_dis"] == bool(g["step0/logs/train_dis"]) and logs["train_dec"] == bool(g["step0/logs/train_dec"])


def test_stage2_full_batch_first_step_matches_oracle():
    """BASELINE configs[2] (Stage II, V = 4096, B = 256, decoder frozen): first-step losses against the CPU oracle run
    live on the same seeded inputs."""
    from oracle import vaegan_oracle as O
    from fmri_hip.params import ArchConfig
    from fmri_hip.steps import CognitiveStep
    B, V, seed = 256, 4096, 1
    cfg_o = O.ArchCfg.px64()
    data = O.synth_batch(B, cfg_o, n_voxels=V, seed=1234, steps=1)
    st = CognitiveStep(ArchConfig.px64(), V, DEV, 2)
    st.load_recipe(seed, True)
    P, _ = _oracle_state(O, cfg_o, V, seed, True, 2)
    opts = {n: O.OptState(kind="rmsprop", lr=1e-4) for n in ("encoder", "decoder", "discriminator")}
    nz = data["noise"][0]
    st.step(data["fmri"].to(DEV), data["x"].to(DEV), nz[0].to(DEV), nz[1].to(DEV), nz[2].to(DEV))
    logs = st.logs()
    ref = O.stage2_step(P, opts, data["fmri"], data["x"], nz, cfg_o, V)
    for k in LOSS_KEYS:
        print(k, logs[k], ref["logs"][k], _rel(logs[k], ref["logs"][k]))
        assert _rel(logs[k], ref["logs"][k]) < 1e-3, (k, logs[k], ref["logs"][k])
    assert logs["train_dis"] and not logs["train_dec"]
