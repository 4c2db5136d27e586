#!/usr/bin/env python3
"""Headline benchmark: Stage-I VAE/GAN training step, 64x64x3 stimuli, latent 128, batch 256 per GPU
(BASELINE.json configs[1]), images/sec, on N MI355X of one node.

    python bench.py --gpus 1 --steps 100 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one full pass of the hot path over one synthetic batch already resident in HBM:
forward (encoder, 2x decoder, fused REC+GAN discriminator), losses, two-stream backward giving the three
gradient sets, SUM all-reduce over RCCL (N > 1), gated RMSprop updates, fp16 weight re-pack.
Prints ONE JSON line on rank 0.

--workload selects the other BASELINE configs (same contract, their own metric names):
    stage1        configs[1]  Stage-I VAE/GAN 64 px, B = 256 per GPU                       (default, the headline)
    stage2        configs[2]  Stage-II cognitive VAE/GAN, V = 4096, decoder frozen, B = 256 per GPU
    dual1         configs[3]  Stage-I WAE / Dual-GAN, B = 128 per GPU (512 over 4 GPUs)
    stage3_px128  configs[4]  Stage-III cognitive WAE, 128 px stimuli, V = 3620, B = 128 per GPU (1024 over 8 GPUs)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "thesis-fmri-reconstruction_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

FLOP_PER_IMAGE = 13.81e9          # SURVEY 8(d) / BASELINE.md 3: algorithmic Stage-I step FLOPs per image
MFMA_PEAK_TFLOPS = 2500.0         # dense fp16/bf16 MFMA peak, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0             # HBM3E peak, MI355X_MICROARCH.md
NBATCH = 8                        # synthetic batches the timed steps rotate through


def host_cores() -> int:
    """CPU threads this process may really use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p) + 0.5)))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("FMRI_CPU_THREADS", "64"))))


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def _time_oracle(O, cfg, batch, literal, budget_s, max_steps):
    """images/sec of the CPU oracle's Stage-I step at ``batch`` (one untimed warm-up step, then as many timed steps as fit
    ``budget_s`` seconds, at least one)."""
    P = O.fill_state(O.vaegan_spec(cfg), 0, False)
    data = O.synth_batch(batch, cfg, seed=1234, steps=1)
    opts = {n: O.OptState(kind="rmsprop", lr=1e-4) for n in ("encoder", "decoder", "discriminator")}
    args = (data["x"], data["noise"][0, 0], data["noise"][0, 1], cfg)
    t0 = time.perf_counter()
    O.stage1_step(P, opts, *args, literal=literal)           # warm-up
    warm = time.perf_counter() - t0
    steps = max(1, min(max_steps, int(budget_s / max(warm, 1e-3))))
    t0 = time.perf_counter()
    for i in range(steps):
        O.stage1_step(P, opts, *args, literal=literal)
    dt = time.perf_counter() - t0
    log(f"cpu baseline: batch {batch} {'literal' if literal else 'pruned'}: warm-up {warm:.1f}s, {steps} steps in {dt:.1f}s")
    return batch * steps / dt, steps


def cpu_baseline(batch: int, steps: int):
    """Oracle (CPU restatement pinned to the reference) timed on this host's cores.  ``value`` = the 'literal' variant
    (three full backward traversals, what train_vgan_stage1.py:410-432 does) at BASELINE configs[0]'s batch 32; the other
    rows SURVEY 8(d) asks for ride along in ``variants``: the 'pruned' restatement (one traversal per gradient set, each
    through its own sub-network only) at batch 32 and the literal step at the headline batch 256."""
    from oracle import vaegan_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    cfg = O.ArchCfg.px64()
    log(f"cpu baseline: {cores} threads")
    v_lit, n_lit = _time_oracle(O, cfg, batch, True, 14.0, steps)
    v_pru, n_pru = _time_oracle(O, cfg, batch, False, 7.0, steps)
    v_big, n_big = _time_oracle(O, cfg, 256, True, 5.0, 1)
    return dict(value=round(v_lit, 3), unit="images/sec", cores=cores, kind="port",
                sample=f"{n_lit} literal Stage-I steps (3 full backward traversals) of the CPU oracle at batch {batch} "
                       f"(BASELINE configs[0]) after 1 warm-up, torch {torch.__version__} fp32, {cores} threads",
                variants=[dict(variant="pruned", batch=batch, steps=n_pru, value=round(v_pru, 3)),
                          dict(variant="literal", batch=256, steps=n_big, value=round(v_big, 3))])


def _family_traffic(kernels, family):
    """Launch-weighted bytes per launch of the kernel family ``family`` in a tools/pmc_traffic.py-style table."""
    want = family.replace(" ", "")
    tot = cnt = 0
    for kname, v in kernels.items():
        n = kname.replace(" ", "")
        n = n[4:] if n.startswith("void") else n
        n = n.split("(")[0].split("<")[0]
        if n == want:
            tot += v["bytes_per_launch"] * v["launches"]
            cnt += v["launches"]
    return int(tot / cnt) if cnt else None


def live_pmc_traffic(family, timeout_s=75):
    """The traffic counters of THIS box: two rocprofv3 child runs of this script (``--pmc FETCH_SIZE`` and ``--pmc
    WRITE_SIZE``, each its own run with ``--kernel-trace`` only, five eagerly issued one-stream steps) summarised like
    tools/pmc_traffic.py: (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per launch of the family (KiB units, the gfx950 read
    correction: MI355X_MICROARCH.md).  Children, never an exec; None when rocprofv3 is missing, a pass fails or times out
    (the committed passes of the same command are the fallback)."""
    import csv
    import shutil
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None
    tmp = tempfile.mkdtemp(prefix="fmri_pmc_", dir="/tmp")
    per = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [rocprof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "x", "--",
                   sys.executable, os.path.abspath(__file__), "--steps", "3", "--warmup", "2", "--eager", "--serial",
                   "--no-cpu-baseline", "--no-hbm-rows", "--no-pmc", "--no-gate-pass"]
            # a process group of its own: on a timeout the profiler AND the python it started are killed together
            proc = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                                    stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = proc.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except OSError:
                    pass
                proc.wait()
                raise
            files = [os.path.join(d, f) for d, _, fs in os.walk(out) for f in fs if f.endswith("counter_collection.csv")]
            if rc != 0 or not files:
                log(f"pmc pass {counter}: rc {rc}, {len(files)} counter files")
                return None
            acc = {}
            for row in csv.DictReader(open(files[0])):
                if row["Counter_Name"] == counter:
                    a = acc.setdefault(row["Kernel_Name"], [0, 0.0])
                    a[0] += 1
                    a[1] += float(row["Counter_Value"])
            per[counter] = acc
        kernels = {}
        for name in set(per["FETCH_SIZE"]) | set(per["WRITE_SIZE"]):
            nf, kf = per["FETCH_SIZE"].get(name, (0, 0.0))
            nw, kw = per["WRITE_SIZE"].get(name, (0, 0.0))
            kernels[name] = {"launches": max(nf, nw),
                             "bytes_per_launch": 2.0 * 1024.0 * kf / max(nf, 1) + 1024.0 * kw / max(nw, 1)}
        return _family_traffic(kernels, family)
    except (subprocess.TimeoutExpired, OSError, KeyError, ValueError) as e:
        log(f"live pmc passes failed: {type(e).__name__}: {e}")
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pmc_traffic(family):
    """(2*FETCH_SIZE + WRITE_SIZE) * 1024 per launch of the kernel family ``family`` (kernel name without template
    arguments), launch-weighted over its instantiations, from the committed rocprofv3 PMC passes of this same command
    (tools/pmc_traffic.py); None when no committed pass holds the family."""
    want = family.replace(" ", "")
    for name in ("r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", name)
        try:
            with open(path) as f:
                kernels = json.load(f)["kernels"]
        except (OSError, ValueError, KeyError):
            continue
        tot = cnt = 0
        for kname, v in kernels.items():
            n = kname.replace(" ", "")
            n = n[4:] if n.startswith("void") else n
            n = n.split("(")[0].split("<")[0]
            if n == want:
                tot += v["bytes_per_launch"] * v["launches"]
                cnt += v["launches"]
        if cnt:
            return int(tot / cnt), name
    return None, None


# A launch's kernel family: the kernel the library routes it to (lib.note(kernel=...), template arguments dropped) or, for
# the single-kernel entry points, the entry point itself.  Coarse groups for the per-step summary:
def _family_of(entry, note):
    if note and note.get("kernel"):
        return note["kernel"].split("<")[0]
    return entry


def _group_of(entry, note):
    if note and note.get("kernel"):
        return "weight-gradient GEMMs" if "wgrad" in note["kernel"] else "conv / deconv / dense forward + data-gradient GEMMs"
    if entry.startswith("fmri_bn_") or entry == "fmri_act_bwd":
        return "BatchNorm + activation passes"
    if entry in ("fmri_apply_batch", "fmri_transpose_f16", "fmri_pack_weight", "fmri_pack_weight_batch",
                 "fmri_unpack_grad", "fmri_rmsprop_dev", "fmri_adam_dev", "fmri_rmsprop", "fmri_adam", "fmri_counter_inc"):
        return "update: gradient map + optimizer + fp16 weight copies"
    if entry == "fmri_reduce_slabs":
        return "split-K slab sums of the dense layers"
    return "losses, latent, gate, layout casts"


def family_table(prof, steps, key=_family_of):
    """HIP-event time of every library launch of ``steps`` one-stream steps, grouped by ``key`` (kernel family or coarse
    group).  GEMM families carry algorithmic FLOPs (TFLOP/s against the dense fp16 MFMA peak), the streaming kernels
    algorithmic bytes (GB/s against the HBM peak) where the caller annotated them.  An event pair brackets the launch's
    slot on the stream, i.e. kernel time + the gap to the next launch: a few per cent on the large kernels, most of the
    ~5 us launches."""
    fam = {}
    for entry, note, e0, e1 in prof:
        f = fam.setdefault(key(entry, note), dict(ms=0.0, n=0, flops=0.0, bytes=0.0, ms_b=0.0))
        ms = e0.elapsed_time(e1)
        f["ms"] += ms
        f["n"] += 1
        if note:
            f["flops"] += note.get("flops", 0.0) or 0.0
            if note.get("bytes"):
                f["bytes"] += note["bytes"]
                f["ms_b"] += ms
    rows = []
    for name, f in sorted(fam.items(), key=lambda kv: -kv[1]["ms"]):
        row = {"family": name, "ms_per_step": round(f["ms"] / steps, 3), "launches_per_step": round(f["n"] / steps, 1)}
        if f["flops"] > 0:
            row["tflops"] = round(f["flops"] / (f["ms"] * 1e-3) / 1e12, 1)
            row["frac_of_mfma_peak"] = round(row["tflops"] / MFMA_PEAK_TFLOPS, 4)
        if f["bytes"] > 0 and f["ms_b"] > 0:
            # (GEMM families carry both: the small-channel and dense ones -- K = 75, N = 3, M = batch rows -- are priced
            # by their algorithmic bytes, input + output + weights once, not by the MFMA peak)
            row["gb_s"] = round(f["bytes"] / (f["ms_b"] * 1e-3) / 1e9, 1)
            row["frac_of_hbm_peak"] = round(row["gb_s"] / HBM_PEAK_GBS, 4)
        rows.append((row, f))
    return rows


# ------------------------------------------------------------------------------------------------------------------
# workloads: name -> (metric, unit, description, batch per GPU, builder)
# a builder returns (step object, run(i) closure issuing one eager step on synthetic batch i % NBATCH, FLOPs per sample)
# ------------------------------------------------------------------------------------------------------------------
def _images(rs, B, px, dev):
    return torch.from_numpy(rs.uniform(-1, 1, (B, 3, px, px)).astype(np.float32)).to(dev)


def _noise(rs, k, B, Z, dev):
    return torch.from_numpy(rs.standard_normal((k, B, Z)).astype(np.float32)).to(dev)


def build_stage1(dev, B, rank, dist_on, sync_bn):
    from fmri_hip.params import ArchConfig, stage1_step_flops
    from fmri_hip.steps import Stage1Step
    cfg = ArchConfig.px64()
    st = Stage1Step(cfg, dev, distributed=dist_on, sync_bn=sync_bn)
    st.load_recipe(0, False)
    xs = [_images(np.random.RandomState(1234 + 97 * i + rank), B, 64, dev) for i in range(NBATCH)]
    nz = [_noise(np.random.RandomState(1236 + 97 * i + rank), 2, B, cfg.latent_dim, dev) for i in range(NBATCH)]
    return st, (lambda i: st.step(xs[i % NBATCH], nz[i % NBATCH][0], nz[i % NBATCH][1])), stage1_step_flops(cfg), \
        [xs, [n[0] for n in nz], [n[1] for n in nz]]


def build_stage2(dev, B, rank, dist_on, sync_bn):
    from fmri_hip.params import ArchConfig, stage2_step_flops
    from fmri_hip.steps import CognitiveStep
    cfg, V = ArchConfig.px64(), 4096
    st = CognitiveStep(cfg, V, dev, 2, distributed=dist_on, sync_bn=sync_bn)
    st.load_recipe(1, False)
    xs = [_images(np.random.RandomState(1234 + 97 * i + rank), B, 64, dev) for i in range(NBATCH)]
    fm = [torch.from_numpy(np.random.RandomState(1235 + 97 * i + rank).standard_normal((B, V)).astype(np.float32)).to(dev)
          for i in range(NBATCH)]
    nz = [_noise(np.random.RandomState(1236 + 97 * i + rank), 3, B, cfg.latent_dim, dev) for i in range(NBATCH)]

    def run(i):
        j = i % NBATCH
        return st.step(fm[j], xs[j], nz[j][0], nz[j][1], nz[j][2])
    return st, run, stage2_step_flops(cfg, V), [fm, xs, [n[0] for n in nz], [n[1] for n in nz], [n[2] for n in nz]]


def build_dual1(dev, B, rank, dist_on, sync_bn):
    from fmri_hip.params import ArchConfig, dual1_step_flops
    from fmri_hip.wae_steps import DualStage1Step
    cfg = ArchConfig.px64()
    st = DualStage1Step(cfg, dev, distributed=dist_on, sync_bn=sync_bn)
    st.load_recipe(8, False)
    xs = [_images(np.random.RandomState(1234 + 97 * i + rank), B, 64, dev) for i in range(NBATCH)]
    nz = [_noise(np.random.RandomState(1236 + 97 * i + rank), 3, B, cfg.latent_dim, dev) for i in range(NBATCH)]

    def run(i):
        j = i % NBATCH
        return st.step(xs[j], nz[j][0], nz[j][1], nz[j][2])
    return st, run, dual1_step_flops(cfg), [xs, [n[0] for n in nz], [n[1] for n in nz], [n[2] for n in nz]]


def build_stage3_px128(dev, B, rank, dist_on, sync_bn):
    """configs[4]: the WAE Stage-III step (train/train_wae_stage3.py) at 128 px / V = 3620: decoder + latent
    discriminator trained on the pixel loss + latent penalty; FLOPs per sample = C + decoder forward, weight and data
    gradient (3 D) + the latent discriminator."""
    from fmri_hip.params import ArchConfig, forward_flops
    from fmri_hip.wae_steps import WaeStep
    cfg, V = ArchConfig.px128(), 3620
    st = WaeStep(cfg, dev, 3, V, distributed=dist_on, sync_bn=sync_bn)
    st.load_recipe(7, False)
    xs = [_images(np.random.RandomState(1234 + 97 * i + rank), B, 128, dev) for i in range(NBATCH)]
    fm = [torch.from_numpy(np.random.RandomState(1235 + 97 * i + rank).standard_normal((B, V)).astype(np.float32)).to(dev)
          for i in range(NBATCH)]
    f = forward_flops(cfg, V)
    return st, (lambda i: st.step(xs[i % NBATCH], fmri=fm[i % NBATCH])), f["C"] + f["E"] + 3.0 * f["D"] + 8.0 * f["W"], \
        [xs, None, fm]


WORKLOADS = {
    "stage1": ("images/sec Stage-I VAE/GAN 64x64 bs256", "images/sec",
               "Stage-I VAE/GAN training step, 64x64x3 random images, latent 128, RMSprop x3, random-init weights "
               "(BASELINE configs[1]; 16-bit type = fp16 storage with fp32 accumulation instead of the bf16 the config "
               "names: same dense MFMA peak, 3 more mantissa bits for the 1e-3 loss bar)", 256, build_stage1),
    "stage2": ("samples/sec Stage-II cognitive VAE/GAN V4096 64x64 bs256", "samples/sec",
               "Stage-II cognitive VAE/GAN step, synthetic 4096-voxel fMRI -> 64x64 image, teacher distillation, decoder "
               "frozen, RMSprop x2 (BASELINE configs[2])", 256, build_stage2),
    "dual1": ("images/sec Stage-I WAE/Dual-GAN 64x64 bs128/GPU", "images/sec",
              "Stage-I Dual WAE + VAE/GAN step (VAE/GAN step + latent discriminator phase + latent penalty), 64x64x3, "
              "128 images per GPU (BASELINE configs[3]: 512 over 4 GPUs)", 128, build_dual1),
    "stage3_px128": ("samples/sec Stage-III cognitive WAE 128x128 V3620 bs128/GPU", "samples/sec",
                     "Stage-III cognitive WAE step, BOLD5000-shaped 3620-voxel fMRI -> 128x128 image, decoder + latent "
                     "discriminator trained, Adam (BASELINE configs[4]: 1024 over 8 GPUs)", 128, build_stage3_px128),
}


def hbm_rows(dev, B):
    """GB/s of the HBM-bound kernels of the path (SURVEY 8d asks for them beside the MFMA roofline), each timed alone
    with HIP events over 20 launches on data larger than the caches' working set of the step: the cognitive encoder's
    Linear(V -> 1024) at batch B, the BatchNorm apply / two-stream backward passes on the largest activation
    (discriminator block 1: 3B x 32 x 32 x 128) and the fused RMSprop update of the encoder's 18 M parameters."""
    from fmri_hip import lib, ops
    from fmri_hip.params import ArchConfig
    from fmri_hip.nets import CognitiveEncoderNet
    P = lib.ptr
    out = {}

    def timed(fn, nbytes, reps=20, inner=10):
        """GPU time per call: `inner` calls recorded into a HIP graph and replayed `reps` times (a loop of eagerly issued
        Python calls measures the host for kernels of a few microseconds: the dense layer's two launches cost ~15 us of
        Python and ~14 us of GPU)."""
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(inner):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / (reps * inner)
        return dict(gb_s=round(nbytes / ms / 1e6, 1), frac_of_peak=round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 3),
                    us=round(ms * 1e3, 1), mbytes=round(nbytes / 1e6, 1))
    V = 4096
    cog = CognitiveEncoderNet(ArchConfig.px64(), V, dev)
    cog.group.load_recipe(np.random.RandomState(3), False)
    x = torch.randn(B, V, device=dev).half()
    cog.fc1.forward(x)
    out["cognitive_fc1_fwd"] = timed(lambda: cog.fc1.forward(x), 2 * B * V + 2 * V * 1024 + 2 * B * 1024)
    M, C = 3 * B * 32 * 32, 128
    raw = torch.randn(M, C, device=dev).half()
    y = torch.empty_like(raw)
    sc = torch.ones(C, device=dev)
    sh = torch.zeros(C, device=dev)
    out["bn_apply"] = timed(lambda: lib.call("fmri_bn_apply", P(raw), P(y), M, C, P(sc), P(sh), 1), 4 * M * C)
    dy2 = torch.randn(2 * M, C, device=dev).half()
    dx2 = torch.empty_like(dy2)
    s4 = torch.zeros(4, C, device=dev)
    out["bn_bwd_apply_2streams"] = timed(
        lambda: lib.call("fmri_bn_bwd_apply2", P(raw), P(dy2), P(dx2), M, C, float(M), P(sh), P(sc), P(sc), P(sh), 1,
                         P(s4)), 10 * M * C)
    n = 18_071_360
    p, g, sq = (torch.randn(n, device=dev) for _ in range(3))
    sq.abs_()
    lr = torch.full((1,), 1e-4, device=dev)
    out["rmsprop_encoder"] = timed(lambda: lib.call("fmri_rmsprop_dev", P(p), P(g), P(sq), n, P(lr), 0.9, 1e-8, 1.0,
                                                    None, 0.0, None), 20 * n)
    return out


def _launch_ranks(n: int, json_fd: int) -> int:
    """Run this script under ``python -m torch.distributed.run`` with ``n`` ranks on 127.0.0.1 as a child process (never
    an exec: the launcher must stay a child of a process that owns no GPU state), pass the child's stdout -- rank 0's one
    JSON line -- through, return its exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("launching: " + " ".join(cmd))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    if r.stdout:
        os.write(json_fd, r.stdout)
    return r.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="stage1")
    ap.add_argument("--batch", type=int, default=0, help="samples per GPU (weak scaling); 0 = the workload's own")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hbm-rows", action="store_true")
    ap.add_argument("--no-gate-pass", dest="gate_pass", action="store_false",
                    help="skip the extra timed pass with the engine's default gate_skip=True")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not run the two rocprofv3 counter passes for roofline.traffic (then the committed passes of "
                         "the same command under profiles/ are quoted)")
    ap.add_argument("--sync-bn", action="store_true",
                    help="data-parallel runs: all-reduce the BatchNorm partial sums (statistics of the GLOBAL batch, one "
                         "small collective per BatchNorm call).  Default: per-rank statistics over the rank's own batch, "
                         "i.e. what the single-GPU reference computes per batch and what torch DDP does by default")
    ap.add_argument("--local-bn", action="store_true", help="(default; kept for compatibility)")
    ap.add_argument("--serial", action="store_true",
                    help="keep every launch on one stream (no weight-gradient side stream): the mode the per-kernel "
                         "roofline pass always uses, and the one to profile with rocprofv3 for per-kernel durations")
    ap.add_argument("--eager", action="store_true", help="issue every launch from Python (no graph capture)")
    ap.add_argument("--graph", action="store_true", help="always replay the captured graph(s) (default: whichever of "
                    "eager / graph replay probes faster on this host)")
    a = ap.parse_args()

    # stdout must carry exactly one JSON line: native libraries (the RCCL version banner at communicator creation)
    # print to file descriptor 1, so fd 1 is pointed at stderr for the run and the result goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and world != a.gpus:
        if "WORLD_SIZE" in os.environ or "RANK" in os.environ:
            raise SystemExit(f"--gpus {a.gpus} under a launcher with WORLD_SIZE={world}: the rank count must match")
        # plain `python bench.py --gpus N`: this process has not touched the GPU; it starts the N ranks as a CHILD
        # (torch.distributed.run, one process per GPU), relays the ranks' single JSON line and exits with the child's code
        sys.exit(_launch_ranks(a.gpus, json_fd))
    if os.environ.get("FMRI_REHEARSE_ON_ONE_GPU") == "1":
        local = 0                      # all ranks share device 0 (with FMRI_DIST_BACKEND=gloo): control-flow rehearsal
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    force_dist = os.environ.get("FMRI_FORCE_DIST") == "1"      # 1-rank rehearsal of the RCCL path
    if world > 1 or force_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:                 # plain `FMRI_FORCE_DIST=1 python bench.py`: a one-rank env:// rendezvous
            for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29577")):
                os.environ.setdefault(k, v)
        backend = os.environ.get("FMRI_DIST_BACKEND", "nccl")          # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    # under a profiler (rocprofv3 preloads its tool library) the run must not start profiler children of its own
    if "rocprofiler" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF") for k in os.environ):
        a.no_pmc = True
    if a.serial:
        os.environ["FMRI_SIDE_STREAM"] = "off"
    from fmri_hip import lib, ops
    lib.load()

    metric, unit, desc, wl_batch, builder = WORKLOADS[a.workload]
    B = a.batch or wl_batch
    st, run_i, flop_per_sample, rot = builder(dev, B, rank, world > 1 or force_dist, a.sync_bn)
    # The headline runs EVERY GEMM of the step in every step (what `gflop_per_sample` counts): the engine's default of
    # conditioning a sub-network's weight gradients on the equilibrium gate (Stage1Step(gate_skip=True) -- the reference
    # does not run `loss_discriminator.backward()` when the gate switches the discriminator off) is off here, and is
    # measured separately below (`gate_skip` in the output line).
    has_gate = hasattr(st, "gate_skip")
    if has_gate:
        st.gate_skip = False

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log(f"rank {rank}/{world}: {a.workload} built, batch {B}; host cores {host_cores()} (cpu_count {os.cpu_count()})")
    for i in range(a.warmup):
        run_i(i)
        if i == 0:
            torch.cuda.synchronize()
            log("first step done")
    barrier()
    log("warm-up done")
    # Launch modes.  eager: every launch issued from Python, over NBATCH rotating synthetic batches.  The step can also be
    # recorded into HIP graphs (graph: one-stream full step; hybrid, headline workload only: recorded forward + eager
    # two-stream backward): the recorded inputs are static buffers, refreshed from the rotating batches by one device
    # copy per input and step inside the timed region.  In multi-process runs the RCCL collectives stay eager calls
    # between the graph segments.  A short probe picks the fastest mode, the same on every rank.
    modes = {"eager": run_i}
    single = world == 1 and not force_dist
    if not a.eager and hasattr(st, "capture"):
        sbuf = [None if r is None else r[0].clone() for r in rot]

        def staged(replay):
            def run(i):
                j = i % NBATCH
                for buf, r in zip(sbuf, rot):
                    if buf is not None:
                        buf.copy_(r[j])
                return replay()
            return run
        # multi-process runs: the full step as graph segments between eager collectives is recorded only on request
        # (--graph).  It has never measured faster than the eager / hybrid step there, and a failed stream capture next to
        # the collective backend's watchdog thread ends the process rather than raising (steps._SegmentRecorder).
        if single or a.graph:
            try:
                g = st.capture(*sbuf)
                modes["graph"] = staged(g)
                modes["graph"](0)
                log("step captured into HIP graph(s)")
            except Exception as e:               # capture is an optimisation: fall back to eager launches
                log(f"graph capture failed ({type(e).__name__}: {e}); continuing with eager launches")
                modes.pop("graph", None)
        if (a.workload == "stage1" and (single or not a.sync_bn) and ops._SIDE["on"] and not a.graph):
            # hybrid: recorded forward + eagerly issued two-stream backward (half the Python work of a step)
            try:
                h = st.capture_forward(*sbuf)
                modes["hybrid"] = staged(h)
                modes["hybrid"](0)
            except Exception as e:
                log(f"forward capture failed ({type(e).__name__}: {e})")
                modes.pop("hybrid", None)

    if world > 1:
        # a launch mode is only usable if every rank managed to record it (the probe below is a collective decision)
        have = torch.tensor([float("graph" in modes), float("hybrid" in modes)], dtype=torch.float32, device=dev)
        dist.all_reduce(have, op=dist.ReduceOp.MIN)
        for i, k in enumerate(("graph", "hybrid")):
            if have[i].item() < 1.0:
                modes.pop(k, None)

    def probe(fn, n=6):
        barrier()
        t = time.perf_counter()
        for i in range(n):
            fn(i)
        barrier()
        tt = torch.tensor([time.perf_counter() - t], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)      # every rank takes the same decision
        return float(tt.item()) / n

    mode = "eager"
    if a.graph and "graph" in modes:
        mode = "graph"
    elif len(modes) > 1:
        times = {k: probe(fn) for k, fn in modes.items()}
        mode = min(times, key=times.get)
        log("probe: " + ", ".join(f"{k} {1e3 * v:.2f} ms/step" for k, v in times.items()) + f" -> {mode}")
    run = modes[mode]
    barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        run(i)
    barrier()
    dt = time.perf_counter() - t0
    log(f"timed region done: {1e3 * dt / a.steps:.2f} ms/step ({mode})")
    logs = st.logs()

    def require_finite(what, lg):
        """A step whose losses are not finite is not a measurement: no `value` is printed and the run fails."""
        bad = {k: v for k, v in lg.items() if isinstance(v, float) and not np.isfinite(v)}
        if bad:
            log(f"NON-FINITE losses after {what}: {bad}")
            if rank == 0:
                os.write(json_fd, (json.dumps({"metric": metric, "error": f"non-finite losses after {what}",
                                               "losses": {k: repr(v) for k, v in lg.items() if isinstance(v, float)},
                                               "launch": mode}) + "\n").encode())
            sys.exit(3)
    require_finite(f"the timed region ({a.warmup} warm-up + {a.steps} timed steps, launch mode {mode})", logs)
    # the same steps with the engine's default gate_skip=True (launches that are recorded into a full-step graph keep the
    # setting they were recorded with, so this pass uses the hybrid / eager launches), and how often the gate trained what
    gate_out = None
    if has_gate and a.gate_pass and a.workload in ("stage1", "stage2"):
        grun = modes.get("hybrid", run_i)
        st.gate_skip = True
        for i in range(10):
            grun(i)
        barrier()
        gsteps, seen = min(a.steps, 100), []
        tg = time.perf_counter()
        for i in range(gsteps):
            grun(i)
            seen.append(st.flags.clone())
        barrier()
        tgd = torch.tensor([time.perf_counter() - tg], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tgd, op=dist.ReduceOp.MAX)
        frac = torch.stack(seen).float().mean(0).tolist()
        st.gate_skip = False
        require_finite("the gate-skip pass", st.logs())
        gate_out = {"ms_per_step": round(1e3 * float(tgd.item()) / gsteps, 3),
                    "value": round(world * B * gsteps / float(tgd.item()), 1), "steps": gsteps,
                    "launch": "hybrid" if "hybrid" in modes else "eager",
                    "steps_training_discriminator": round(frac[0], 3), "steps_training_decoder": round(frac[1], 3),
                    "note": "engine default: weight-gradient GEMMs of a sub-network the equilibrium gate does not train in "
                            "a step retire at once (fmri_wgrad_if; the reference skips that loss.backward(), "
                            "train_vgan_stage1.py:420-431).  Data dependent -- on this synthetic data the gate keeps the "
                            "discriminator off in most steps -- hence not the headline: `value` runs every GEMM"}
        log(f"gate-skip pass: {gate_out['ms_per_step']} ms/step, discriminator trained in {frac[0]:.2f}, decoder in "
            f"{frac[1]:.2f} of the steps")
    # dominant-kernel timing: HIP events around every fmri_igemm launch over a few eagerly issued steps (events cannot be
    # placed inside a replayed graph).  The side stream is switched off for this pass: next to concurrently running
    # weight-gradient kernels a launch's duration says nothing about the kernel.
    ops.join_side()
    side_was, ops._SIDE["on"] = ops._SIDE["on"], False
    lib.PROFILE = [] if rank == 0 else None
    prof_steps = min(a.steps, 5)
    e_first, e_last = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e_first.record()
    for i in range(prof_steps):
        run_i(i)
    e_last.record()
    barrier()
    prof, lib.PROFILE = lib.PROFILE, None
    require_finite("the per-kernel profiling pass", st.logs())
    prof_ms_per_step = e_first.elapsed_time(e_last) / max(prof_steps, 1)
    ops._SIDE["on"] = side_was
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        value = world * B * a.steps / dt
        # dominant kernel FAMILY = the group of library launches with the largest total HIP-event time in the profiled
        # steps: GEMM forward / data-gradient kernels, the weight-gradient kernels, BatchNorm and the rest all compete
        fams = family_table(prof, max(prof_steps, 1))
        ps = max(prof_steps, 1)
        if fams:
            top, tf = fams[0]
        else:
            top, tf = {"family": "none", "ms_per_step": 0.0}, dict(ms=0.0, n=1, flops=0.0, bytes=0.0, ms_b=0.0)
        label = top["family"]
        nl = max(tf["n"], 1)
        ms = tf["ms"]
        mfma_bound = tf["flops"] > 0
        if mfma_bound:
            achieved, peak, runit = tf["flops"] / (ms * 1e-3) / 1e12 if ms > 0 else 0.0, MFMA_PEAK_TFLOPS, "TFLOP/s"
        else:
            achieved = tf["bytes"] / (tf["ms_b"] * 1e-3) / 1e9 if tf["ms_b"] > 0 else 0.0
            peak, runit = HBM_PEAK_GBS, "GB/s"
        # the committed PMC passes are of the Stage-I workload: no traffic figure for the other workloads
        traffic, traffic_src = None, None
        # live counter passes only for the default configuration (the child runs profile THAT workload)
        if a.workload == "stage1" and world == 1 and not a.no_pmc and not a.batch:
            traffic = live_pmc_traffic(label)
            traffic_src = "live" if traffic else None
        if traffic is None and a.workload == "stage1":
            traffic, traffic_src = pmc_traffic(label)
        alg_bytes = int(tf["bytes"] / nl) if tf["bytes"] > 0 else None
        lib_ms = sum(f["ms"] for _, f in fams) / ps
        finite = all(np.isfinite(v) for v in logs.values() if isinstance(v, float))
        out = {
            "metric": metric, "value": round(value, 1), "unit": unit,
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": desc, "name": a.workload, "batch_per_gpu": B, "global_batch": B * world,
                       "synthetic_batches": NBATCH, "gate_skip": False if has_gate else None,
                       "parallelism": f"dp{world}" + ("" if world == 1 else ("-syncbn" if a.sync_bn else "-localbn"))},
            "roofline": {"bound": "mfma" if mfma_bound else "hbm", "achieved": round(achieved, 1), "peak": peak,
                         "unit": runit, "frac": round(achieved / peak, 4), "traffic": traffic,
                         "algorithmic_bytes": alg_bytes,
                         "traffic_over_algorithmic": round(traffic / alg_bytes, 2) if traffic and alg_bytes else None,
                         "traffic_source": ("live" if traffic_src == "live" else f"committed:{traffic_src}") if traffic else None,
                         "traffic_unit": ("bytes per launch leaving L2, (2*FETCH_SIZE + WRITE_SIZE)*1024, "
                                          + ("measured on THIS box by two rocprofv3 --pmc child passes of this script "
                                             "(FETCH_SIZE, WRITE_SIZE: separate runs, --kernel-trace only, 5 one-stream steps)"
                                             if traffic_src == "live" else
                                             f"of the rocprofv3 PMC passes of this workload committed as profiles/{traffic_src}")
                                          + "; algorithmic_bytes = input + output + weights once, per launch") if traffic else None,
                         "kernel": label,
                         "measured": "HIP events around each library launch, 5 steps issued on ONE stream (the timed "
                                     "region overlaps weight gradients on a second stream when launched eagerly); the "
                                     "family with the largest total time is reported",
                         "launches_per_step": nl // ps,
                         "avg_launch_ms": round(ms / nl, 4),
                         "avg_launch_gflop": round(tf["flops"] / nl / 1e9, 2) if mfma_bound else None},
            "families": {"ms_per_step_one_stream": round(prof_ms_per_step, 3),
                         "ms_per_step_library_launches": round(lib_ms, 3),
                         "ms_per_step_torch_ops_and_gaps": round(prof_ms_per_step - lib_ms, 3),
                         "groups": [row for row, _ in family_table(prof, ps, _group_of)],
                         "rows": [row for row, _ in fams if row["ms_per_step"] >= 0.02]},
            "launch": {"graph": "hip-graph replay" if single else "hip-graph segments + eager collectives",
                       "hybrid": "forward replayed from a HIP graph, backward eager with weight gradients on a side "
                                 "stream" + ("" if single else " and eager collectives"),
                       "eager": "eager, one stream" if not ops._SIDE["on"]
                       else "eager, weight gradients on a side stream"}[mode],
            "step_mfma_frac": round(value / world * flop_per_sample / 1e12 / MFMA_PEAK_TFLOPS, 4),
            "gflop_per_sample": round(flop_per_sample / 1e9, 2),
            "losses_last_step": {k: v for k, v in logs.items() if isinstance(v, float)},
            "losses_finite": bool(finite),
        }
        if gate_out is not None:
            out["gate_skip"] = gate_out
        if world == 1 and not a.no_hbm_rows:
            out["hbm_bound_kernels"] = dict(peak_gb_s=HBM_PEAK_GBS, rows=hbm_rows(dev, 256))
        if world == 1 and not a.no_cpu_baseline and a.workload == "stage1":
            out["cpu_baseline"] = cpu_baseline(32, 10)        # ~10-15 s of CPU work (bounded to 25 s inside)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
