#!/usr/bin/env python3
"""Headline benchmark: Stage-I VAE/GAN training step, 64x64x3 stimuli, latent 128, batch 256 per GPU
(BASELINE.json configs[1]), images/sec, on N MI355X of one node.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one full pass of the hot path over one synthetic batch already resident in HBM:
forward (encoder, 2x decoder, fused REC+GAN discriminator), losses, two-stream backward giving the three
gradient sets, SUM all-reduce over RCCL (N > 1), gated RMSprop updates, fp16 weight re-pack.
Prints ONE JSON line on rank 0.
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


def cpu_baseline(batch: int, steps: int):
    """Oracle (CPU restatement pinned to the reference) timed on this host's cores: the 'literal' variant
    (three full backward traversals, what train_vgan_stage1.py:410-432 does)."""
    from oracle import vaegan_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    cfg = O.ArchCfg.px64()
    P = O.fill_state(O.vaegan_spec(cfg), 0, False)
    data = O.synth_batch(batch, cfg, seed=1234, steps=1)
    opts = {n: O.OptState(kind="rmsprop", lr=1e-4) for n in ("encoder", "decoder", "discriminator")}
    args = (data["x"], data["noise"][0, 0], data["noise"][0, 1], cfg)
    log(f"cpu baseline: {cores} threads, batch {batch}")
    t0 = time.perf_counter()
    O.stage1_step(P, opts, *args, literal=True)            # warm-up
    warm = time.perf_counter() - t0
    log(f"cpu baseline warm-up step {warm:.1f}s")
    steps = max(1, min(steps, int(25.0 / max(warm, 1e-3))))   # bound the sample to ~25 s of CPU work
    t0 = time.perf_counter()
    for i in range(steps):
        O.stage1_step(P, opts, *args, literal=True)
        log(f"cpu baseline step {i + 1}/{steps}")
    dt = time.perf_counter() - t0
    return dict(value=round(batch * steps / dt, 3), unit="images/sec", cores=cores, kind="port",
                sample=f"{steps} literal Stage-I steps (3 full backward traversals) of the CPU oracle at batch {batch} "
                       f"(BASELINE configs[0]) after 1 warm-up, torch {torch.__version__} fp32, {cores} threads")


def pmc_traffic(label):
    """(2*FETCH_SIZE + WRITE_SIZE) * 1024 per launch of the kernel `label`, from the committed rocprofv3 PMC passes of
    this same command (tools/pmc_traffic.py); None when the file or the kernel is not there."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as f:
            kernels = json.load(f)["kernels"]
    except (OSError, ValueError, KeyError):
        return None
    want = label.replace(" ", "")
    for name, v in kernels.items():
        n = name.replace(" ", "")
        n = n[4:] if n.startswith("void") else n
        if n.split("(")[0] == want:
            return v["bytes_per_launch"]
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU (weak scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sync-bn", action="store_true",
                    help="data-parallel runs: all-reduce the BatchNorm partial sums (statistics of the GLOBAL batch, 32 "
                         "small collectives per step).  Default: per-rank statistics over the rank's own 256 images, "
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
        raise SystemExit(f"--gpus {a.gpus} needs torch.distributed.run with {a.gpus} ranks (WORLD_SIZE={world})")
    if os.environ.get("FMRI_REHEARSE_ON_ONE_GPU") == "1":
        local = 0                      # all ranks share device 0 (with FMRI_DIST_BACKEND=gloo): control-flow rehearsal
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    force_dist = os.environ.get("FMRI_FORCE_DIST") == "1"      # 1-rank rehearsal of the RCCL path
    if world > 1 or force_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("FMRI_DIST_BACKEND", "nccl")          # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    if a.serial:
        os.environ["FMRI_SIDE_STREAM"] = "off"
    from fmri_hip import lib, ops
    from fmri_hip.params import ArchConfig
    from fmri_hip.steps import Stage1Step
    lib.load()

    cfg = ArchConfig.px64()
    B = a.batch
    st = Stage1Step(cfg, dev, distributed=world > 1 or force_dist, sync_bn=a.sync_bn)
    st.load_recipe(0, False)
    x = torch.from_numpy(np.random.RandomState(1234 + rank).uniform(-1, 1, (B, 3, 64, 64)).astype(np.float32)).to(dev)
    nz = torch.from_numpy(np.random.RandomState(1236 + rank).standard_normal((2, B, cfg.latent_dim))
                          .astype(np.float32)).to(dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log(f"rank {rank}/{world}: model built, batch {B}; host cores {host_cores()} (cpu_count {os.cpu_count()})")
    for i in range(a.warmup):
        st.step(x, nz[0], nz[1])
        if i == 0:
            torch.cuda.synchronize()
            log("first step done")
    barrier()
    log("warm-up done")
    # The step (forward, losses, gate, two-stream backward, three optimizer updates, weight repack) is recorded once
    # into HIP graphs and the K timed steps are K replays -- the same launches, issued by the GPU front end instead of
    # ~370 Python/ctypes calls, so the number does not depend on the host CPU of the box.  In multi-process runs the
    # RCCL collectives stay eager calls between the graph segments.  --eager issues every launch from Python.
    eager_run = lambda: st.step(x, nz[0], nz[1])
    modes = {"eager": eager_run}
    single = world == 1 and not force_dist
    if not a.eager:
        try:
            modes["graph"] = st.capture(x, nz[0], nz[1])
            modes["graph"]()
            log("step captured into HIP graph(s)")
        except Exception as e:               # capture is an optimisation: fall back to eager launches
            log(f"graph capture failed ({type(e).__name__}: {e}); continuing with eager launches")
            modes.pop("graph", None)
        if (single or not a.sync_bn) and ops._SIDE["on"] and not a.graph:
            # hybrid: recorded forward + eagerly issued two-stream backward (half the Python work of a step)
            try:
                modes["hybrid"] = st.capture_forward(x, nz[0], nz[1])
                modes["hybrid"]()
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
        for _ in range(n):
            fn()
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
    use_graph = mode == "graph"
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        run()
    barrier()
    dt = time.perf_counter() - t0
    log(f"timed region done: {1e3 * dt / a.steps:.2f} ms/step ({mode})")
    # dominant-kernel timing: HIP events around every launch of that kernel over a few eagerly issued steps (events
    # cannot be placed inside a replayed graph)
    # The side stream is switched off for this pass: next to concurrently running weight-gradient kernels a launch's
    # duration says nothing about the kernel (igemm_win: 276 us alone, 345 us while sharing the CUs).
    ops.join_side()
    side_was, ops._SIDE["on"] = ops._SIDE["on"], False
    ops.PROFILE = [] if rank == 0 else None
    prof_steps = min(a.steps, 5)
    for _ in range(prof_steps):
        st.step(x, nz[0], nz[1])
    barrier()
    prof, ops.PROFILE = ops.PROFILE, None
    ops._SIDE["on"] = side_was
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    logs = st.logs()

    if rank == 0:
        value = world * B * a.steps / dt
        # dominant kernel = the fmri_igemm instantiation with the largest total HIP-event time in the profiled steps
        by = {}
        for label, e0, e1, f in prof:
            acc = by.setdefault(label, [0.0, 0.0, 0])
            acc[0] += e0.elapsed_time(e1)
            acc[1] += f
            acc[2] += 1
        label, (ms, fl, nl) = max(by.items(), key=lambda kv: kv[1][0]) if by else ("none", (0.0, 0.0, 0))
        nl = max(nl, 1)
        achieved = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        out = {
            "metric": "images/sec Stage-I VAE/GAN 64x64 bs256", "value": round(value, 1), "unit": "images/sec",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": "Stage-I VAE/GAN training step, 64x64x3 random images, latent 128, "
                                   "RMSprop x3, random-init weights (BASELINE configs[1])",
                       "batch_per_gpu": B, "global_batch": B * world,
                       "parallelism": f"dp{world}" + ("" if world == 1 else ("-syncbn" if a.sync_bn else "-localbn"))},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / MFMA_PEAK_TFLOPS, 4), "traffic": pmc_traffic(label),
                         "traffic_unit": "bytes per launch leaving L2 (PMC passes of this workload recorded in "
                                         "profiles/r01_pmc_traffic.json; hardware counters cannot be read in-process)",
                         "kernel": label,
                         "measured": "HIP events around each launch of the kernel, 5 steps issued on ONE stream (the "
                                     "timed region overlaps weight gradients on a second stream when launched eagerly)",
                         "launches_per_step": nl // max(prof_steps, 1),
                         "avg_launch_ms": round(ms / nl, 4),
                         "avg_launch_gflop": round(fl / nl / 1e9, 2)},
            "launch": {"graph": "hip-graph replay" if single else "hip-graph segments + eager collectives",
                       "hybrid": "forward replayed from a HIP graph, backward eager with weight gradients on a side "
                                 "stream" + ("" if single else " and eager collectives"),
                       "eager": "eager, one stream" if not ops._SIDE["on"]
                       else "eager, weight gradients on a side stream"}[mode],
            "step_mfma_frac": round(value / world * FLOP_PER_IMAGE / 1e12 / MFMA_PEAK_TFLOPS, 4),
            "losses_last_step": {k: logs[k] for k in ("loss_encoder", "loss_decoder", "loss_discriminator", "kl")},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(32, 10)        # ~10-15 s of CPU work (bounded to 25 s inside)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
