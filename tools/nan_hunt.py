#!/usr/bin/env python3
"""Census + root-cause tool for non-finite Stage-I trajectories.

One process = one arm (the engine's switches are read from the environment at import):

    python tools/nan_hunt.py --steps 150 --repeats 10 --mode eager            # two-stream eager step (default arm)
    FMRI_SIDE_STREAM=off python tools/nan_hunt.py ...                        # one stream
    python tools/nan_hunt.py --mode hybrid ...                               # recorded forward + eager two-stream backward
    python tools/nan_hunt.py --det --repeats 2 ...                           # deterministic reductions: checksums must agree

Every repeat builds a fresh Stage1Step with bench.py's weights (recipe 0) and rotating synthetic batches and runs
``--steps`` steps with NO host synchronisation; the per-step loss block (and, with --diag, max |mu|, max logvar and the
cotangent normalisation factors) is kept on the device and read at the end.  Prints one JSON line per repeat and a summary.

--snap keeps a state snapshot per step; after a non-finite repeat the step that first went wrong is replayed from its
snapshot on ONE stream with a per-launch sentinel (every tensor handed to the library is checked after the launch) and the
first launch with a non-finite operand is named.  A replay that stays finite points at an ordering problem between the
streams, one that reproduces the non-finite value at the arithmetic.
"""
import argparse
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "thesis-fmri-reconstruction_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

NBATCH = 8


def build(dev, B, gate_skip, seed_shift=0):
    from fmri_hip.params import ArchConfig
    from fmri_hip.steps import Stage1Step
    cfg = ArchConfig.px64()
    st = Stage1Step(cfg, dev, gate_skip=gate_skip)
    st.load_recipe(0, False)
    mk = lambda s, shape, normal: torch.from_numpy(
        (np.random.RandomState(s).standard_normal(shape) if normal else np.random.RandomState(s).uniform(-1, 1, shape))
        .astype(np.float32)).to(dev)
    xs = [mk(1234 + 97 * i + seed_shift, (B, 3, 64, 64), False) for i in range(NBATCH)]
    nz = [mk(1236 + 97 * i + seed_shift, (2, B, cfg.latent_dim), True) for i in range(NBATCH)]
    return st, xs, nz


def groups_of(st):
    return (("encoder", st.enc, st.opt_enc), ("decoder", st.dec, st.opt_dec), ("discriminator", st.dis, st.opt_dis))


def snapshot(st):
    snap = {}
    for name, net, opt in groups_of(st):
        g = net.group
        for h in g.flush_hooks:
            h()
        snap[name] = dict(data=g.data.clone(), s1=opt.s1.clone(), bufs={k: v.clone() for k, v in g.bufs.items()})
    return snap


def restore(st, snap):
    for name, net, opt in groups_of(st):
        g = net.group
        s = snap[name]
        g.data.copy_(s["data"])
        opt.s1.copy_(s["s1"])
        for k, v in s["bufs"].items():
            g.bufs[k].copy_(v)
        g.version += 1
        g.buf_version += 1
        g.drop_pending()


def checksum(st):
    h = hashlib.sha256()
    for name, net, opt in groups_of(st):
        h.update(net.group.data.cpu().numpy().tobytes())
        h.update(opt.s1.cpu().numpy().tobytes())
    return h.hexdigest()[:16]


def finite_report(st):
    out = {}
    for name, net, opt in groups_of(st):
        g = net.group
        bad = []
        for k, v in g.views.items():
            if not torch.isfinite(v).all():
                bad.append(k)
        if not torch.isfinite(opt.s1).all():
            bad.append("<rmsprop state>")
        for pw in getattr(g, "packed", []):
            if not torch.isfinite(pw.buf.float()).all():
                bad.append("<fp16 copy rows=%d>" % pw.specs[0].rows)
        for k, v in g.bufs.items():
            if v.is_floating_point() and not torch.isfinite(v).all():
                bad.append(k)
        if bad:
            out[name] = bad
    return out


class Sentinel:
    """Per-launch non-finite check: every tensor whose pointer is handed to the library through ``_P`` is checked after
    the launch (host sync per launch: replay of a single step only)."""

    def __init__(self):
        from fmri_hip import lib, nets, ops, steps, wae_steps
        self.lib, self.mods = lib, (nets, ops, steps, wae_steps)
        self.args, self.seen, self.events, self.n = [], set(), [], 0

    def __enter__(self):
        lib = self.lib
        self._ptr, self._call = lib.ptr, lib.call
        me = self

        def ptr(t):
            if t is not None:
                me.args.append(t)
            return me._ptr(t)

        def call(name, *a):
            args, me.args = me.args, []
            me._call(name, *a)
            torch.cuda.synchronize()
            me.n += 1
            for i, t in enumerate(args):
                if not t.is_floating_point():
                    continue
                key = (t.data_ptr(), t.numel())
                if key in me.seen:
                    continue
                f = t.float()
                nbad = int((~torch.isfinite(f)).sum())
                if nbad:
                    me.seen.add(key)
                    fin = f[torch.isfinite(f)]
                    me.events.append(dict(launch=me.n, entry=name, arg=i, shape=list(t.shape), dtype=str(t.dtype),
                                          nonfinite=nbad, of=t.numel(),
                                          max_abs_finite=float(fin.abs().max()) if fin.numel() else None))
        lib.ptr, lib.call = ptr, call
        for m in self.mods:
            m._P = ptr
        return self

    def __exit__(self, *exc):
        self.lib.ptr, self.lib.call = self._ptr, self._call
        for m in self.mods:
            m._P = self._ptr
        return False


def run_other(a, dev, rep):
    """The other bench workloads (stage2, dual1, stage3_px128): bench.py's own builders, eager steps, finiteness of the
    loss block and of every parameter / optimizer-state buffer at the end."""
    import bench
    from fmri_hip import ops
    metric, unit, desc, wl_batch, builder = bench.WORKLOADS[a.workload]
    st, run_i, _, _ = builder(dev, a.batch or wl_batch, 0, False, False)
    n = 4 if a.workload == "stage3_px128" else 10
    rec = []
    for i in range(a.steps):
        run_i(i)
        rec.append(st.scal[:n].clone())
    ops.join_side()
    torch.cuda.synchronize()
    R = torch.stack(rec).cpu().numpy()
    fin = np.isfinite(R).all(1)
    bad_state = []
    for name in ("enc", "img_enc", "cog", "dec", "dis", "wd", "teacher_enc"):
        net = getattr(st, name, None)
        if net is not None and not torch.isfinite(net.group.data).all():
            bad_state.append(name)
    for name in ("opt_enc", "opt_dec", "opt_dis", "opt_wd"):
        o = getattr(st, name, None)
        if o is not None and not (torch.isfinite(o.s1).all() and (o.s2 is None or torch.isfinite(o.s2).all())):
            bad_state.append(name)
    return dict(rep=rep, workload=a.workload, mode="eager", side=ops._SIDE["on"], det=ops.deterministic(), steps=a.steps,
                finite=bool(fin.all()) and not bad_state, first_bad_step=None if fin.all() else int(np.argmin(fin)),
                nonfinite_state=bad_state, checksum="-", last=[float("%.5g" % v) for v in R[-1]])


def run_repeat(a, dev, rep):
    if a.workload != "stage1":
        return run_other(a, dev, rep)
    from fmri_hip import ops
    from fmri_hip.steps import (S_GDEC, S_NA, S_NB, S_NE)
    st, xs, nz = build(dev, a.batch, a.gate_skip)
    Z = st.cfg.latent_dim
    eager = lambda i: st.step(xs[i % NBATCH], nz[i % NBATCH][0], nz[i % NBATCH][1])
    run = eager
    n0 = 0
    if a.mode in ("hybrid", "graph"):
        for i in range(a.pre):                       # bench.py runs its warm-up eagerly before recording anything
            eager(i)
        n0 = a.pre
        sb = [xs[0].clone(), nz[0][0].clone(), nz[0][1].clone()]
        replay = st.capture_forward(*sb) if a.mode == "hybrid" else st.capture(*sb)

        def run(i):
            j = i % NBATCH
            sb[0].copy_(xs[j]); sb[1].copy_(nz[j][0]); sb[2].copy_(nz[j][1])
            return replay()
    rec, diag, snaps = [], [], []
    for i in range(n0, n0 + a.steps):
        if a.snap:
            snaps.append(snapshot(st))
        run(i)
        rec.append(st.scal[:22].clone())
        if a.diag:
            h = st.fw["head32"]
            diag.append(torch.stack([h[:, :Z].abs().max(), h[:, Z:].max(), h[:, Z:].min(), st.scal[S_NA], st.scal[S_NB],
                                     st.scal[S_NE], st.scal[S_GDEC], st.flags[0].float(), st.flags[1].float()]))
    ops.join_side()
    torch.cuda.synchronize()
    R = torch.stack(rec).cpu().numpy()
    fin = np.isfinite(R[:, :10]).all(1)
    first_bad = int(np.argmin(fin)) if not fin.all() else None
    out = dict(rep=rep, mode=a.mode, side=ops._SIDE["on"], det=ops.deterministic(), steps=a.steps,
               finite=bool(fin.all()), first_bad_step=first_bad, checksum=checksum(st),
               last=dict(kl=float(R[-1, 3]), mse=float(R[-1, 4]), nle=float(R[-1, 5]), bce_o=float(R[-1, 0]),
                         bce_p=float(R[-1, 1]), bce_s=float(R[-1, 2])))
    bad_state = finite_report(st)
    if bad_state:
        out["nonfinite_state"] = {k: v[:6] for k, v in bad_state.items()}
    if a.diag:
        D = torch.stack(diag).cpu().numpy()
        k = first_bad if first_bad is not None else len(D) - 1
        lo = max(0, k - 4)
        out["diag_cols"] = ["max|mu|", "max logvar", "min logvar", "nA", "nB", "nE", "gDec", "train_dis", "train_dec"]
        out["diag_tail"] = [[float("%.4g" % v) for v in row] for row in D[lo:k + 1]]
        out["max_logvar_over_run"] = float(np.nanmax(D[:, 1]))
        out["max_mu_over_run"] = float(np.nanmax(D[:, 0]))
        out["loss_tail"] = [[float("%.5g" % v) for v in row[:6]] for row in R[lo:k + 1]]
    if a.snap and first_bad is not None:
        out["replay"] = replay_from(a, st, xs, nz, snaps, first_bad, n0)
    return out


def replay_from(a, st, xs, nz, snaps, k, n0):
    """Replay steps k-1 and k (one stream, per-launch sentinel) from the snapshots taken before them."""
    from fmri_hip import ops
    res = {}
    side_was, ops._SIDE["on"] = ops._SIDE["on"], False
    try:
        for s in (k - 1, k):
            if s < 0:
                continue
            snap = snaps[s]
            pre_bad = [n for n in snap if not (torch.isfinite(snap[n]["data"]).all() and torch.isfinite(snap[n]["s1"]).all())]
            restore(st, snap)
            i = n0 + s
            with Sentinel() as sen:
                st.step(xs[i % NBATCH], nz[i % NBATCH][0], nz[i % NBATCH][1])
            torch.cuda.synchronize()
            losses = st.scal[:10].tolist()
            res[f"step_{s}"] = dict(state_before_nonfinite=pre_bad, losses_finite=bool(np.isfinite(losses).all()),
                                    launches=sen.n, first_events=sen.events[:12], state_after=finite_report(st))
    finally:
        ops._SIDE["on"] = side_was
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--repeats", type=int, default=10)
    ap.add_argument("--batch", type=int, default=0, help="0 = 256 (stage1) / the workload's own")
    ap.add_argument("--workload", choices=("stage1", "stage2", "dual1", "stage3_px128"), default="stage1")
    ap.add_argument("--pre", type=int, default=20, help="eager steps before a recording (hybrid / graph modes)")
    ap.add_argument("--mode", choices=("eager", "hybrid", "graph"), default="eager")
    ap.add_argument("--det", action="store_true")
    ap.add_argument("--diag", action="store_true")
    ap.add_argument("--snap", action="store_true")
    ap.add_argument("--gate-skip", action="store_true", help="engine default (bench headline runs with it off)")
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    if a.workload == "stage1" and not a.batch:
        a.batch = 256
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from fmri_hip import lib, ops
    lib.load()
    if a.det:
        ops.set_deterministic(True)
    nbad, sums = 0, []
    for r in range(a.repeats):
        out = run_repeat(a, dev, r)
        out["tag"] = a.tag
        nbad += 0 if out["finite"] else 1
        sums.append(out["checksum"])
        print(json.dumps(out), flush=True)
    print(json.dumps(dict(tag=a.tag, summary=True, mode=a.mode, side=ops._SIDE["on"], det=a.det, repeats=a.repeats,
                          steps=a.steps, nonfinite_runs=nbad, distinct_checksums=len(set(sums)))), flush=True)


if __name__ == "__main__":
    main()
