"""GPU box: API-path step -- wall, host issue time, cache-hit counters of the discriminator bridge."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "thesis-fmri-reconstruction_amd"))
import numpy as np, torch
import configs.models_config as mc
mc.use_px64()
import models.vae_gan as vg
hits = {"same_dir_calls": 0, "same_dir_ok": 0}
_orig = vg._same_direction
def counted(a, b):
    r, ok = _orig(a, b); hits["same_dir_calls"] += 1; hits["same_dir_ok"] += int(ok); return r, ok
vg._same_direction = counted
dev = "cuda:0"; B = 256
rs = np.random.RandomState(0)
x = torch.from_numpy(rs.uniform(-1, 1, (B, 3, 64, 64)).astype(np.float32)).to(dev)
model = vg.VaeGan(device=dev, z_size=128).to(dev); model.train()
mk = lambda p: torch.optim.RMSprop(params=p, lr=1e-4, alpha=0.9, eps=1e-8, weight_decay=0, momentum=0, centered=False)
oe, od, os_ = mk(model.encoder.parameters()), mk(model.decoder.parameters()), mk(model.discriminator.parameters())
lam = 1e-6
T = {}
def seg(name, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    a = T.setdefault(name, [0.0, 0.0]); a[0] += t1 - t0; a[1] += t2 - t0
    return r
def step(timed):
    S = seg if timed else (lambda n, f: f())
    out = S("forward", lambda: model(x))
    x_tilde, disc_class, disc_layer, mus, lv = out
    def losses():
        nle, kld, mse, bo, bp, bs = vg.VaeGan.loss(x, x_tilde, disc_layer[:B], disc_layer[B:-B], disc_layer[-B:],
                                                   disc_class[:B], disc_class[B:-B], disc_class[-B:], mus, lv)
        le = torch.sum(kld) + torch.sum(mse); ld = torch.sum(bo) + torch.sum(bp) + torch.sum(bs)
        return le, ld, torch.sum(lam * mse) - (1.0 - lam) * ld
    le, ld, lg = S("loss", losses)
    S("bwd_enc", lambda: (model.zero_grad(), le.backward(retain_graph=True)))
    S("opt_enc", lambda: oe.step())
    S("bwd_dec", lambda: (model.zero_grad(), lg.backward(retain_graph=True)))
    S("opt_dec", lambda: od.step())
    S("bwd_dis", lambda: (model.discriminator.zero_grad(), ld.backward()))
    S("opt_dis", lambda: os_.step())
for _ in range(3): step(False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): step(False)
torch.cuda.synchronize(); print(f"untimed: {1e2 * (time.perf_counter() - t0):.2f} ms/step", hits)
for _ in range(5): step(True)
for k, (h, w) in T.items():
    print(f"{k:10s} host {1e3 * h / 5:7.2f} ms   wall {1e3 * w / 5:7.2f} ms")

ts = []
for _ in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(False); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print("per-step wall (sync only between steps):", " ".join(f"{t:.1f}" for t in ts))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): step(False)
torch.cuda.synchronize(); print(f"untimed again: {1e2 * (time.perf_counter() - t0):.2f} ms/step")
print(torch.cuda.memory_stats()["num_alloc_retries"], torch.cuda.memory_stats()["num_device_alloc"], torch.cuda.memory_stats()["num_device_free"], f"{torch.cuda.max_memory_allocated()/1e9:.2f} GB max alloc, {torch.cuda.memory_reserved()/1e9:.2f} GB reserved")
import fmri_hip.ops as ops
ops._SIDE["on"] = False
for _ in range(3): step(False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): step(False)
torch.cuda.synchronize(); print(f"side stream off: {1e2 * (time.perf_counter() - t0):.2f} ms/step")
