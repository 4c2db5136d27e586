"""Probe (GPU box): cProfile of the scripts' literal Stage-I loop body on the drop-in modules (host-bound path)."""
import cProfile, pstats, io, os, sys, runpy
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "thesis-fmri-reconstruction_amd"))
import numpy as np, torch
import configs.models_config as mc
mc.use_px64()
import models.vae_gan as vg
dev = "cuda:0"; B = 256
rs = np.random.RandomState(0)
x = torch.from_numpy(rs.uniform(-1, 1, (B, 3, 64, 64)).astype(np.float32)).to(dev)
model = vg.VaeGan(device=dev, z_size=128).to(dev); model.train()
mk = lambda p: torch.optim.RMSprop(params=p, lr=1e-4, alpha=0.9, eps=1e-8, weight_decay=0, momentum=0, centered=False)
oe, od, os_ = mk(model.encoder.parameters()), mk(model.decoder.parameters()), mk(model.discriminator.parameters())
lam = 1e-6
def step():
    x_tilde, disc_class, disc_layer, mus, lv = model(x)
    nle, kld, mse, bo, bp, bs = vg.VaeGan.loss(x, x_tilde, disc_layer[:B], disc_layer[B:-B], disc_layer[-B:],
                                               disc_class[:B], disc_class[B:-B], disc_class[-B:], mus, lv)
    le = torch.sum(kld) + torch.sum(mse)
    ld = torch.sum(bo) + torch.sum(bp) + torch.sum(bs)
    lg = torch.sum(lam * mse) - (1.0 - lam) * ld
    model.zero_grad(); le.backward(retain_graph=True); oe.step()
    model.zero_grad(); lg.backward(retain_graph=True); od.step()
    model.discriminator.zero_grad(); ld.backward(); os_.step()
    return le
torch.autograd.set_multithreading_enabled(False)      # backward on this thread: cProfile sees the bridge's Python
for _ in range(15): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
torch.cuda.synchronize(); pr.disable()
for key in ("tottime", "cumtime"):
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(key).print_stats(35); print(s.getvalue()[:6000])
