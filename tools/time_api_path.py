"""GPU box: ms/step of the reference scripts' LITERAL Stage-I loop body (train/train_vgan_stage1.py:330-432: model(x),
VaeGan.loss, three backward(retain_graph=True) passes, three torch.optim.RMSprop steps) on the drop-in modules
(thesis-fmri-reconstruction_amd/models/vae_gan.py over the HIP engine), beside the fused Stage1Step of the same batch."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "thesis-fmri-reconstruction_amd"))
import numpy as np, torch
import configs.models_config as mc
mc.use_px64()
import models.vae_gan as vg
dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rs = np.random.RandomState(0)
x = torch.from_numpy(rs.uniform(-1, 1, (B, 3, 64, 64)).astype(np.float32)).to(dev)
model = vg.VaeGan(device=dev, z_size=128).to(dev)
model.train()
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
for _ in range(15): step()          # the caching allocator keeps growing for ~10 steps (214 device allocations)
torch.cuda.synchronize(); t0 = time.perf_counter(); n = 20
for _ in range(n): l = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"API path (literal loop body, B={B}): {1e3 * dt:.2f} ms/step  {B / dt:.0f} images/s  loss_encoder {float(l.detach()):.1f}", flush=True)
if os.environ.get("API_ONLY") == "1":      # (under rocprofv3: the kernel table then holds the API path's launches only)
    sys.exit(0)
from fmri_hip.params import ArchConfig
from fmri_hip.steps import Stage1Step
st = Stage1Step(ArchConfig.px64(), dev); st.load_recipe(0, False)
e, z = (torch.from_numpy(rs.standard_normal((B, 128)).astype(np.float32)).to(dev) for _ in range(2))
for _ in range(3): st.step(x, e, z)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n): st.step(x, e, z)
torch.cuda.synchronize(); dt2 = (time.perf_counter() - t0) / n
print(f"fused Stage1Step (eager, B={B}):      {1e3 * dt2:.2f} ms/step  {B / dt2:.0f} images/s", flush=True)
