"""GPU box: eval-mode forward of the drop-in VaeGan (encoder -> decoder, models/vae_gan.py:288-297 under model.eval()) at
B = 256, with the eval BatchNorm folded into the convolutions' epilogues (default) and without (FMRI_EPI_AFFINE=off)."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "thesis-fmri-reconstruction_amd"))
import numpy as np, torch
import configs.models_config as mc
mc.use_px64()
import models.vae_gan as vg
dev = "cuda:0"; B = 256
rs = np.random.RandomState(0)
x = torch.from_numpy(rs.uniform(-1, 1, (B, 3, 64, 64)).astype(np.float32)).to(dev)
model = vg.VaeGan(device=dev, z_size=128).to(dev)
model.eval()
with torch.no_grad():
    for _ in range(10): y = model(x)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 50
    for _ in range(n): y = model(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
out = y[0] if isinstance(y, (tuple, list)) else y
print(f"eval forward B={B}: {1e3 * dt:.3f} ms  {B / dt:.0f} images/s  out norm {float(out.float().norm()):.4f}  FMRI_EPI_AFFINE={os.environ.get('FMRI_EPI_AFFINE', 'on')}")
