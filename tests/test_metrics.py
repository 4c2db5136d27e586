"""Validation metrics: the oracle restatement against goldens produced by the reference's own
PearsonCorrelation / StructuralSimilarity modules (CPU), and the HIP kernels against the oracle and the goldens (GPU)."""
import os

import numpy as np
import pytest
import torch

from oracle import metrics_oracle as MO


def _inputs(shape):
    n, c, h, w, seed = [int(v) for v in shape]
    rs = np.random.RandomState(seed)
    a = torch.from_numpy(rs.uniform(-1, 1, (n, c, h, w)).astype(np.float32))
    b = 0.6 * a + 0.4 * torch.from_numpy(rs.uniform(-1, 1, (n, c, h, w)).astype(np.float32))
    return a, b


def _cases(golden_dir):
    g = np.load(os.path.join(golden_dir, "metrics.npz"))
    return g, [str(t) for t in g["meta/cases"]]


def test_metrics_oracle_matches_reference(golden_dir):
    g, tags = _cases(golden_dir)
    for tag in tags:
        a, b = _inputs(g[f"{tag}/shape"])
        s, c = MO.structural_similarity(a, b)
        assert MO.pearson_correlation(a, b).item() == pytest.approx(float(g[f"{tag}/pcc"]), rel=1e-6)
        assert s.item() == pytest.approx(float(g[f"{tag}/ssim"]), rel=1e-6)
        assert c.item() == pytest.approx(float(g[f"{tag}/contrast"]), rel=1e-6)
        assert float(g[f"{tag}/ssim_default"]) == pytest.approx(float(g[f"{tag}/ssim"]), rel=1e-7)


@pytest.mark.gpu
def test_metrics_hip_matches_oracle_and_reference(golden_dir):
    from train.train_utils import PearsonCorrelation, StructuralSimilarity
    pcc, ssim = PearsonCorrelation(), StructuralSimilarity()
    g, tags = _cases(golden_dir)
    for tag in tags:
        a, b = _inputs(g[f"{tag}/shape"])
        ad, bd = a.cuda(), b.cuda()
        s, c = ssim(ad, bd, full=True)
        so, co = MO.structural_similarity(a, b)
        # fp32 sums in a different order (separable window, fp64 global accumulation): 1e-5 relative
        assert pcc(ad, bd).item() == pytest.approx(float(g[f"{tag}/pcc"]), rel=1e-5)
        assert s.item() == pytest.approx(float(g[f"{tag}/ssim"]), rel=1e-5)
        assert c.item() == pytest.approx(float(g[f"{tag}/contrast"]), rel=1e-5)
        assert s.item() == pytest.approx(so.item(), rel=1e-5) and c.item() == pytest.approx(co.item(), rel=1e-5)
        assert ssim(ad, bd).item() == pytest.approx(s.item(), rel=1e-7)
    # identical images: SSIM = 1, PCC = 1; a 3-D input is one image
    x = torch.rand(3, 64, 64, device="cuda")
    assert ssim(x, x).item() == pytest.approx(1.0, abs=1e-6)
    assert pcc(x, x).item() == pytest.approx(1.0, abs=1e-6)
    with pytest.raises(RuntimeError):
        pcc(x.cpu(), x.cpu())
