"""TEST INFRASTRUCTURE ONLY -- torch-CPU fp32 restatement of the reference's validation metrics.

  * ``pearson_correlation``  = ``PearsonCorrelation.forward``     (train/train_utils.py:276-292)
  * ``structural_similarity`` = ``StructuralSimilarity.forward``  (train/train_utils.py:343-420, incl. ``gaussian``
    :313-326 and ``create_window`` :328-341)

Pinned by tests/golden/metrics.npz (tests/golden/make_golden.py imports the reference module and evaluates both
classes on seeded inputs).  Only tests/ may import this module.
"""
import math

import torch
import torch.nn.functional as F


def pearson_correlation(y_pred: torch.Tensor, y_true: torch.Tensor) -> torch.Tensor:
    vx = y_pred - torch.mean(y_pred)                                   # :286
    vy = y_true - torch.mean(y_true)                                   # :287
    return torch.sum(vx * vy) / (torch.sqrt(torch.sum(vx ** 2)) * torch.sqrt(torch.sum(vy ** 2)))   # :289


def gaussian_window(window_size: int, channels: int) -> torch.Tensor:
    g = torch.tensor([math.exp(-(x - window_size // 2) ** 2 / float(2 * 1.5 ** 2)) for x in range(window_size)])
    g = (g / g.sum()).unsqueeze(1)                                     # :324-326
    w2 = g.mm(g.t()).float().unsqueeze(0).unsqueeze(0)                 # :337
    return w2.expand(channels, 1, window_size, window_size).contiguous()


def structural_similarity(img1: torch.Tensor, img2: torch.Tensor, window_size: int = 11):
    """Returns (mean ssim, mean contrast term) -- the ``full=True`` pair of the reference."""
    pad = window_size // 2                                             # :375
    channels, height, width = img1.shape[-3:]
    real = min(window_size, height, width)                             # :384
    window = gaussian_window(real, channels)
    conv = lambda t: F.conv2d(t, window, padding=pad, groups=channels)
    mu1, mu2 = conv(img1), conv(img2)                                  # :389-390
    mu1_sq, mu2_sq, mu12 = mu1 ** 2, mu2 ** 2, mu1 * mu2
    s1 = conv(img1 * img1) - mu1_sq                                    # :398-400
    s2 = conv(img2 * img2) - mu2_sq
    s12 = conv(img1 * img2) - mu12
    C1, C2 = 0.01 ** 2, 0.03 ** 2                                      # :403-404
    contrast = torch.mean((2.0 * s12 + C2) / (s1 + s2 + C2))           # :406-407
    ssim = ((2 * mu12 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))   # :409-414
    return ssim.mean(), contrast
