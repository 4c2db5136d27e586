"""Evaluation metrics of the reference's validation loop on the MI355X engine.

Drop-in for the two metric modules of ``train/train_utils.py`` that the training scripts instantiate for
per-epoch validation (``PearsonCorrelation`` :267-292, ``StructuralSimilarity`` :295-420): same class names,
constructor and ``forward`` signatures, same values -- computed by HIP kernels (csrc/metrics.hip) on device
tensors without a host round trip.  Everything else of that module (plots, image dumps, the ``evaluate`` loop)
is outside the accelerated path.
"""
import torch
from torch import nn

from fmri_hip import lib

_P = lib.ptr


def _dev32(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError("fmri_hip metrics run on the GPU only (got a CPU tensor); there is no CPU fallback")
    return t.detach().contiguous().float()


class PearsonCorrelation(nn.Module):
    """Pearson correlation coefficient over the whole batch (reference train/train_utils.py:267-292)."""

    def __init__(self):
        super(PearsonCorrelation, self).__init__()

    def forward(self, y_pred, y_true):
        a, b = _dev32(y_pred), _dev32(y_true)
        if a.shape != b.shape:
            raise ValueError("y_pred and y_true must have the same shape")
        ws = torch.empty(5, dtype=torch.float64, device=a.device)
        out = torch.empty((), dtype=torch.float32, device=a.device)
        lib.call("fmri_pcc", _P(a), _P(b), a.numel(), _P(ws), _P(out))
        return out


class StructuralSimilarity(nn.Module):
    """Mean local SSIM, 11x11 Gaussian window (reference train/train_utils.py:295-420)."""

    def __init__(self, mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225]):
        super(StructuralSimilarity, self).__init__()
        self.mean = mean
        self.std = std

    def forward(self, img1, img2, val_range=255, window_size=11, window=None, size_average=True, full=False):
        if window is not None or window_size != 11 or not size_average:
            raise NotImplementedError("the HIP kernel implements the reference defaults (11x11 window, size_average)")
        a, b = _dev32(img1), _dev32(img2)
        if a.shape != b.shape or a.dim() not in (3, 4):
            raise ValueError("img1 and img2 must be [N,C,H,W] or [C,H,W] tensors of the same shape")
        H, W = a.shape[-2], a.shape[-1]
        planes = a.numel() // (H * W)
        ws = torch.empty(2, dtype=torch.float64, device=a.device)
        out = torch.empty(2, dtype=torch.float32, device=a.device)
        lib.call("fmri_ssim", _P(a), _P(b), planes, H, W, _P(ws), _P(out[0:1]), _P(out[1:2]))
        if full:
            return out[0], out[1]
        return out[0]
