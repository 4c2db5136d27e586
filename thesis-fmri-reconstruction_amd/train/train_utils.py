"""Evaluation metrics of the reference's validation loop on the MI355X engine.

Drop-in for the two metric modules of ``train/train_utils.py`` that the training scripts instantiate for
per-epoch validation (``PearsonCorrelation`` :267-292, ``StructuralSimilarity`` :295-420): same class names,
constructor and ``forward`` signatures, same values -- computed by HIP kernels (csrc/metrics.hip) on device
tensors without a host round trip.  Everything else of that module (plots, image dumps, the ``evaluate`` loop)
is outside the accelerated path and is FORWARDED: a name this module does not define (``evaluate``,
``objective_assessment``, ``EarlyStopping`` ... -- e.g. ``from train.train_utils import evaluate`` at
train/train_vgan_stage1.py:25, inference/inference_gan.py:24) is looked up in the ``train/train_utils.py`` this
one shadows, i.e. the next one on the ``train`` package's search path (train/__init__.py extends it over
``sys.path``).  That file is executed from where it lies on first use; nothing of it is copied here.
"""
import importlib.util
import os
import sys

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


_SHADOWED = None


def _shadowed_module():
    """The train/train_utils.py of the project this package overlays (loaded once, on first use)."""
    global _SHADOWED
    if _SHADOWED is None:
        import train as _pkg
        here = os.path.abspath(__file__)
        for d in list(_pkg.__path__):
            cand = os.path.join(d, "train_utils.py")
            if os.path.isfile(cand) and os.path.abspath(cand) != here:
                spec = importlib.util.spec_from_file_location("train._shadowed_train_utils", cand)
                mod = importlib.util.module_from_spec(spec)
                sys.modules[spec.name] = mod
                try:
                    spec.loader.exec_module(mod)
                except BaseException:
                    sys.modules.pop(spec.name, None)
                    raise
                _SHADOWED = mod
                break
        else:
            _SHADOWED = False
    return _SHADOWED


def __getattr__(name):
    if name.startswith("__"):
        raise AttributeError(name)
    mod = _shadowed_module()
    if mod and hasattr(mod, name):
        return getattr(mod, name)
    raise AttributeError(
        f"module 'train.train_utils' has no attribute {name!r}: the MI355X engine defines only PearsonCorrelation and "
        f"StructuralSimilarity, and no other train/train_utils.py follows it on sys.path" if not mod else
        f"module 'train.train_utils' has no attribute {name!r} (neither the engine's nor {mod.__file__})")
