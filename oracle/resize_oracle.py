"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the head of the reference's image transforms.

The COCO pipeline of the training scripts (train/train_vgan_stage1.py:162-165) starts with

    transforms.CenterCrop((image_crop, image_crop)),  transforms.Resize((image_size, image_size))

on the decoded PIL image.  The arithmetic lives in two un-vendored dependencies, pinned in environment.yml:
torchvision==0.5.0 (:235) and pillow==8.0.1 (:173):

  * ``torchvision.transforms.functional.center_crop`` (0.5.0): ``i = int(round((h - th) / 2.))``,
    ``j = int(round((w - tw) / 2.))`` (Python 3 ``round``: half to even), then ``img.crop((j, i, j + tw, i + th))`` --
    PIL fills what lies outside the image with zeros;
  * ``torchvision.transforms.functional.resize`` with a (h, w) size: ``img.resize((w, h), Image.BILINEAR)``;
  * Pillow's ``ImagingResample`` for 8-bit images (src/libImaging/Resample.c): separable triangle filter whose support is
    the scale factor when shrinking (antialiasing), coefficients normalised in double precision and quantised to
    22-bit fixed point, horizontal pass first with the intermediate image ROUNDED TO uint8, then the vertical pass; a
    pass whose input and output size agree is skipped.

Pinned by tests/golden/resize.npz, generated with the Pillow of this container (12.2.0; the 8-bit resampling
arithmetic is unchanged since 8.0.1) by tests/golden/make_golden.py::case_resize.  Only tests/ may import this module.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def center_crop_box(h: int, w: int, th: int, tw: int):
    """(top, left) of torchvision 0.5.0's center_crop."""
    return int(round((h - th) / 2.0)), int(round((w - tw) / 2.0))


def resample_coeffs(in_size: int, out_size: int):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc for the bilinear filter (support 1.0) over the whole input.
    Returns (ksize, bounds int32 [out][2] = (first input index, count), coef int32 [out][ksize])."""
    scale = float(in_size) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    coef = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)            # C (int) cast: truncation toward zero
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = []
        ww = 0.0
        for x in range(xmax):
            a = (x + xmin - center + 0.5) * ss
            a = -a if a < 0.0 else a
            w = 1.0 - a if a < 1.0 else 0.0
            k.append(w)
            ww += w
        for x in range(xmax):
            v = k[x] / ww if ww != 0.0 else k[x]
            coef[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, coef


def _pass(img: np.ndarray, bounds: np.ndarray, coef: np.ndarray, axis: int) -> np.ndarray:
    """One 8-bit resampling pass along ``axis`` (0: rows = vertical, 1: columns = horizontal) of img [H][W][C] uint8."""
    src = img.astype(np.int64)
    n_out = bounds.shape[0]
    shape = list(img.shape)
    shape[axis] = n_out
    out = np.empty(shape, np.uint8)
    for o in range(n_out):
        lo, cnt = int(bounds[o, 0]), int(bounds[o, 1])
        k = coef[o, :cnt].astype(np.int64)
        if axis == 1:
            acc = (src[:, lo:lo + cnt, :] * k[None, :, None]).sum(1)
        else:
            acc = (src[lo:lo + cnt, :, :] * k[:, None, None]).sum(0)
        acc = acc + (1 << (PRECISION_BITS - 1))
        v = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)        # clip8
        if axis == 1:
            out[:, o, :] = v
        else:
            out[o, :, :] = v
    return out


def resize_bilinear_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """PIL ``Image.resize((out_w, out_h), BILINEAR)`` of an 8-bit image [H][W][C]."""
    h, w, _ = img.shape
    if w != out_w:
        _, b, k = resample_coeffs(w, out_w)
        img = _pass(img, b, k, axis=1)
    if h != out_h:
        _, b, k = resample_coeffs(h, out_h)
        img = _pass(img, b, k, axis=0)
    return img


def center_crop_u8(img: np.ndarray, th: int, tw: int) -> np.ndarray:
    h, w, c = img.shape
    i, j = center_crop_box(h, w, th, tw)
    out = np.zeros((th, tw, c), np.uint8)
    y0, y1 = max(i, 0), min(i + th, h)
    x0, x1 = max(j, 0), min(j + tw, w)
    if y1 > y0 and x1 > x0:
        out[y0 - i:y1 - i, x0 - j:x1 - j] = img[y0:y1, x0:x1]
    return out


def crop_resize(images, crop: int, size: int) -> np.ndarray:
    """List of uint8 images [H][W][C] (C = 1 or 3, any sizes) -> uint8 [N][size][size][3] (grey replicated, as
    GreyToColor does after ToTensor, data_loader.py:374-401)."""
    out = []
    for img in images:
        r = resize_bilinear_u8(center_crop_u8(img, crop, crop), size, size)
        if r.shape[2] == 1:
            r = np.repeat(r, 3, axis=2)
        out.append(r)
    return np.stack(out)


# ---- seeded test inputs shared by the golden generator (tests/golden/make_golden.py::case_resize) and the tests
RESIZE_CASES = [  # (H, W, C, crop, size): decoded-image shapes around the scripts' defaults (image_crop 375, sizes 64 / 100 / 128)
    (375, 500, 3, 375, 64), (500, 375, 3, 375, 64), (480, 640, 3, 375, 100), (427, 640, 1, 375, 64),
    (120, 90, 3, 375, 64), (375, 375, 3, 375, 128), (64, 64, 3, 64, 64), (333, 501, 3, 374, 100),
    (200, 300, 1, 101, 128), (77, 500, 3, 375, 64),
]


def resize_inputs():
    rs = np.random.RandomState(4321)
    imgs = []
    for h, w, c, _, _ in RESIZE_CASES:
        base = rs.randint(0, 256, (h // 7 + 2, w // 7 + 2, c)).astype(np.uint8)      # blocky + noise: edges and texture
        img = np.kron(base, np.ones((7, 7, 1), np.uint8))[:h, :w].astype(np.int32) + rs.randint(-20, 21, (h, w, c))
        imgs.append(np.clip(img, 0, 255).astype(np.uint8))
    return imgs


