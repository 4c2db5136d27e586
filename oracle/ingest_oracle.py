"""TEST INFRASTRUCTURE ONLY -- numpy/scipy restatement of the tail of the reference's image transforms.

Per image (uint8 HWC, already cropped and resized), in the order of the training scripts:
  * ``transforms.RandomHorizontalFlip``  (train/train_vgan_stage1.py:166; torchvision 0.5.0: reverse the width axis)
  * ``RandomShift`` = ``scipy.ndimage.shift(img, [x_shift, y_shift, 0], prefilter=False, order=0, mode='nearest')``
    (data_preprocessing/data_loader.py:186-217; scipy is the third-party dependency that defines the arithmetic)
  * ``transforms.ToTensor``              (uint8 HWC -> float32 CHW / 255)
  * ``GreyToColor``                      (data_loader.py:374-401: one channel repeated three times)
  * ``transforms.Normalize(mean, std)``  (train_vgan_stage1.py:169: (x - mean) / std per channel)
The random draws are inputs.  Pinned by tests/golden/ingest.npz.  Only tests/ may import this module.
"""
import numpy as np
from scipy.ndimage import shift as nd_shift


def ingest(images_u8: np.ndarray, flip=None, shifts=None, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)) -> np.ndarray:
    """images_u8 [N,H,W,C] uint8 (C = 1 or 3) -> float32 [N,3,H,W]."""
    out = []
    for n, img in enumerate(images_u8):
        if flip is not None and flip[n]:
            img = img[:, ::-1, :]
        if shifts is not None:
            img = nd_shift(img, [int(shifts[n][0]), int(shifts[n][1]), 0], prefilter=False, order=0, mode="nearest")
        t = np.ascontiguousarray(img.transpose(2, 0, 1)).astype(np.float32) / np.float32(255.0)
        if t.shape[0] == 1:
            t = np.repeat(t, 3, axis=0)
        m = np.asarray(mean, np.float32).reshape(3, 1, 1)
        s = np.asarray(std, np.float32).reshape(3, 1, 1)
        out.append((t - m) / s)
    return np.stack(out).astype(np.float32)
