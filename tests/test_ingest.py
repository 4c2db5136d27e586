"""Batch ingest: oracle against the golden, HIP kernel against the oracle (bit-exact up to the fp16 store)."""
import os

import numpy as np
import pytest
import torch

from oracle import ingest_oracle as IO


def test_ingest_oracle_matches_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "ingest.npz"))
    for tag in ("rgb", "grey"):
        got = IO.ingest(g[f"{tag}/img"], g[f"{tag}/flip"], g[f"{tag}/shift"])
        np.testing.assert_array_equal(got, g[f"{tag}/out"])


@pytest.mark.gpu
def test_ingest_hip_matches_oracle(golden_dir):
    from fmri_hip.ops import ingest_u8
    g = np.load(os.path.join(golden_dir, "ingest.npz"))
    for tag in ("rgb", "grey"):
        img = torch.from_numpy(g[f"{tag}/img"]).cuda()
        flip, shift = torch.from_numpy(g[f"{tag}/flip"]), torch.from_numpy(g[f"{tag}/shift"])
        o16, o32 = ingest_u8(img, flip=flip, shift=shift, want16=True, want32=True)
        ref = g[f"{tag}/out"]
        np.testing.assert_allclose(o32.cpu().numpy(), ref, rtol=0, atol=2e-7)       # (v/255 - m) * (1/std) vs / std
        r16 = torch.from_numpy(ref).permute(0, 2, 3, 1).half()
        assert torch.equal(o16[..., :3].cpu(), r16) or (o16[..., :3].cpu().float() - r16.float()).abs().max() < 1e-3
        assert (o16[..., 3:] == 0).all()
    # larger batch without augmentation, ImageNet statistics
    rs = np.random.RandomState(3)
    big = rs.randint(0, 256, (64, 64, 64, 3)).astype(np.uint8)
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    _, o32 = ingest_u8(torch.from_numpy(big).cuda(), mean=mean, std=std, want16=False, want32=True)
    np.testing.assert_allclose(o32.cpu().numpy(), IO.ingest(big, mean=mean, std=std), rtol=0, atol=1e-6)
