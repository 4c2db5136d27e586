"""Head of the image pipeline: CenterCrop + Resize on the device, pinned double-buffered staging.

CPU: the oracle's restatement of torchvision 0.5.0 center_crop / resize over Pillow's 8-bit resampling is BIT-EXACT on
the PIL-generated golden (tests/golden/resize.npz), and the library's host-side coefficient function equals the oracle's
tables.  GPU: the HIP kernel (through the C ABI) is bit-exact against oracle and golden; the staging ring overlaps copies
with compute and returns the same tensors as the direct calls."""
import os

import numpy as np
import pytest
import torch

from oracle import ingest_oracle as IO
from oracle import resize_oracle as R


def test_oracle_matches_pil_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "resize.npz"))
    assert [tuple(c) for c in g["meta/cases"]] == R.RESIZE_CASES
    for idx, (img, (h, w, c, crop, size)) in enumerate(zip(R.resize_inputs(), R.RESIZE_CASES)):
        got = R.resize_bilinear_u8(R.center_crop_u8(img, crop, crop), size, size)
        assert np.array_equal(got, g[f"out/{idx}"]), (idx, h, w, c, crop, size)


def test_center_crop_rounding_is_half_to_even():
    # torchvision 0.5.0: int(round((h - th) / 2.)) -- Python 3 round
    assert [R.center_crop_box(h, 10, 4, 4)[0] for h in (5, 7, 9, 11, 3, 1)] == [0, 2, 2, 4, 0, -2]


def test_library_coefficients_equal_the_oracle():
    from fmri_hip import build, lib as L
    build.build(verbose=False)
    lib = L.load()
    for crop, size in ((375, 64), (375, 100), (375, 128), (374, 100), (101, 128), (500, 17), (64, 64), (3, 7)):
        ks, b, k = R.resample_coeffs(crop, size)
        cap = ks + 2
        bb = np.zeros((size, 2), np.int32)
        kk = np.full((size, cap), -7, np.int32)
        r = lib.fmri_resize_coeffs(crop, size, bb.ctypes.data, kk.ctypes.data, cap)
        if crop == size:
            assert r == 0
            continue
        assert r == ks, (crop, size, r, ks)
        assert np.array_equal(bb, b)
        assert np.array_equal(kk.reshape(-1)[:size * ks].reshape(size, ks), k), (crop, size)
    # a table that does not fit its capacity is an error, not an overflow
    assert lib.fmri_resize_coeffs(375, 64, bb.ctypes.data, kk.ctypes.data, 3) < 0


def _pack(images, dev):
    offs, dims, chunks, pos = [], [], [], 0
    for a in images:
        offs.append(pos)
        dims.append(a.shape)
        chunks.append(a.reshape(-1))
        pad = (-a.size) % 16
        chunks.append(np.zeros(pad, np.uint8))
        pos += a.size + pad
    pool = torch.from_numpy(np.concatenate(chunks)).to(dev)
    return pool, torch.tensor(offs, dtype=torch.int64, device=dev), torch.tensor(dims, dtype=torch.int32, device=dev)


@pytest.mark.gpu
def test_crop_resize_kernel_is_bit_exact(golden_dir):
    from fmri_hip import ops
    dev = "cuda:0"
    g = np.load(os.path.join(golden_dir, "resize.npz"))
    imgs = R.resize_inputs()
    seen = 0
    for crop, size in sorted({(c[3], c[4]) for c in R.RESIZE_CASES}):
        idxs = [i for i, c in enumerate(R.RESIZE_CASES) if (c[3], c[4]) == (crop, size)]
        pool, offs, dims = _pack([imgs[i] for i in idxs], dev)
        out = ops.crop_resize_u8(pool, offs, dims, crop, size).cpu().numpy()
        want = R.crop_resize([imgs[i] for i in idxs], crop, size)
        assert np.array_equal(out, want), (crop, size)
        for j, i in enumerate(idxs):
            ref = g[f"out/{i}"]
            ref = np.repeat(ref, 3, axis=2) if ref.shape[2] == 1 else ref
            assert np.array_equal(out[j], ref), (i, R.RESIZE_CASES[i])
            seen += 1
    assert seen == len(R.RESIZE_CASES)
    # a COCO-shaped batch at the scripts' defaults: random sizes, 10 % grey images
    rs = np.random.RandomState(7)
    batch = []
    for _ in range(48):
        h, w = int(rs.randint(300, 641)), int(rs.randint(300, 641))
        batch.append(rs.randint(0, 256, (h, w, 1 if rs.rand() < 0.1 else 3)).astype(np.uint8))
    pool, offs, dims = _pack(batch, dev)
    out = ops.crop_resize_u8(pool, offs, dims, 375, 64).cpu().numpy()
    assert np.array_equal(out, R.crop_resize(batch, 375, 64))


@pytest.mark.gpu
def test_host_stager_overlaps_and_matches_direct_calls():
    from fmri_hip import ops
    dev = "cuda:0"
    rs = np.random.RandomState(11)
    batches = []
    for b in range(5):
        batches.append([rs.randint(0, 256, (int(rs.randint(380, 500)), int(rs.randint(380, 500)), 3)).astype(np.uint8)
                        for _ in range(16)])
    st = ops.HostStager(dev, crop=375, size=64, depth=2, capacity=1 << 25)
    # keep the main stream busy so that the staging of batch i + 1 runs beside it
    busy = torch.randn(4096, 4096, device=dev)
    results, ticket = [], st.submit(batches[0], want16=True, want32=True)
    for b in range(5):
        nxt = st.submit(batches[b + 1], want16=True, want32=True) if b + 1 < 5 else None    # staged on the side stream ...
        for _ in range(4):
            busy = busy @ busy * 1e-4                                                       # ... beside this
        x16, x32 = st.take(ticket)
        results.append((x16.float().cpu(), x32.cpu()))
        ticket = nxt
    torch.cuda.synchronize()
    assert st.stream != torch.cuda.current_stream() and all(e.query() for e in st.free_evt if e is not None)
    for b in range(5):
        u8 = R.crop_resize(batches[b], 375, 64)
        want = IO.ingest(u8)                                                                 # fp32 [N,3,64,64]
        got16, got32 = results[b]
        assert np.abs(got32.numpy() - want).max() < 1e-6
        assert np.abs(got16.numpy()[..., :3].transpose(0, 3, 1, 2) - want).max() < 1e-3      # fp16 storage
        assert float(got16[..., 3:].abs().max()) == 0.0
    # a batch that does not fit the staging buffers is refused, not truncated
    with pytest.raises(ValueError):
        ops.HostStager(dev, 375, 64, capacity=1 << 12).submit(batches[0])
