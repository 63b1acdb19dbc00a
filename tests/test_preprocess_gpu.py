"""-m gpu: md_image_preprocess (uint8 HWC -> warped, normalised bf16 in the network's input layout) against the numpy
restatement (oracle/np_ops.py::image_preprocess; cv2 is absent, parity unpinned).  Tolerance: one bf16 rounding of the
normalised value (rtol 2^-7) plus fp32 FMA-vs-separate accumulation noise."""
import numpy as np
import pytest
import torch

from oracle import np_ops
from tests.conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs MI355X")]
DEV = "cuda:0"
MEAN, STD = (0.408, 0.447, 0.470), (0.289, 0.274, 0.278)   # centernet/default_config.yaml mean / std


def _run(img, mat, out_hw, stem):
    from minddet_amd import nn_ops

    y = nn_ops.image_preprocess(torch.from_numpy(img).to(DEV), torch.from_numpy(mat).to(DEV), MEAN, STD, out_hw, stem_layout=stem)
    return y.float().cpu().numpy()


def test_identity_matrix_is_exact_normalisation():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (2, 32, 64, 3), dtype=np.uint8)
    mat = np.tile(np.array([1, 0, 0, 0, 1, 0], np.float32), (2, 1))
    y = _run(img, mat, (32, 64), stem=True)
    assert y.shape == (2, 32 + 16, 64 + 16, 4)
    ref = ((img.astype(np.float32) * np.float32(1 / 255.0) - np.array(MEAN, np.float32)) / np.array(STD, np.float32))
    ref_bf = torch.from_numpy(ref).to(torch.bfloat16).float().numpy()
    np.testing.assert_array_equal(y[:, 7:7 + 32, 7:7 + 64, :3], ref_bf)
    assert (y[..., 3] == 0).all() and (y[:, :7] == 0).all() and (y[:, 7 + 32:] == 0).all() and (y[:, :, :7] == 0).all() and (y[:, :, 7 + 64:] == 0).all()


@pytest.mark.parametrize("stem", [True, False])
def test_affine_warp_vs_oracle(stem):
    from minddet_amd import det_ops

    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (3, 120, 200, 3), dtype=np.uint8)
    out_hw = (64, 128)
    mats = []
    for b in range(3):
        c = np.array([100.0 + 7 * b, 60.0 - 3 * b], np.float32)
        s = 210.0 + 15 * b
        t = det_ops.get_affine_transform(c, s, (out_hw[1], out_hw[0]), inv=True)   # output pixel -> source pixel
        mats.append(np.asarray(t, np.float32).reshape(6))
    mat = np.stack(mats)
    y = _run(img, mat, out_hw, stem)
    ref = np_ops.image_preprocess(img, mat, MEAN, STD, out_hw)
    got = y[:, 7:7 + out_hw[0], 7:7 + out_hw[1], :3] if stem else y[..., :3]
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, rtol=2.0 ** -7, atol=2e-2)
    if not stem:
        assert y.shape == (3, 64, 128, 8) and (y[..., 3:] == 0).all()
    # the warp really samples outside the source for part of the output: those pixels are the normalised black level
    black = (0.0 - np.array(MEAN, np.float32)) / np.array(STD, np.float32)
    assert np.isclose(ref, black, atol=1e-6).all(-1).any()
