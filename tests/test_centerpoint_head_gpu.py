"""-m gpu: CenterPoint head predict + post_processing (center_head.py:273-463) vs the numpy/C oracle:
decode within fp tolerance (expf/atan2f), masks exact away from the threshold, TopK order / rotated-NMS keep
list / count bit-exact from the device tensors (the NMS is the reference operator's device twin)."""
import numpy as np
import pytest
import torch

import oracle
from oracle import np_ops
from tests.conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs MI355X")]
DEV = "cuda:0"
CFG = dict(post_center_limit_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], max_per_img=500,
           nms=dict(nms_pre_max_size=1000, nms_post_max_size=83, nms_iou_threshold=0.2), score_threshold=0.1,
           pc_range=[-51.2, -51.2], out_size_factor=4, voxel_size=[0.2, 0.2])
OFF = dict(reg=0, height=2, dim=3, rot=6, vel=8, hm=10)


def test_center_head_post_processing():
    from minddet_amd import det_ops

    rng = np.random.default_rng(0)
    B, H, W, ncls = 2, 128, 128, 2
    head = np.zeros((B, H, W, 16), np.float32)
    head[..., 0:2] = rng.uniform(0, 1, (B, H, W, 2))
    head[..., 2] = rng.normal(0, 1, (B, H, W))
    head[..., 3:6] = rng.normal(0.8, 0.3, (B, H, W, 3))
    head[..., 6:8] = rng.normal(0, 1, (B, H, W, 2))
    head[..., 8:10] = rng.normal(0, 1, (B, H, W, 2))
    head[..., 10:12] = rng.normal(-2.5, 1.5, (B, H, W, 2))
    hb = torch.from_numpy(head).to(torch.bfloat16)
    post = det_ops.CenterHeadPost(OFF, ncls, CFG)
    (boxes, scores, labels, count), aux = post(hb.to(DEV), return_aux=True)
    torch.cuda.synchronize()
    hf = hb.float().numpy()
    s_o, l_o, b_o, nb_o, mask_o = np_ops.centerpoint_decode(hf, OFF, ncls, CFG)
    s_d, l_d = aux["scores"].cpu().numpy(), aux["labels"].cpu().numpy()
    near = np.abs(np.where(mask_o, s_o, 1.0) - CFG["score_threshold"]) < 1e-5
    agree = (s_d > -1) == mask_o
    assert agree[~near].all()
    both = (s_d > -1) & mask_o
    np.testing.assert_allclose(s_d[both], s_o[both], rtol=2e-6, atol=1e-7)
    assert (l_d[both] == l_o[both]).mean() > 0.9999  # argmax can flip only on a 1-ulp tie
    np.testing.assert_allclose(aux["boxes"].cpu().numpy()[both], b_o[both], rtol=3e-6, atol=2e-5)
    np.testing.assert_allclose(aux["nms_boxes"].cpu().numpy()[both], nb_o[both], rtol=3e-6, atol=2e-5)
    assert (aux["boxes"].cpu().numpy()[~(s_d > -1)] == 0).all()
    # from the DEVICE scores / nms boxes onward: exact
    nbd = aux["nms_boxes"].cpu().numpy()
    for b in range(B):
        v, order = np_ops.topk_desc_stable(s_d[b], 1000)
        np.testing.assert_array_equal(aux["order"].cpu().numpy()[b], order)
        keep_o, num_o = oracle.nms_rot_aot(nbd[b][order], CFG["nms"]["nms_iou_threshold"])
        np.testing.assert_array_equal(aux["keep"].cpu().numpy()[b], keep_o)
        assert int(aux["num_out"][b]) == num_o
        mask_num = int((v > -1).sum())
        assert int(count[b]) == min(num_o, mask_num, 83)
        c = int(count[b])
        np.testing.assert_array_equal(scores.cpu().numpy()[b, :c], v[keep_o[:c]])
        np.testing.assert_array_equal(labels.cpu().numpy()[b, :c], l_d[b][order][keep_o[:c]])
        np.testing.assert_array_equal(boxes.cpu().numpy()[b, :c], aux["boxes"].cpu().numpy()[b][order][keep_o[:c]])
    assert int(count.sum()) > 0


def test_center_head_all_masked():
    from minddet_amd import det_ops

    head = torch.full((1, 16, 16, 16), -10.0).to(torch.bfloat16)
    (boxes, scores, labels, count) = det_ops.CenterHeadPost(OFF, 2, CFG)(head.to(DEV))
    # every cell masked: TopK returns -1 scores, zero boxes have zero area -> the operator drops all of them
    assert int(count[0]) == 0
