"""-m gpu parity: HIP NMS / IoU kernels through the C ABI vs the CPU oracle and the golden
fixtures the reference produced.  Bit-exact for keep indices / masks; IoU values within a
stated fp tolerance (1e-6 abs for rotated polygons: float trig is evaluated in double on both
sides, the residue is ocml-vs-glibc last-bit differences in double)."""
import numpy as np
import pytest
import torch

import oracle
from oracle import np_ops
from tests.conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs MI355X")]

DEV = "cuda:0"


def rot_boxes(n, seed, span):
    rng = np.random.default_rng(seed)
    b = np.zeros((n, 7), np.float32)
    b[:, 0:2] = rng.uniform(-span, span, (n, 2))
    b[:, 2] = rng.uniform(-2, 2, n)
    b[:, 3:6] = rng.uniform(1, 5, (n, 3))
    b[:, 6] = rng.uniform(-np.pi, np.pi, n)
    return b


def aligned_boxes(n, seed, W=1344.0, H=800.0):
    rng = np.random.default_rng(seed)
    cx, cy = rng.uniform(0, W, n), rng.uniform(0, H, n)
    w = np.exp(rng.uniform(np.log(8), np.log(512), n))
    h = np.exp(rng.uniform(np.log(8), np.log(512), n))
    b = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], -1)
    b[:, 0::2] = np.clip(b[:, 0::2], 0, W)
    b[:, 1::2] = np.clip(b[:, 1::2], 0, H)
    return b.astype(np.float32)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_rot_nms_aot_golden(golden, tag):
    from minddet_amd import det_ops

    boxes, thr = golden[f"rotnms_{tag}_boxes"], float(golden[f"rotnms_{tag}_thr"])
    keep, num = det_ops.NMS()(T(boxes), thr)
    assert int(num) == int(golden[f"rotnms_{tag}_num"])
    np.testing.assert_array_equal(keep.cpu().numpy(), golden[f"rotnms_{tag}_keep"])


@pytest.mark.parametrize("n,span,thr", [(1, 5, 0.2), (63, 6, 0.1), (64, 6, 0.3), (65, 6, 0.3), (900, 15, 0.2),
                                          (1000, 30, 0.2), (4096, 40, 0.01), (2500, 20, 0.7)])
def test_rot_nms_both_conventions_vs_oracle(n, span, thr):
    from minddet_amd import det_ops

    boxes = rot_boxes(n, 100 + n, span)
    k_o, n_o = oracle.nms_rot_aot(boxes, thr)
    keep, num = det_ops.NMS()(T(boxes), thr)
    assert int(num) == n_o
    np.testing.assert_array_equal(keep.cpu().numpy(), k_o)
    k_m, n_m = oracle.nms_rot_mask(boxes, thr)
    keep, num = det_ops.NumGpu()(T(boxes), torch.tensor([thr], device=DEV))
    assert int(num[0]) == n_m
    np.testing.assert_array_equal(keep.cpu().numpy(), k_m)


def test_rot_nms_empty_and_degenerate():
    from minddet_amd import det_ops

    keep, num = det_ops.NMS()(torch.zeros((0, 7), device=DEV), 0.2)
    assert int(num) == 0 and keep.numel() == 0
    # all-zero boxes (the reference's own smoke input, nms_cpu.py:30-32): every box has zero
    # area -> AOT convention drops all of them
    keep, num = det_ops.NMS()(torch.zeros((100, 7), device=DEV), 0.1)
    k_o, n_o = oracle.nms_rot_aot(np.zeros((100, 7), np.float32), 0.1)
    assert int(num) == n_o == 0
    # identical boxes: first survives
    b = np.tile(np.array([[1, 2, 0, 3, 2, 1, 0.3]], np.float32), (70, 1))
    keep, num = det_ops.NMS()(T(b), 0.5)
    k_o, n_o = oracle.nms_rot_aot(b, 0.5)
    assert int(num) == n_o == 1
    np.testing.assert_array_equal(keep.cpu().numpy(), k_o)


def test_nms_normal_gpu_vs_oracle():
    from minddet_amd import det_ops

    for n, span, thr in [(1000, 12, 0.3), (777, 6, 0.1), (130, 3, 0.7)]:
        boxes = rot_boxes(n, 7 + n, span)
        k_o, n_o = oracle.nms_normal_mask(boxes, thr)
        keep, num = det_ops.NmsNormalGpu()(T(boxes), thr)
        assert int(num[0]) == n_o
        np.testing.assert_array_equal(keep.cpu().numpy(), k_o)


def test_iou_bev_matrix_golden_and_oracle(golden):
    from minddet_amd import det_ops

    got = det_ops.boxes_iou_bev(T(golden["ioubev_a"]), T(golden["ioubev_b"])).cpu().numpy()
    np.testing.assert_allclose(got, golden["ioubev_out"], rtol=0, atol=2e-6)  # vs the reference itself
    np.testing.assert_allclose(got, oracle.boxes_iou_bev(golden["ioubev_a"], golden["ioubev_b"]), rtol=0, atol=1e-6)
    # reference harness shape 52640 x 6 (iou_gpu.py:107-108), checked on a strided sample of rows
    a, b = rot_boxes(52640, 5, 60), rot_boxes(6, 6, 60)
    b[:, :2] = a[:6, :2] + 0.3
    got = det_ops.boxes_iou_bev(T(a), T(b)).cpu().numpy()
    ov = det_ops.boxes_overlap_bev(T(a), T(b)).cpu().numpy()
    sel = np.concatenate([np.arange(0, 52640, 41), np.arange(6)])
    np.testing.assert_allclose(got[sel], oracle.boxes_iou_bev(a[sel], b), rtol=0, atol=1e-6)
    np.testing.assert_allclose(ov[sel], oracle.boxes_overlap_bev(a[sel], b), rtol=0, atol=2e-5)
    assert got.shape == (52640, 6) and (got >= 0).all() and (got <= 1 + 1e-5).all()


def test_iou3d_symmetry_property():
    from minddet_amd import det_ops

    a = rot_boxes(500, 21, 10)
    m = det_ops.boxes_iou_bev(T(a), T(a)).cpu().numpy()
    assert np.abs(np.diag(m) - 1).max() < 1e-4
    i3 = det_ops.boxes_iou3d_gpu(T(a), T(a)).cpu().numpy()
    assert np.abs(np.diag(i3) - 1).max() < 1e-4 and (i3 <= 1 + 1e-4).all()


@pytest.mark.parametrize("eps,key", [(0.0, "ioujit_out_eps0"), (1.0, "ioujit_out_eps1")])
def test_iou_jit_bit_exact(golden, eps, key):
    from minddet_amd import det_ops

    got = det_ops.iou_jit(T(golden["ioujit_boxes"]), T(golden["ioujit_query"]), eps).cpu().numpy()
    np.testing.assert_array_equal(got, golden[key])
    # larger: anchors-in-mask x gt scale
    b, q = aligned_boxes(107136, 1), aligned_boxes(50, 2)
    got = det_ops.iou_jit(T(b), T(q), 0.0).cpu().numpy()
    np.testing.assert_array_equal(got, np_ops.iou_jit(b, q, 0.0))


@pytest.mark.parametrize("thr", [0.01, 0.5, 0.7])
def test_nms_jit_golden(golden, thr):
    from minddet_amd import det_ops

    dets = golden["nmsjit_dets"]
    keep = det_ops.nms_jit(T(dets), thr, 0.0).cpu().numpy()
    np.testing.assert_array_equal(keep, golden[f"nmsjit_keep_{thr}"])


def test_nms_jit_eps_and_plus1_golden(golden):
    from minddet_amd import det_ops

    dets = golden["nmsjit_dets"]
    keep = det_ops.nms_jit(T(dets), 0.5, 1.0).cpu().numpy()
    np.testing.assert_array_equal(keep, golden["nmsjit_keep_0.5_eps1"])
    order = torch.sort(T(dets[:, 4]), descending=True, stable=True)[1]
    _, idx, num = det_ops.nms_aligned(T(dets)[order, :4].contiguous(), 0.5, mode=det_ops.NMS_MODE_PLUS1, max_output=100)
    got = order[idx[: int(num[0])].long()].cpu().numpy()
    np.testing.assert_array_equal(got, golden["applynms_keep"])


@pytest.mark.parametrize("n,mode,thr", [(900, 0, 0.01), (1000, 2, 0.7), (4096, 2, 0.5), (30000, 2, 0.45)])
def test_nms_aligned_sizes_vs_oracle(n, mode, thr):
    from minddet_amd import det_ops

    W, H = (1344.0, 800.0) if n < 20000 else (6400.0, 6400.0)
    b = aligned_boxes(n, n, W, H)
    m_o = oracle.nms_aligned(b, thr, 0.0, mode)
    mask, idx, num = det_ops.nms_aligned(T(b), thr, 0.0, mode)
    np.testing.assert_array_equal(mask.cpu().numpy(), m_o)
    assert int(num[0]) == int(m_o.sum())
    np.testing.assert_array_equal(idx.cpu().numpy()[: int(num[0])], np.nonzero(m_o)[0])
    assert (idx.cpu().numpy()[int(num[0]):] == 0).all()


def test_nms_aligned_batched_classwise():
    from minddet_amd import det_ops

    B, n = 5, 1000
    rng = np.random.default_rng(9)
    boxes = np.stack([aligned_boxes(n, 50 + i, 400, 300) for i in range(B)])
    group = rng.integers(0, 80, (B, n)).astype(np.int32)
    count = np.array([1000, 0, 1, 64, 513], np.int32)
    mask, idx, num = det_ops.nms_aligned(T(boxes), 0.5, mode=2, count=T(count), group=T(group))
    mask, idx, num = mask.cpu().numpy(), idx.cpu().numpy(), num.cpu().numpy()
    for i in range(B):
        c = count[i]
        m_o = oracle.nms_aligned(boxes[i, :c], 0.5, 0.0, 2, groups=group[i, :c])
        np.testing.assert_array_equal(mask[i, :c], m_o)
        assert (mask[i, c:] == 0).all() and num[i] == m_o.sum()
        np.testing.assert_array_equal(idx[i, : num[i]], np.nonzero(m_o)[0])
    # per-class loop equivalence (centernet/src/post_process.py:41-52 shape of the computation)
    c = int(count[0])
    per_class = np.zeros(c, np.uint8)
    for k in range(80):
        sel = np.nonzero(group[0, :c] == k)[0]
        per_class[sel] = oracle.nms_aligned(boxes[0, sel], 0.5, 0.0, 2)
    np.testing.assert_array_equal(mask[0, :c], per_class)


def test_nms_idempotent_property():
    from minddet_amd import det_ops

    b = aligned_boxes(8000, 77)
    mask, idx, num = det_ops.nms_aligned(T(b), 0.5, mode=2)
    kept = T(b)[idx[: int(num[0])].long()].contiguous()
    mask2, _, num2 = det_ops.nms_aligned(kept, 0.5, mode=2)
    assert int(num2[0]) == int(num[0]) and bool(mask2.all())


def test_circle_nms_vs_oracle():
    from minddet_amd import det_ops

    rng = np.random.default_rng(4)
    d = np.concatenate([rng.uniform(-50, 50, (1500, 2)), rng.uniform(0, 1, (1500, 1))], 1).astype(np.float32)
    d[:, 2] += np.arange(1500) * 1e-7
    order = np.argsort(-d[:, 2], kind="stable")
    m = oracle.circle_nms(d[order], 4.0).astype(bool)
    keep = det_ops.circle_nms(T(d), 4.0).cpu().numpy()
    np.testing.assert_array_equal(keep, order[m])


@pytest.mark.parametrize("criterion", [-1, 0, 1, 2])
def test_rotate_iou_eval_vs_oracle(criterion):
    from minddet_amd import det_ops

    rng = np.random.default_rng(31)
    n, k = 700, 45
    b = np.concatenate([rng.uniform(-10, 10, (n, 2)), rng.uniform(1, 5, (n, 2)), rng.uniform(-np.pi, np.pi, (n, 1))], 1).astype(np.float32)
    q = np.concatenate([rng.uniform(-10, 10, (k, 2)), rng.uniform(1, 5, (k, 2)), rng.uniform(-np.pi, np.pi, (k, 1))], 1).astype(np.float32)
    got = det_ops.rotate_iou_gpu_eval(T(b), T(q), criterion).cpu().numpy()
    ref = oracle.rotate_iou_eval(b, q, criterion)
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-5)   # cosf/sinf of ocml vs glibc, then polygon area
    if criterion == 2:
        # independent cross-check against the polygon-clipping overlap of the other reference formulation
        b7 = np.zeros((n, 7), np.float32); b7[:, [0, 1, 3, 4]] = b[:, :4]; b7[:, 6] = -b[:, 4]
        q7 = np.zeros((k, 7), np.float32); q7[:, [0, 1, 3, 4]] = q[:, :4]; q7[:, 6] = -q[:, 4]
        ov = oracle.boxes_overlap_bev(b7, q7)
        assert np.abs(got - ov).max() < 5e-2   # MARGIN 1e-2 of check_in_box2d only affects grazing contacts


@pytest.mark.parametrize("cfg", [(9, 1000, 100, 0.5, 3), (32, 4096, 300, 0.45, 80), (8, 257, 64, 0.7, 1), (12, 640, 20, 0.3, 5)])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_quota_nms_batched_equals_list_by_list(cfg, mode):
    """md_nms_aligned with an output quota, ragged counts and class-keyed groups: a batch of lists gives, list by list, the keep
    masks, kept indices and counts of single-list calls.  (Written for a matrix-free quota kernel that was measured and dropped --
    DESIGN.md section 9 -- and kept as a consistency test of the batched path.)  Clustered boxes so that suppression happens."""
    from minddet_amd import det_ops

    L, n, quota, thr, ngroups = cfg
    rng = np.random.default_rng(L * n + mode)
    centers = rng.uniform(50, 600, (L, 40, 2)).astype(np.float32)
    pick = rng.integers(0, 40, (L, n))
    c = np.take_along_axis(centers, np.repeat(pick[..., None], 2, 2), 1) + rng.normal(0, 6, (L, n, 2)).astype(np.float32)
    wh = rng.uniform(20, 80, (L, n, 2)).astype(np.float32)
    boxes = np.concatenate([c - wh / 2, c + wh / 2], -1).astype(np.float32)
    count = rng.integers(n // 2, n + 1, L).astype(np.int32)
    count[0] = n
    group = rng.integers(0, ngroups, (L, n)).astype(np.int32)
    bt, ct, gt = torch.from_numpy(boxes).to(DEV), torch.from_numpy(count).to(DEV), torch.from_numpy(group).to(DEV)
    keep, kidx, num = det_ops.nms_aligned(bt, thr, mode=mode, count=ct, group=gt, max_output=quota)
    assert int(num.max()) <= quota and int(num.min()) > 0
    for l in range(L):
        k1, i1, n1 = det_ops.nms_aligned(bt[l:l + 1], thr, mode=mode, count=ct[l:l + 1], group=gt[l:l + 1], max_output=quota)
        assert int(n1[0]) == int(num[l])
        assert torch.equal(k1[0], keep[l]) and torch.equal(i1[0, : int(n1[0])], kidx[l, : int(num[l])])


@pytest.mark.parametrize("n,quota,ngroups", [(4096, 300, 80), (2048, 100, 80), (4096, 300, 1)])
def test_quota_prefix_pass_equals_the_full_pass_and_the_oracle(n, quota, ngroups):
    """md_nms_aligned's quota prefix pass (mask corner + scan over the first P boxes, full pass gated on the device for the lists that
    did not fill the quota): per list identical to the oracle's greedy NMS cut at `quota` kept boxes -- lists that fill the quota
    inside the prefix, lists that need the full pass (a long run of near-duplicates first) and short lists, in ONE batch; and identical
    with a caller workspace that only holds the mask (single full pass)."""
    from minddet_amd import det_ops

    L = 6
    rng = np.random.default_rng(n + quota + ngroups)
    c = rng.uniform(0, 2000, (L, n, 2)).astype(np.float32)
    wh = rng.uniform(10, 60, (L, n, 2)).astype(np.float32)
    # lists 1 and 4: the first 3000 / 1800 boxes are near-copies of a few boxes per class -> the prefix collapses, the full pass must run
    for l, m in ((1, min(3000, n - 200)), (4, min(1800, n - 200))):
        c[l, :m] = np.array([500.0, 500.0], np.float32) + rng.normal(0, 1.0, (m, 2)).astype(np.float32)
        wh[l, :m] = 50.0
    boxes = np.concatenate([c - wh / 2, c + wh / 2], -1).astype(np.float32)
    count = np.array([n, n, 700, n - 5, n, 2 * 64 + 3], np.int32)   # list 2 / 5: shorter than the prefix
    group = rng.integers(0, ngroups, (L, n)).astype(np.int32)
    bt, ct, gt = T(boxes), T(count), T(group)
    keep, kidx, num = det_ops.nms_aligned(bt, 0.5, mode=det_ops.NMS_MODE_STRICT, count=ct, group=gt, max_output=quota)
    ws = torch.empty((L * n * ((n + 63) // 64) * 8,), dtype=torch.uint8, device=DEV)
    keep2, kidx2, num2 = det_ops.nms_aligned(bt, 0.5, mode=det_ops.NMS_MODE_STRICT, count=ct, group=gt, max_output=quota, workspace=ws)
    assert torch.equal(keep, keep2) and torch.equal(kidx, kidx2) and torch.equal(num, num2)
    needed_full = 0
    for l in range(L):
        m = int(count[l])
        k_o = oracle.nms_aligned(boxes[l, :m], 0.5, 0.0, 2, groups=group[l, :m]).astype(bool)
        kept = np.nonzero(k_o)[0][:quota]
        assert int(num[l]) == len(kept)
        np.testing.assert_array_equal(kidx.cpu().numpy()[l, :len(kept)], kept)
        km = np.zeros(n, np.uint8); km[kept] = 1
        np.testing.assert_array_equal(keep.cpu().numpy()[l], km)
        P = (max(512, 4 * quota) + 63) // 64 * 64
        needed_full += int(m > P and k_o[:P].sum() < quota)
    assert needed_full >= 1   # the batch really exercises the gated full pass
