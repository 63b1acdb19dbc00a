"""-m gpu: edge cases through the C ABI -- empty inputs, ragged / degenerate data, size limits, wrong dtypes.
Errors surface as return codes (iou-bev-nms-org.cpp:238 convention), mapped to MindDetHipError by the host."""
import numpy as np
import pytest
import torch

from tests.conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs MI355X")]
DEV = "cuda:0"


def test_empty_inputs_are_ok():
    from minddet_amd import det_ops, nn_ops

    z7 = torch.zeros((0, 7), device=DEV)
    assert det_ops.boxes_iou_bev(z7, torch.zeros((5, 7), device=DEV)).shape == (0, 5)
    assert det_ops.boxes_iou_bev(torch.zeros((5, 7), device=DEV), z7).shape == (5, 0)
    assert det_ops.iou_jit(torch.zeros((0, 4), device=DEV), torch.zeros((3, 4), device=DEV)).shape == (0, 3)
    keep, num = det_ops.NumGpu()(z7, 0.5)
    assert int(num[0]) == 0
    m, i, n = det_ops.nms_aligned(torch.zeros((0, 4), device=DEV), 0.5, mode=2)
    assert int(n[0]) == 0
    v, i, c = det_ops.topk_segmented(torch.zeros((0,), device=DEV), torch.zeros((3,), dtype=torch.int32, device=DEV), 10)
    assert c.tolist() == [0, 0] and (i == 0).all()
    pc = nn_ops.pack_conv(torch.randn((64, 64, 3, 3)), stride=1, pad=1).to(DEV)
    y = nn_ops.conv2d(torch.zeros((0, 8, 8, 64), dtype=torch.bfloat16, device=DEV), pc)
    assert y.shape == (0, 8, 8, 64)
    out = det_ops.roi_align([torch.zeros((1, 8, 8, 16), dtype=torch.bfloat16, device=DEV)], torch.zeros((0, 5), device=DEV), 7, [0.25])
    assert out.shape == (0, 7, 7, 16)


def test_empty_batches_through_the_newer_ops():
    """Zero images / zero cells through the fused heat-map op, the YOLO decodes and the channel-slice conv."""
    from minddet_amd import det_ops, nn_ops

    bf = dict(dtype=torch.bfloat16, device=DEV)
    heat, hm = det_ops.heat_peaks(torch.zeros((0, 16, 16, 88), **bf), 0, 80, with_hm=True)
    assert heat.shape == (0, 80, 16, 16) and hm.shape == heat.shape
    boxes = torch.zeros((0, 48, 4), device=DEV)
    scores, labels = torch.zeros((0, 48), device=DEV), torch.zeros((0, 48), dtype=torch.int32, device=DEV)
    det_ops.yolo_decode(torch.zeros((0, 4, 4, 256), **bf), boxes, scores, labels, 80, 3, 8.0, [10, 13, 16, 30, 33, 23], 0.25, 0, 48)
    boxes8 = torch.zeros((0, 16, 4), device=DEV)
    det_ops.yolov8_decode(torch.zeros((0, 4, 4, 144), **bf), boxes8, torch.zeros((0, 16), device=DEV),
                          torch.zeros((0, 16), dtype=torch.int32, device=DEV), 80, 16, 8.0, 0.25, 0, 16)
    pc3 = nn_ops.pack_conv(torch.randn((64, 64, 3, 3)) * 0.1, pad=1).to(DEV)
    z = nn_ops.conv2d(torch.zeros((0, 8, 8, 192), **bf), pc3, x_c_off=64)
    assert z.shape == (0, 8, 8, 64)
    torch.cuda.synchronize()


def test_degenerate_boxes_do_not_crash():
    from minddet_amd import det_ops

    b = torch.zeros((130, 7), device=DEV)
    b[:, 3:5] = 1.0
    b[3, 0] = float("nan")
    b[7, 3] = float("inf")
    b[9, 6] = 1e30
    keep, num = det_ops.NumGpu()(b, 0.5)
    torch.cuda.synchronize()
    assert 1 <= int(num[0]) <= 130
    a = torch.tensor([[0.0, 0.0, -5.0, 3.0]], device=DEV)  # inverted box: zero IoU by the iw > 0 test
    assert float(det_ops.iou_jit(a, a)[0, 0]) == 0.0


def test_size_limits_and_bad_arguments_return_codes():
    from minddet_amd import _lib, det_ops, nn_ops

    with pytest.raises(_lib.MindDetHipError, match="rc=4"):
        det_ops.nms_aligned(torch.zeros((70000, 4), device=DEV), 0.5, mode=2)
    with pytest.raises(_lib.MindDetHipError, match="rc=4"):
        det_ops.topk_segmented(torch.zeros((100,), device=DEV), torch.tensor([0, 100], dtype=torch.int32, device=DEV), 5000)
    with pytest.raises(_lib.MindDetHipError, match="rc=2"):
        _lib.call("NmsGpu", [torch.zeros((4, 6), device=DEV), torch.zeros((1,), device=DEV),
                             torch.zeros((4,), dtype=torch.int64, device=DEV), torch.zeros((1,), dtype=torch.int32, device=DEV)])
    with pytest.raises(_lib.MindDetHipError, match="rc=2"):
        _lib.call("NmsGpu", [torch.zeros((4, 7), device=DEV), torch.zeros((1,), device=DEV),
                             torch.zeros((4,), dtype=torch.int32, device=DEV), torch.zeros((1,), dtype=torch.int32, device=DEV)])
    with pytest.raises(_lib.MindDetHipError, match="rc=1"):
        _lib.call("md_iou_aligned", [torch.zeros((4, 4), device=DEV)])
    # keep-list NMS ops: an output shorter than the box list (or a missing num / operand) is an argument error, not an
    # out-of-bounds device write; the 65536-box cap is a size error (rc 4)
    for op, kdt in (("NmsGpu", torch.int64), ("NmsNormalGpu", torch.int64), ("boxes_iou_nms_gpu", torch.int32)):
        guard = torch.full((8,), -5, dtype=kdt, device=DEV)
        with pytest.raises(_lib.MindDetHipError, match="rc=2"):
            _lib.call(op, [torch.rand((8, 7), device=DEV), torch.full((1,), 0.5, device=DEV), guard[:4], torch.zeros((1,), dtype=torch.int32, device=DEV)])
        with pytest.raises(_lib.MindDetHipError, match="rc=2"):
            _lib.call(op, [torch.rand((8, 7), device=DEV), torch.full((1,), 0.5, device=DEV), guard, torch.zeros((0,), dtype=torch.int32, device=DEV)])
        with pytest.raises(_lib.MindDetHipError, match="rc=2"):
            _lib.call(op, [torch.rand((8, 7), device=DEV), None, guard, torch.zeros((1,), dtype=torch.int32, device=DEV)])
        torch.cuda.synchronize()
        assert bool((guard == -5).all())
        with pytest.raises(_lib.MindDetHipError, match="rc=4"):
            _lib.call(op, [torch.zeros((65537, 7), device=DEV), torch.full((1,), 0.5, device=DEV), torch.zeros((65537,), dtype=kdt, device=DEV),
                           torch.zeros((1,), dtype=torch.int32, device=DEV)])
    pc = nn_ops.pack_conv(torch.randn((64, 64, 3, 3)), stride=1, pad=1).to(DEV)
    with pytest.raises(_lib.MindDetHipError, match="rc=2"):  # output shape does not match the conv geometry
        nn_ops.conv2d(torch.zeros((1, 8, 8, 64), dtype=torch.bfloat16, device=DEV), pc,
                      out=torch.zeros((1, 9, 8, 64), dtype=torch.bfloat16, device=DEV))
    with pytest.raises(_lib.MindDetHipError):
        _lib.call("NmsGpu", [torch.zeros((4, 7)), torch.zeros((1,)), torch.zeros((4,), dtype=torch.int64),
                             torch.zeros((1,), dtype=torch.int32)])  # host tensors are refused before the call
    # md_stem_pool: the image must be a multiple of 16 x 64 and arrive in the 4-channel stem layout
    ps = nn_ops.pack_stem(torch.randn((64, 3, 7, 7))).to(DEV)
    with pytest.raises(_lib.MindDetHipError, match="rc=2"):
        nn_ops.stem_pool(torch.zeros((1, 40 + 16, 64 + 16, 4), dtype=torch.bfloat16, device=DEV), ps)   # H % 16 != 0
    with pytest.raises(_lib.MindDetHipError, match="rc=2"):
        _lib.call("md_stem_pool", [torch.zeros((1, 48, 80, 8), dtype=torch.bfloat16, device=DEV), ps.w, ps.bias,
                                   torch.zeros((1, 8, 16, 64), dtype=torch.bfloat16, device=DEV)])      # 8-channel input
    assert nn_ops.stem_pool(torch.zeros((0, 32 + 16, 64 + 16, 4), dtype=torch.bfloat16, device=DEV), ps).shape == (0, 8, 16, 64)
    # md_deform_cols: the offset tensor must carry 3*k*k channels ; md_image_preprocess: uint8 HWC input only
    with pytest.raises(_lib.MindDetHipError, match="rc=2"):
        _lib.call("md_deform_cols", [torch.zeros((1, 8, 8, 64), dtype=torch.bfloat16, device=DEV),
                                     torch.zeros((1, 8, 8, 24), dtype=torch.bfloat16, device=DEV),
                                     torch.zeros((1, 8, 8, 576), dtype=torch.bfloat16, device=DEV)], extra=nn_ops._PoolAttrs3(3, 1, 1, 0))
    with pytest.raises(_lib.MindDetHipError, match="rc=2"):
        nn_ops.image_preprocess(torch.zeros((1, 8, 8, 3), dtype=torch.float32, device=DEV),
                                torch.zeros((1, 6), dtype=torch.float32, device=DEV), (0, 0, 0), (1, 1, 1), (16, 64))
    # md_assign_targets: more ground-truth boxes than the kernel's LDS table -> size error
    with pytest.raises(_lib.MindDetHipError, match="rc=4"):
        det_ops.assign_targets(torch.zeros((8, 7), device=DEV), torch.ones((1025, 7), device=DEV), None, 0.6, 0.45)


def test_caller_workspace_is_used():
    from minddet_amd import _lib, det_ops
    import oracle

    rng = np.random.default_rng(1)
    b = np.zeros((500, 7), np.float32)
    b[:, :2] = rng.uniform(-10, 10, (500, 2)); b[:, 3:6] = rng.uniform(1, 4, (500, 3)); b[:, 6] = rng.uniform(-3, 3, 500)
    bt = torch.from_numpy(b).to(DEV)
    keep = torch.empty((500,), dtype=torch.int32, device=DEV)
    num = torch.empty((1,), dtype=torch.int32, device=DEV)
    need = (500 * 80 + 255) // 256 * 256 + 500 * 8 * 8
    ws = torch.empty((need,), dtype=torch.uint8, device=DEV)
    _lib.call("boxes_iou_nms_gpu", [bt, torch.tensor([0.3], device=DEV), keep, num, ws])
    k_o, n_o = oracle.nms_rot_aot(b, 0.3)
    assert int(num[0]) == n_o and (keep.cpu().numpy() == k_o).all()
    with pytest.raises(_lib.MindDetHipError, match="rc=4"):  # workspace too small
        _lib.call("boxes_iou_nms_gpu", [bt, torch.tensor([0.3], device=DEV), keep, num, ws[:1000]])


def test_scratch_pool_without_a_workspace_param():
    """a caller that passes NO workspace (a MindSpore binding written as INTEGRATION.md 1 shows) gets the library's
    per-(device, stream) pool -- same results as with a caller workspace, on two streams at once, repeatedly (the buffer is reused, so call
    i + 1 must not disturb call i's queued kernels: stream order), with growth (a larger N after a smaller one) and after md_scratch_release."""
    from minddet_amd import _lib
    import oracle

    rng = np.random.default_rng(5)

    def boxes(n):
        b = np.zeros((n, 7), np.float32)
        b[:, :2] = rng.uniform(-12, 12, (n, 2)); b[:, 3:6] = rng.uniform(1, 4, (n, 3)); b[:, 6] = rng.uniform(-3, 3, n)
        return b

    sets = [boxes(n) for n in (300, 1500, 700, 4000, 64)]        # grows twice, shrinks in between (no re-allocation then)
    want = [oracle.nms_rot_aot(b, 0.3) for b in sets]
    thr = torch.tensor([0.3], device=DEV)
    s1, s2 = torch.cuda.Stream(device=DEV), torch.cuda.Stream(device=DEV)
    for round_ in range(2):
        outs = []
        for i, b in enumerate(sets):
            st = (s1, s2)[i & 1]
            with torch.cuda.stream(st):
                bt = torch.from_numpy(b).to(DEV, non_blocking=False)
                keep = torch.full((b.shape[0],), -1, dtype=torch.int32, device=DEV)
                num = torch.full((1,), -1, dtype=torch.int32, device=DEV)
                _lib.call("boxes_iou_nms_gpu", [bt, thr, keep, num])       # four params: no workspace
                outs.append((bt, keep, num))
        torch.cuda.synchronize()
        for (bt, keep, num), (k_o, n_o) in zip(outs, want):
            assert int(num[0]) == n_o and (keep.cpu().numpy() == k_o).all()
        assert _lib.lib().md_scratch_release() == 0                           # the second round starts from an empty pool
        torch.cuda.synchronize()
