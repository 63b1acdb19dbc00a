"""-m gpu: graphs.SplitForward (test_cfg.streams): the batch as equal parts on separate HIP streams must give exactly the single-stream
outputs (same kernels on the same images: bit-identical), repeatedly, with the lazily built constants created race-free, for every detector
family that carries it; the big-shard configs that enable it (Mask R-CNN R101-FPN, YOLOv8l) at their bench batch."""
import pytest
import torch

from tests.conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs MI355X")]
DEV = "cuda:0"


def _images(n, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.zeros((n, h, w, 8))
    x[..., :3] = torch.randn((n, h, w, 3), generator=g)
    return x.to(torch.bfloat16).to(DEV)


@pytest.mark.parametrize("cfgp,hw", [("configs/mask_rcnn/mask_rcnn_tiny.py", (128, 192)), ("configs/faster_rcnn/faster_rcnn_tiny.py", (128, 192)),
                                     ("configs/yolov8/yolov8_tiny.py", (128, 160))])
def test_split_forward_is_bit_identical_tiny(cfgp, hw):
    from minddet.models import Config, build_detector
    from minddet_amd.graphs import SplitForward

    cfg = Config.fromfile(cfgp)
    m = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).to(DEV)
    assert m.streams == int((cfg.test_cfg or {}).get("streams", 1)) and m.forward_split.n == m.streams
    x = _images(8, hw[0], hw[1], 11)
    ref = m.forward(x)
    torch.cuda.synchronize()
    for parts in (2, 4):
        sp = SplitForward(m, parts)          # the FIRST call builds the graph's lazy constants before the streams fork
        for rep in range(3):
            out = sp(x)
            torch.cuda.synchronize()
            assert len(out) == len(ref)
            for a, b in zip(out, ref):
                assert a.shape == b.shape and torch.equal(a, b), (cfgp, parts, rep)
    # the parts' inputs made ON the parts' streams from batch-first tensors (bench.py: md_image_preprocess of a uint8 part)
    scale = torch.full((8, 1, 1, 1), 2.0, dtype=torch.bfloat16, device=DEV)
    out = SplitForward(m, 2)((x * 0.5, scale), prepare=lambda a, b: a * b)
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(out, ref))
    # an indivisible batch / one stream falls back to the plain forward
    out = SplitForward(m, 3)(x)
    assert all(torch.equal(a, b) for a, b in zip(out, ref))
    # the sticky pre-NMS prefix flags live in one tensor per (batch, device, stream): the parts' streams never share a row
    assert len({k[2] for k in m.prefix_status._t}) >= 3


@pytest.mark.parametrize("cfgp,hw", [("configs/faster_rcnn/faster_rcnn_tiny.py", (128, 192)), ("configs/yolov8/yolov8_tiny.py", (128, 160))])
def test_split_forward_after_weight_reload(cfgp, hw):
    """r03 ADVICE: .to() after a weight change drops the lazily built packs (fused blocks, merged convs); the pass that rebuilds them must be a
    single-stream one again -- the priming key carries the model's pack generation"""
    from minddet.models import Config, build_detector
    from minddet_amd.graphs import SplitForward

    cfg = Config.fromfile(cfgp)
    m = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).to(DEV)
    x = _images(8, hw[0], hw[1], 5)
    sp = SplitForward(m, 2)
    before = sp(x)
    torch.cuda.synchronize()
    gen0, n_primed = m._pack_gen, len(sp._primed)
    for mod in m.conv_modules():          # new weights in every conv
        mod.weight = mod.weight * 0.5 + 0.01
    m.to(DEV)
    assert m._pack_gen == gen0 + 1
    out = sp(x)                            # must prime again (new key), then fork
    torch.cuda.synchronize()
    assert len(sp._primed) == n_primed + 1
    ref = m.forward(x)
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(out, ref))
    assert not all(torch.equal(a, b) for a, b in zip(out, before))   # the new weights are the ones that ran


def test_split_forward_mask_rcnn_r101_bench_shard():
    """configs/mask_rcnn/mask_rcnn_r101_fpn.py enables two streams: at a 8-image batch (two 4-image halves) the outputs, pasted masks included,
    equal the single-stream ones bit for bit"""
    from minddet.models import Config, build_detector
    from minddet_amd import nn_ops
    from minddet_amd.data import synthetic_images

    cfg = Config.fromfile("configs/mask_rcnn/mask_rcnn_r101_fpn.py")
    m = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).to(DEV)
    assert m.streams == 2
    x = nn_ops.to_stem_layout(synthetic_images(8, 800, 1344, seed=20240317, device=DEV))
    ref = m.forward(x, paste=True)
    out = m.forward_split(x, paste=True)
    out2 = m.forward_split(x, paste=True)
    torch.cuda.synchronize()
    assert len(out) == 4
    for a, b, c in zip(out, ref, out2):
        assert torch.equal(a, b) and torch.equal(c, b)


def test_split_forward_yolov8l_bench_shard():
    from minddet.models import Config, build_detector
    from minddet_amd import nn_ops
    from minddet_amd.data import synthetic_images

    cfg = Config.fromfile("configs/yolov8/yolov8l.py")
    m = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).to(DEV)
    assert m.streams == 2
    x = synthetic_images(32, 640, 640, seed=20240317, device=DEV)
    if nn_ops.stem_layout_ok(640, 640) and getattr(m, "stem", None) is not None:
        x = nn_ops.to_stem_layout(x)
    ref = m.forward(x)
    for _ in range(3):
        out = m.forward_split(x)
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(out, ref))
