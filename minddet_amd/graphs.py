"""Inference graphs as sequences of libminddet_hip.so calls (NHWC bf16 activations).

Reference-present graphs (cited per class) and the build-authored two-stage graph (absent from the
reference, SURVEY 0.2 -- standard public architecture, "parity unpinned").  Each module keeps its
fp32 parameters in the reference's layout (`.weight [Cout,Cin,kh,kw]`, `.bn = (gamma, beta, mean,
var, eps)`) so that the CPU oracle (oracle/nets.py) can run the same graph, and a packed bf16 copy
for the kernels.  Random init follows SURVEY 8(d): He-normal std = sqrt(2/(k*k*Cout))
(centernet/src/resnet.py:210-213), BN gamma=1, beta=0, mean~N(0,0.1), var~U(0.5,1.5).
"""
import math
import os

import numpy as np
import torch

from . import det_ops, nn_ops
from .registry import BACKBONES, DETECTORS, HEADS, NECKS, ROI_HEAD, build_backbone, build_head, build_neck, build_roi_head


# per-level proposal selection (slice, top-k, decode) on a side HIP stream behind the level's conv, joined before the NMS: +0.4 % on
# the benchmark (same-box 1 856 -> 1 863 images/s); 0 = everything on one stream (A/B)
RPN_OVERLAP = os.environ.get("MD_RPN_OVERLAP", "1") == "1"
# stride-1 bottleneck blocks with 64 mid channels (ResNet-50 / 101 stage 1) as ONE md_bottleneck launch instead of three md_conv2d
# launches (bit-identical results; MD_FUSE_BLOCKS=0 keeps the layer-by-layer path for A/B)
FUSE_BLOCKS = os.environ.get("MD_FUSE_BLOCKS", "1") == "1"
# first block of the later stages: conv3 + the strided 1x1 downsample conv as one K-concatenated GEMM (md_conv1x1_dual; MD_FUSE_DUAL=0: two launches)
FUSE_DUAL = os.environ.get("MD_FUSE_DUAL", "1") == "1"
RPN_FUSED_HEAD = os.environ.get("MD_RPN_FUSED", "1") == "1"  # 0: two md_conv2d launches per level (A/B)


class SplitForward:
    """`model.forward` on the S equal parts of a batch, each part on its own HIP stream, outputs concatenated in input order.

    Why: a layer's launch ends with a partly filled round of workgroups (Mask R-CNN's 32-image shard: 525 ping-pong tiles on 256 CUs = 2.05
    rounds -> 3) and with completion skew; with two streams the other half's next launch fills those CUs.  Same kernels on the same images
    -> bit-identical outputs (tests/test_split_forward_gpu.py).  Measured r03, same box, tools/two_stream_halves.py: Mask R-CNN R101 b32
    26.50 -> 25.04 ms/step (-5.5 %), YOLOv8l b32 7.50 -> 7.28 (-2.9 %), Faster R-CNN R50 b120 56.74 -> 56.10 (-1.1 %); YOLOv5s b32 gets 60 %
    SLOWER (its 1.8 ms step is host-enqueue-bound: twice the launches); 3 / 4 / 8 streams lose everywhere.  Opt-in per config: test_cfg.streams.
    The lazily built constants of the graph (anchors, segment offsets) are created by one single-stream pass per part shape before two
    streams read them."""

    def __init__(self, model, streams=2):
        self.model, self.n = model, int(streams)
        self._streams = None
        self._primed = set()

    def __call__(self, images, prepare=None, **kw):
        """images: the batch -- or, with `prepare`, a tuple of batch-first tensors that `prepare(*part)` turns into a part's input batch ON the
        part's stream (pre-processing from uint8: otherwise that launch would sit alone on the caller's stream between two joins)"""
        ins = images if isinstance(images, (tuple, list)) else (images,)
        B = ins[0].shape[0]
        if self.n <= 1 or B < self.n or B % self.n or kw.get("return_aux"):
            return self.model.forward(images if prepare is None else prepare(*ins), **kw)
        dev = ins[0].device
        if self._streams is None or self._streams[0].device != dev:   # (a model moved to another GPU gets streams, and a priming pass, there)
            self._streams = [torch.cuda.Stream(device=dev) for _ in range(self.n)]
            self._primed = set()
        step = B // self.n

        def part(i):
            p = [t[i * step:(i + 1) * step] for t in ins]
            return p[0] if prepare is None else prepare(*p)

        cur = torch.cuda.current_stream(dev)
        # the model's pack generation is part of the key: .to() after a weight reload drops the lazily built packs (fused blocks, merged
        # convs), and the pass that rebuilds them must again be a single-stream one (r03 ADVICE: part 1 would read them unsynchronised)
        key = (getattr(self.model, "_pack_gen", 0), step, prepare is not None) + tuple(ins[0].shape[1:])
        if key not in self._primed:
            self.model.forward(part(0), **kw)
            self._primed.add(key)
        outs = []
        for i, st in enumerate(self._streams):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs.append(self.model.forward(part(i), **kw))
        for st in self._streams:
            cur.wait_stream(st)
        res = []
        for k in range(len(outs[0])):
            ts = [o[k] for o in outs]
            for t in ts:
                t.record_stream(cur)   # allocated on a part's stream, read (and later freed) on the caller's
            res.append(torch.cat(ts, 0))
        return tuple(res)


def _split_forward_of(model, test_cfg):
    """test_cfg.streams (default 1) -> model.streams and model.forward_split (== model.forward for one stream)"""
    model.streams = int((test_cfg or {}).get("streams", 1))
    model.forward_split = SplitForward(model, model.streams)
    model._pack_gen = 0


def _packs_rebuilt(model):
    """called by every detector's .to(): the packed weights (and with them every lazily derived pack) are new objects from here on"""
    model._pack_gen = getattr(model, "_pack_gen", 0) + 1


class ParamInit:
    """Deterministic parameter source (numpy Generator, seed 7 by default)."""

    def __init__(self, seed=7):
        self.rng = np.random.default_rng(seed)

    def conv(self, cout, cin, k, std=None):
        std = math.sqrt(2.0 / (k * k * cout)) if std is None else std
        return torch.from_numpy(self.rng.normal(0.0, std, (cout, cin, k, k)).astype(np.float32))

    def bn(self, c, eps=1e-5):
        return (torch.ones(c), torch.zeros(c), torch.from_numpy(self.rng.normal(0, 0.1, c).astype(np.float32)),
                torch.from_numpy(self.rng.uniform(0.5, 1.5, c).astype(np.float32)), eps)

    def bias(self, c, value=None, std=0.01):
        if value is not None:
            return torch.full((c,), float(value))
        return torch.from_numpy(self.rng.normal(0, std, c).astype(np.float32))


class ConvModule:
    """Conv2d [+ BatchNorm2d(eval)] [+ ReLU] -> one md_conv2d call."""

    def __init__(self, init, cin, cout, k, stride=1, pad=0, bn=True, relu=True, bias=False, bn_eps=1e-5, std=None,
                 bias_value=None, act=None):
        # act: None -> `relu` decides ('relu' after the residual add / none); 'silu' -> SiLU before the residual add
        self.act = act if act is not None else ("relu" if relu else None)
        relu = self.act == "relu"
        self.cin, self.cout, self.k, self.stride, self.pad, self.relu = cin, cout, k, stride, pad, relu
        self.weight = init.conv(cout, cin, k, std)
        self.bn = init.bn(cout, bn_eps) if bn else None
        self.bias = init.bias(cout, bias_value) if bias else None
        self.packed = None

    def to(self, device):
        self.packed = nn_ops.pack_conv(self.weight, bias=self.bias, bn=self.bn, stride=self.stride, pad=self.pad,
                                       relu=self.act).to(device)
        return self

    def __call__(self, x, residual=None, out=None, c_off=0, x_c_off=None, res_c_off=None):
        return nn_ops.conv2d(x, self.packed, residual=residual, out=out, c_off=c_off, x_c_off=x_c_off, res_c_off=res_c_off)

    def macs(self, ho, wo):
        return ho * wo * self.cout * self.cin * self.k * self.k


def merged_conv(mods, device):
    """Several ConvModules that read the same input (same kernel / stride / pad / activation) as ONE md_conv2d launch whose
    output channels are the modules' outputs side by side: the same arithmetic per output channel, the input read once."""
    m0 = mods[0]
    if any((m.cin, m.k, m.stride, m.pad, m.act, m.bn is None, m.bias is None) != (m0.cin, m0.k, m0.stride, m0.pad, m0.act, m0.bn is None, m0.bias is None)
           for m in mods):
        raise ValueError("merged_conv: the convs must share input, geometry and activation")
    w = torch.cat([m.weight for m in mods], 0)
    bn = None if m0.bn is None else tuple(torch.cat([torch.as_tensor(m.bn[i]) for m in mods]) for i in range(4)) + (m0.bn[4],)
    bias = None if m0.bias is None else torch.cat([m.bias for m in mods])
    return nn_ops.pack_conv(w, bias=bias, bn=bn, stride=m0.stride, pad=m0.pad, relu=m0.act).to(device)


# ----------------------------------------------------------------------------- ResNet (centernet/src/resnet.py)
class BasicBlock:
    """centernet/src/resnet.py:109-136"""
    expansion = 1

    def __init__(self, init, inplanes, planes, stride=1, downsample=None):
        self.conv1 = ConvModule(init, inplanes, planes, 3, stride, 1)
        self.conv2 = ConvModule(init, planes, planes, 3, 1, 1, relu=True)  # ReLU applied after the residual add
        self.downsample = downsample

    def modules(self):
        return [self.conv1, self.conv2] + ([self.downsample] if self.downsample else [])

    def __call__(self, x):
        residual = self.downsample(x) if self.downsample is not None else x
        out = self.conv1(x)
        return self.conv2(out, residual=residual)


class Bottleneck:
    """centernet/src/resnet.py:139-178 (stride on the 3x3 conv)"""
    expansion = 4

    def __init__(self, init, inplanes, planes, stride=1, downsample=None):
        self.conv1 = ConvModule(init, inplanes, planes, 1)
        self.conv2 = ConvModule(init, planes, planes, 3, stride, 1)
        self.conv3 = ConvModule(init, planes, planes * 4, 1, relu=True)  # ReLU after the residual add
        self.downsample = downsample
        self._fused = False
        self._dual = False

    def modules(self):
        return [self.conv1, self.conv2, self.conv3] + ([self.downsample] if self.downsample else [])

    def __call__(self, x):
        if FUSE_BLOCKS:
            if self._fused is False:   # packed on first use (the convs are packed by ResNet.to)
                self._fused = nn_ops.pack_bottleneck(self.conv1.packed, self.conv2.packed, self.conv3.packed,
                                                     self.downsample.packed if self.downsample is not None else None)
            if self._fused is not None:
                # one launch for the whole block (md_bottleneck): x is read once, the 64-channel intermediates stay in LDS, and the
                # first block's 1x1 downsample conv is computed from the same x tile
                res = self.downsample(x) if (self.downsample is not None and self._fused.wd is None) else None
                return nn_ops.bottleneck(x, self._fused, residual=res)
        if FUSE_DUAL and self.downsample is not None:
            # first block of a stage: conv3 and the (strided) 1x1 downsample conv as ONE GEMM over [t2 ; x] (md_conv1x1_dual)
            if self._dual is False:
                self._dual = nn_ops.pack_dual(self.conv3.packed, self.downsample.packed)
            if self._dual is not None:
                return nn_ops.conv1x1_dual(self.conv2(self.conv1(x)), x, self._dual)
        residual = self.downsample(x) if self.downsample is not None else x
        return self.conv3(self.conv2(self.conv1(x)), residual=residual)


@BACKBONES.register_module
class ResNet:
    """centernet/src/resnet.py:181-252: 7x7/2 stem + BN + ReLU, zero-pad + MaxPool(3,2), 4 stages;
    returns (C2, C3, C4, C5).  depth 18/34 -> BasicBlock, 50/101 -> Bottleneck."""
    SETTINGS = {18: (BasicBlock, [2, 2, 2, 2]), 34: (BasicBlock, [3, 4, 6, 3]), 50: (Bottleneck, [3, 4, 6, 3]),
                101: (Bottleneck, [3, 4, 23, 3])}

    def __init__(self, depth=50, base_width=64, seed=7, init=None, layers=None):
        block, default_layers = self.SETTINGS[depth]
        layers = layers or default_layers
        init = init or ParamInit(seed)
        self.block, self.inplanes = block, base_width
        self.conv1 = ConvModule(init, 3, base_width, 7, 2, 3)
        self.stages = []
        for i, n in enumerate(layers):
            self.stages.append(self._make_layer(init, block, base_width * (2 ** i), n, 1 if i == 0 else 2))
        self.out_channels = [base_width * (2 ** i) * block.expansion for i in range(4)]

    def _make_layer(self, init, block, planes, blocks, stride):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = ConvModule(init, self.inplanes, planes * block.expansion, 1, stride, 0, relu=False)
        layers = [block(init, self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(init, self.inplanes, planes))
        return layers

    def modules(self):
        out = [self.conv1]
        for st in self.stages:
            for b in st:
                out += b.modules()
        return out

    def to(self, device):
        for m in self.modules():
            m.to(device)
        # the fused-block / dual-GEMM packs are derived from the conv packs just rebuilt (PackedDual and PackedBottleneck.b12 hold their own
        # copies): drop them, or a forward pass after a weight change + .to() would keep running the old weights in those launches
        for st in self.stages:
            for b in st:
                if isinstance(b, Bottleneck):
                    b._fused = b._dual = False
        self.stem = None
        if self.conv1.cout == 64 and self.conv1.relu:  # the fused conv + BN + ReLU + maxpool kernel (md_stem_pool)
            self.stem = nn_ops.pack_stem(self.conv1.weight, bn=self.conv1.bn, bias=self.conv1.bias).to(device)
        return self

    def __call__(self, x):
        """x: [N,H,W,8] bf16 NHWC (3 real channels), or the same batch already in the stem layout
        ([N,H+16,W+16,4], nn_ops.to_stem_layout) -- then the whole stem is one launch."""
        if x.shape[3] == 4:
            if self.stem is None:
                raise nn_ops._lib.MindDetHipError("ResNet: stem-layout input needs the 64-channel ReLU stem")
            x = nn_ops.stem_pool(x, self.stem)
        else:
            x = self.conv1(x)
            x = nn_ops.maxpool2d(x, 3, 2, 1, zero_pad=True)
        outs = []
        for st in self.stages:
            for b in st:
                x = b(x)
            outs.append(x)
        return tuple(outs)


# ----------------------------------------------------------------------------- FPN / RPN / RoI head (build-authored)
@NECKS.register_module
class FPN:
    """Lin et al. 2017: 1x1 laterals, top-down nearest upsample + add, 3x3 output convs, extra level by
    stride-2 subsampling (max_pool k=1 s=2) of the last output.  parity unpinned."""

    def __init__(self, in_channels, out_channels=256, num_outs=5, seed=11, init=None):
        init = init or ParamInit(seed)
        self.lateral = [ConvModule(init, c, out_channels, 1, bn=False, relu=False, bias=True) for c in in_channels]
        self.output = [ConvModule(init, out_channels, out_channels, 3, 1, 1, bn=False, relu=False, bias=True)
                       for _ in in_channels]
        self.num_outs = num_outs

    def modules(self):
        return self.lateral + self.output

    def to(self, device):
        for m in self.modules():
            m.to(device)
        return self

    def __call__(self, feats):
        # top-down path: lateral_i + nearest_up(merged_{i+1}).  When the pyramid halves exactly the add is fused into
        # the lateral 1x1 conv as an upsampled residual (md_conv2d res_upsample); otherwise the streaming kernel runs.
        lats = [None] * len(feats)
        lats[-1] = self.lateral[-1](feats[-1])
        for i in range(len(feats) - 2, -1, -1):
            h, w = feats[i].shape[1], feats[i].shape[2]
            top = lats[i + 1]
            if top.shape[1] == (h + 1) // 2 and top.shape[2] == (w + 1) // 2:
                lats[i] = nn_ops.conv2d(feats[i], self.lateral[i].packed, residual=top, res_upsample=True)
            else:
                lats[i] = nn_ops.upsample_add(self.lateral[i](feats[i]), top)
        outs = [o(l) for o, l in zip(self.output, lats)]
        while len(outs) < self.num_outs:
            outs.append(nn_ops.maxpool2d(outs[-1], 1, 2, 0, zero_pad=False))
        return outs


@HEADS.register_module
class RPNHead:
    """Shared 3x3 conv + ReLU, then one fused 1x1 conv producing [A objectness | 4A deltas] channels.
    Proposal generation: per level top-k (nms_pre) on the logits, delta2bbox + clip, NMS (iou_thr),
    per image top-`max_per_img` across levels.  parity unpinned (absent from the reference)."""

    def __init__(self, in_channels=256, feat_channels=256, strides=(4, 8, 16, 32, 64), scale=8.0,
                 ratios=(0.5, 1.0, 2.0), nms_pre=1000, max_per_img=1000, nms_thr=0.7, seed=13, init=None):
        init = init or ParamInit(seed)
        self.A = len(ratios)
        self.conv = ConvModule(init, in_channels, feat_channels, 3, 1, 1, bn=False, relu=True, bias=True, std=0.01)
        self.out = ConvModule(init, feat_channels, 5 * self.A, 1, bn=False, relu=False, bias=True, std=0.01)
        self.strides, self.scale, self.ratios = tuple(strides), scale, tuple(ratios)
        self.nms_pre, self.max_per_img, self.nms_thr = nms_pre, max_per_img, nms_thr
        self._cache = {}
        self._side = None

    def modules(self):
        return [self.conv, self.out]

    def to(self, device):
        for m in self.modules():
            m.to(device)
        return self

    def _static(self, feats, img_hw):
        B = feats[0].shape[0]
        sizes = tuple((f.shape[1], f.shape[2]) for f in feats)
        key = (B, sizes, img_hw)
        if key not in self._cache:
            dev = feats[0].device
            anchors = det_ops.fpn_anchors(sizes, self.strides, self.scale, self.ratios, device=dev)
            per, offs, o = [], [], 0
            for (h, w) in sizes:
                n = h * w * self.A
                per.append(anchors[o:o + n])
                offs.append(torch.arange(0, (B + 1) * n, n, dtype=torch.int32, device=dev))
                o += n
            L, k = len(sizes), self.nms_pre
            self._cache[key] = dict(anchors=per, seg=offs,
                                    merged_seg=torch.arange(0, (B + 1) * L * k, L * k, dtype=torch.int32, device=dev))
        return self._cache[key]

    def __call__(self, feats, img_hw):
        st = self._static(feats, img_hw)
        B, L, k, A = feats[0].shape[0], len(feats), self.nms_pre, self.A
        dev = feats[0].device
        boxes = torch.empty((L, B, k, 4), dtype=torch.float32, device=dev)
        scores = torch.empty((L, B, k), dtype=torch.float32, device=dev)
        counts = torch.empty((L, B), dtype=torch.int32, device=dev)
        heads = []
        # RPN_OVERLAP: the per-level selection (slice, top-k, decode: small latency-bound launches) runs on a side stream behind the
        # level's conv while the main stream goes on with the next level's conv; the streams join before the NMS
        main = torch.cuda.current_stream(dev) if RPN_OVERLAP else None
        if RPN_OVERLAP and self._side is None:
            self._side = torch.cuda.Stream(dev)
        keep_alive = []
        if RPN_OVERLAP:
            # written on the side stream, allocated on the main one: the caching allocator must not hand their memory to a
            # main-stream allocation while side-stream work may still touch it, whatever happens to the Python references
            for t in (boxes, scores, counts):
                t.record_stream(self._side)
        try:
            for l, f in enumerate(feats):
                if RPN_FUSED_HEAD and self.conv.cout == 256 and self.out.packed.cout == 16 and self.conv.relu:
                    head = nn_ops.conv2d_head(f, self.conv.packed, self.out.packed)   # [B,H,W,16], one launch per level
                else:
                    head = self.out(self.conv(f))
                heads.append(head)

                def select(l=l, f=f, head=head):
                    logits = nn_ops.slice_cast(head, 0, A)              # [B,H,W,A] fp32
                    _, idx, cnt = det_ops.topk_segmented(logits, st["seg"][l], k, out_cnt=counts[l],
                                                         max_segment=f.shape[1] * f.shape[2] * A)
                    det_ops.rpn_decode(head, st["anchors"][l], idx, cnt, A, img_hw, out_boxes=boxes[l], out_scores=scores[l])
                    keep_alive.append((logits, idx, cnt))

                if RPN_OVERLAP:
                    head.record_stream(self._side)
                    ev = torch.cuda.Event()
                    ev.record(main)
                    with torch.cuda.stream(self._side):
                        self._side.wait_event(ev)
                        select()
                else:
                    select()
        finally:
            if RPN_OVERLAP:
                main.wait_stream(self._side)   # always joined, also when a level's selection raised
        keep, _, _ = det_ops.nms_aligned(boxes.view(L * B, k, 4), self.nms_thr, mode=det_ops.NMS_MODE_STRICT,
                                         count=counts.view(-1))
        mboxes, mscores = det_ops.rpn_merge(boxes, scores, keep.view(L, B, k))
        topv, topi, cnt = det_ops.topk_segmented(mscores, st["merged_seg"], self.max_per_img, max_segment=L * k)
        rois, roi_scores = det_ops.make_rois(mboxes, topv, topi, cnt)
        return rois, roi_scores, cnt, dict(heads=heads, boxes=boxes, scores=scores, counts=counts, keep=keep,
                                           mboxes=mboxes, mscores=mscores)


@ROI_HEAD.register_module
class StandardRoIHead:
    """RoIAlign 7x7 over P2..P5 -> FC 1024 -> FC 1024 -> fused [cls (nc+1) | reg (4 nc)] FC, softmax,
    class-specific delta2bbox, score threshold, class-wise NMS, top max_per_img.  parity unpinned."""

    def __init__(self, in_channels=256, fc_channels=1024, num_classes=80, roi_size=7, sampling_ratio=2,
                 featmap_strides=(4, 8, 16, 32), score_thr=0.05, nms_thr=0.5, max_per_img=100, nms_pre=2048,
                 seed=17, init=None):
        init = init or ParamInit(seed)
        self.nc, self.P, self.C = num_classes, roi_size, in_channels
        k_in = in_channels * roi_size * roi_size
        # FC weights are stored in the conv layout [Cout, Cin, 1, 1]; fc1's Cin axis is ordered (h, w, c) to
        # match the NHWC RoIAlign output flattened as-is.
        self.fc1 = ConvModule(init, k_in, fc_channels, 1, bn=False, relu=True, bias=True, std=math.sqrt(2.0 / k_in))
        self.fc2 = ConvModule(init, fc_channels, fc_channels, 1, bn=False, relu=True, bias=True,
                              std=math.sqrt(2.0 / fc_channels))
        self.reg_offset = (num_classes + 1 + 7) // 8 * 8
        n_out = self.reg_offset + 4 * num_classes
        self.fc_out = ConvModule(init, fc_channels, n_out, 1, bn=False, relu=False, bias=True, std=0.01)
        # channels [nc+1, reg_offset) are alignment padding: zero their weights so they stay inert
        self.fc_out.weight[num_classes + 1:self.reg_offset] = 0
        self.fc_out.bias[num_classes + 1:self.reg_offset] = 0
        # class logits get a wider init so that the synthetic benchmark has above-threshold candidates
        self.fc_out.weight[:num_classes + 1] *= 20.0
        self.strides, self.sampling = tuple(featmap_strides), sampling_ratio
        self.score_thr, self.nms_thr, self.max_per_img, self.nms_pre = score_thr, nms_thr, max_per_img, nms_pre
        self._cache = {}
        # nms_pre: the class-wise NMS runs on the top-nms_pre (roi, class) candidates of an image instead of on all of them (the
        # public definition: every candidate above score_thr).  The cut is result-identical whenever the prefix yields max_per_img
        # survivors or was not full; the one other case raises a sticky per-image flag on the device (DESIGN.md "pre-NMS prefix")
        self.prefix_status = det_ops.PrefixStatus()

    def modules(self):
        return [self.fc1, self.fc2, self.fc_out]

    def to(self, device):
        for m in self.modules():
            m.to(device)
        return self

    def __call__(self, feats, rois, roi_cnt, img_hw):
        B = feats[0].shape[0]
        R = rois.shape[0]
        post = R // B
        dev = rois.device
        pooled = det_ops.roi_align(list(feats[:len(self.strides)]), rois, self.P, [1.0 / s for s in self.strides],
                                   self.sampling, True)
        x = nn_ops.linear(pooled.view(R, self.P * self.P * self.C), self.fc1.packed)
        x = nn_ops.linear(x, self.fc2.packed)
        cls_reg = nn_ops.linear(x, self.fc_out.packed)
        cand = det_ops.rcnn_scores(cls_reg, roi_cnt, self.nc, self.score_thr)
        key = (B, post)
        if key not in self._cache:
            self._cache[key] = torch.arange(0, (B + 1) * post * self.nc, post * self.nc, dtype=torch.int32, device=dev)
        sv, si, sc = det_ops.topk_segmented(cand, self._cache[key], self.nms_pre, max_segment=post * self.nc)
        boxes, labels = det_ops.rcnn_decode_selected(cls_reg, rois, si, sc, self.nc, self.reg_offset, img_hw)
        keep, kidx, num = det_ops.nms_aligned(boxes, self.nms_thr, mode=det_ops.NMS_MODE_STRICT, count=sc, group=labels,
                                              max_output=self.max_per_img)
        dets, count = det_ops.pack_detections(boxes, sv, labels, kidx, num, self.max_per_img, sel_cnt=sc,
                                              status=self.prefix_status.tensor(B, dev))
        return dets, count, dict(pooled=pooled, cls_reg=cls_reg, cand=cand, sel_scores=sv, sel_idx=si, sel_cnt=sc,
                                 boxes=boxes, labels=labels, keep=keep)


@DETECTORS.register_module
class FasterRCNN:
    """Two-stage detector: backbone -> neck -> rpn_head -> roi_head.  `forward(images)` takes
    [B,H,W,8] bf16 NHWC (RGB in channels 0..2, 5 zero channels) and returns (dets [B,max_det,6], count [B])."""

    def __init__(self, backbone, neck, rpn_head, roi_head, train_cfg=None, test_cfg=None):
        self.backbone = build_backbone(backbone)
        neck = dict(neck)
        neck.setdefault("in_channels", self.backbone.out_channels)
        self.neck = build_neck(neck)
        self.rpn_head = build_head(rpn_head)
        self.roi_head = build_roi_head(roi_head)
        self.test_cfg = test_cfg
        _split_forward_of(self, test_cfg)

    def to(self, device):
        for m in (self.backbone, self.neck, self.rpn_head, self.roi_head):
            m.to(device)
        _packs_rebuilt(self)
        return self

    def conv_modules(self):
        return self.backbone.modules() + self.neck.modules() + self.rpn_head.modules() + self.roi_head.modules()

    @property
    def prefix_status(self):
        """sticky device flags of the second stage's top-nms_pre cut (det_ops.PrefixStatus)"""
        return self.roi_head.prefix_status

    def extract_feat(self, images):
        return self.neck(self.backbone(images))

    def forward(self, images, return_aux=False):
        img_hw = (images.shape[1], images.shape[2])
        if images.shape[3] == 4:  # stem layout (nn_ops.to_stem_layout): the border is not image
            img_hw = (img_hw[0] - nn_ops.STEM_PAD_LO - nn_ops.STEM_PAD_HI, img_hw[1] - nn_ops.STEM_PAD_LO - nn_ops.STEM_PAD_HI)
        feats = self.extract_feat(images)
        rois, roi_scores, roi_cnt, aux_rpn = self.rpn_head(feats, img_hw)
        dets, count, aux_roi = self.roi_head(feats, rois, roi_cnt, img_hw)
        if return_aux:
            return dets, count, dict(feats=feats, rois=rois, roi_scores=roi_scores, roi_cnt=roi_cnt, rpn=aux_rpn, roi=aux_roi)
        return dets, count

    __call__ = forward

    def macs_per_image(self, h, w, num_rois=None):
        """Algorithmic MACs (SURVEY 8d formula) of every conv/FC for one h x w image."""
        total, hh, ww = 0, h, w
        bb = self.backbone

        def out_hw(m, hh, ww):
            return (hh + 2 * m.pad - m.k) // m.stride + 1, (ww + 2 * m.pad - m.k) // m.stride + 1

        hh, ww = out_hw(bb.conv1, hh, ww)
        total += bb.conv1.macs(hh, ww)
        hh, ww = (hh + 2 - 3) // 2 + 1, (ww + 2 - 3) // 2 + 1
        sizes = []
        for st in bb.stages:
            for b in st:
                h_in, w_in = hh, ww
                for m in b.modules():
                    if m is b.downsample:
                        ho, wo = out_hw(m, h_in, w_in)
                        total += m.macs(ho, wo)
                    else:
                        hh, ww = out_hw(m, hh, ww)
                        total += m.macs(hh, ww)
            sizes.append((hh, ww))
        for l, (a, b_) in zip(self.neck.lateral, sizes):
            total += l.macs(a, b_)
        for o, (a, b_) in zip(self.neck.output, sizes):
            total += o.macs(a, b_)
        lv = list(sizes)
        while len(lv) < self.neck.num_outs:
            lv.append(((lv[-1][0] - 1) // 2 + 1, (lv[-1][1] - 1) // 2 + 1))
        for (a, b_) in lv:
            total += self.rpn_head.conv.macs(a, b_) + self.rpn_head.out.macs(a, b_)
        r = num_rois if num_rois is not None else self.rpn_head.max_per_img
        for m in self.roi_head.modules():
            total += r * m.cout * m.cin
        return total


# ----------------------------------------------------------------------------- CenterNet (reference-present graph)
class DeconvModule:
    """Conv2dTranspose [+ BatchNorm2d] [+ ReLU] -> s*s sub-pixel md_conv2d launches (nn_ops.pack_conv_transpose)."""

    def __init__(self, init, cin, cout, k, stride, pad, bn=True, relu=True, bn_eps=1e-5):
        self.cin, self.cout, self.k, self.stride, self.pad, self.relu = cin, cout, k, stride, pad, relu
        std = math.sqrt(2.0 / (k * k * cout))
        self.weight_t = torch.from_numpy(init.rng.normal(0.0, std, (cin, cout, k, k)).astype(np.float32))
        self.bn = init.bn(cout, bn_eps) if bn else None
        self.packed = None

    def to(self, device):
        self.packed = nn_ops.pack_conv_transpose(self.weight_t, bn=self.bn, stride=self.stride, pad=self.pad,
                                                 relu=self.relu).to(device)
        return self

    def __call__(self, x, out=None, c_off=0):
        return nn_ops.conv_transpose2d(x, self.packed, out=out, c_off=c_off)


@ROI_HEAD.register_module
class FCNMaskHead:
    """Mask R-CNN mask branch (He et al. 2017, FPN variant): RoIAlign 14x14 over P2..P5 on the final detections -> 4 x
    [3x3 conv 256 + ReLU] -> Conv2dTranspose 2x2 s2 + ReLU -> 1x1 conv to num_classes -> the detection's own class channel,
    sigmoid: [B, max_det, 28, 28].  Absent from the reference (Mask R-CNN is a README bullet): parity unpinned."""

    def __init__(self, in_channels=256, conv_channels=256, num_convs=4, num_classes=80, roi_size=14, sampling_ratio=2,
                 featmap_strides=(4, 8, 16, 32), seed=19, init=None):
        init = init or ParamInit(seed)
        self.nc, self.P, self.C = num_classes, roi_size, in_channels
        self.convs = []
        cin = in_channels
        for _ in range(num_convs):
            self.convs.append(ConvModule(init, cin, conv_channels, 3, 1, 1, bn=False, relu=True, bias=True))
            cin = conv_channels
        self.upsample = DeconvModule(init, cin, conv_channels, 2, 2, 0, bn=False, relu=True)
        self.logits = ConvModule(init, conv_channels, num_classes, 1, bn=False, relu=False, bias=True, std=0.05)
        self.strides, self.sampling = tuple(featmap_strides), sampling_ratio

    def modules(self):
        return self.convs + [self.upsample, self.logits]

    def to(self, device):
        for m in self.modules():
            m.to(device)
        return self

    def __call__(self, feats, dets):
        B, D = dets.shape[0], dets.shape[1]
        rois = det_ops.dets_to_rois(dets)                                          # [B*D, 5]
        x = det_ops.roi_align(list(feats[:len(self.strides)]), rois, self.P, [1.0 / s for s in self.strides], self.sampling, True)
        for m in self.convs:
            x = m(x)
        x = self.logits(self.upsample(x))                                          # [B*D, 2P, 2P, nc (padded to 8)]
        masks = det_ops.mask_select(x, dets.view(B * D, 6), self.nc)
        return masks.view(B, D, 2 * self.P, 2 * self.P), dict(rois=rois, logits=x)


@DETECTORS.register_module
class MaskRCNN(FasterRCNN):
    """Faster R-CNN + FCNMaskHead on the final detections: forward -> (dets [B,max_det,6], count [B], masks [B,max_det,28,28]).
    test_cfg.paste_masks (or forward(paste=True)): a fourth output, the masks pasted into the image at `mask_thr_binary` (md_paste_masks):
    [B,max_det,H,ceil(W/32)] int32 bit masks (bit j of word k of a row = pixel 32 k + j)."""

    def __init__(self, backbone, neck, rpn_head, roi_head, mask_head, train_cfg=None, test_cfg=None):
        super().__init__(backbone, neck, rpn_head, roi_head, train_cfg, test_cfg)
        self.mask_head = build_roi_head(mask_head)
        tc = test_cfg or {}
        self.paste = bool(tc.get("paste_masks", False))
        self.mask_thr = float(tc.get("mask_thr_binary", 0.5))

    def to(self, device):
        super().to(device)
        self.mask_head.to(device)
        return self

    def conv_modules(self):
        return super().conv_modules() + self.mask_head.modules()

    def forward(self, images, return_aux=False, paste=None):
        dets, count, aux = FasterRCNN.forward(self, images, return_aux=True)
        masks, aux_mask = self.mask_head(aux["feats"], dets)
        out = (dets, count, masks)
        if self.paste if paste is None else paste:
            img_hw = (images.shape[1], images.shape[2])
            if images.shape[3] == 4:  # stem layout: the border is not image
                img_hw = (img_hw[0] - nn_ops.STEM_PAD_LO - nn_ops.STEM_PAD_HI, img_hw[1] - nn_ops.STEM_PAD_LO - nn_ops.STEM_PAD_HI)
            out = out + (det_ops.paste_masks(masks, dets, img_hw, self.mask_thr, bits=True),)
        if return_aux:
            aux["mask"] = aux_mask
            return out + (aux,)
        return out

    __call__ = forward


class DeformConvModule:
    """ModulatedDeformConv2d (DCNv2) + BatchNorm2d + ReLU -- centernet/src/resnet.py:24-106, used by the CenterNet neck
    (centernet_det.py:123-160): offset conv (3*k*k channels, has_bias) -> md_deform_cols -> 1x1 md_conv2d over the columns.
    The reference zero-initialises the offset conv; here it gets a small random init so that random-weight runs and tests
    exercise the sampling path."""

    def __init__(self, init, cin, cout, k=3, stride=1, pad=1, bn=True, relu=True, bn_eps=1e-5, offset_std=0.02):
        self.cin, self.cout, self.k, self.stride, self.pad, self.relu = cin, cout, k, stride, pad, relu
        self.act = "relu" if relu else None
        self.weight = init.conv(cout, cin, k, None)
        self.bn = init.bn(cout, bn_eps) if bn else None
        self.bias = None
        self.offset_weight = init.conv(3 * k * k, cin, k, offset_std)
        self.offset_bias = torch.from_numpy(init.rng.normal(0.0, 0.1, (3 * k * k,)).astype(np.float32))
        self.packed = self.packed_offset = None

    def to(self, device):
        self.packed = nn_ops.pack_conv(self.weight, bias=self.bias, bn=self.bn, stride=self.stride, pad=self.pad, relu=self.act,
                                       korder=0).to(device)
        self.packed_offset = nn_ops.pack_conv(self.offset_weight, bias=self.offset_bias, stride=self.stride, pad=self.pad,
                                              relu=False).to(device)
        return self

    def __call__(self, x):
        return nn_ops.deform_conv2d(x, self.packed_offset, self.packed)

    def macs(self, ho, wo):
        return ho * wo * (self.cout + 3 * self.k * self.k) * self.cin * self.k * self.k


@DETECTORS.register_module
class CenterNet:
    """centernet/src/centernet_det.py:79-174 (GatherDetectionFeatureCell) + :374-399 (CenterNetDetEval):
    ResNet-18 -> 3 x [3x3 conv + BN + ReLU, Conv2dTranspose 4x4 s2 p1 + BN + ReLU] (512->256->128->64)
    -> hm / wh / reg heads (3x3 conv 64 + ReLU, 1x1 conv; hm bias -2.19, :29-69) -> sigmoid+clip ->
    DetectionDecode (decode.py:123-196).
    The three ModulatedDeformConv2d 3x3 layers (DCNv2, resnet.py:24-106) run as DeformConvModule (published DCNv2
    definition: the primitive's arithmetic is inside un-vendored MindSpore, parity unpinned); `dcn=False` substitutes plain
    3x3 convs.
    The three heads are evaluated as one fused 3x3 conv (64 -> 3*head_conv) and one block-diagonal 1x1
    conv (same arithmetic per output: the off-block weights are exact zeros)."""

    def __init__(self, depth=18, num_classes=80, head_conv=64, K=100, base_width=64, seed=7, train_cfg=None,
                 test_cfg=None, dcn=True):
        init = ParamInit(seed)
        _split_forward_of(self, test_cfg)
        self.backbone = ResNet(depth, base_width=base_width, init=init)
        cin = self.backbone.out_channels[-1]
        self.neck = []
        for cout in (base_width * 4, base_width * 2, base_width):
            self.neck.append(DeformConvModule(init, cin, cout, 3, 1, 1) if dcn else ConvModule(init, cin, cout, 3, 1, 1))
            self.neck.append(DeconvModule(init, cout, cout, 4, 2, 1))
            cin = cout
        self.num_classes, self.head_conv, self.K = num_classes, head_conv, K
        hc = head_conv
        self.heads = {}
        for name, cout, b2 in (("hm", num_classes, -2.19), ("wh", 2, 0.0), ("reg", 2, 0.0)):
            std = None if name == "hm" else 0.001
            c1 = ConvModule(init, cin, hc, 3, 1, 1, bn=False, relu=True, bias=True, std=std,
                            bias_value=None if name == "hm" else 0.0)
            c2 = ConvModule(init, hc, cout, 1, bn=False, relu=False, bias=True, std=std, bias_value=b2)
            self.heads[name] = (c1, c2)
        # fused forms
        self.head1 = ConvModule(init, cin, 3 * hc, 3, 1, 1, bn=False, relu=True, bias=True)
        self.n_out = num_classes + 4
        self.head2 = ConvModule(init, 3 * hc, self.n_out, 1, bn=False, relu=False, bias=True)
        self.fuse_heads()
        self.decode = det_ops.DetectionDecode(reg_offset=True, K=K)

    def fuse_heads(self):
        """(Re)build the fused head convs from self.heads -- call after the per-head weights change (weights.py)."""
        hc = self.head_conv
        self.head1.weight = torch.cat([self.heads[n][0].weight for n in ("hm", "wh", "reg")], 0)
        self.head1.bias = torch.cat([self.heads[n][0].bias for n in ("hm", "wh", "reg")], 0)
        w2 = torch.zeros((self.n_out, 3 * hc, 1, 1))
        b2 = torch.zeros((self.n_out,))
        o = 0
        for i, n in enumerate(("hm", "wh", "reg")):
            c2 = self.heads[n][1]
            w2[o:o + c2.cout, i * hc:(i + 1) * hc] = c2.weight
            b2[o:o + c2.cout] = c2.bias
            o += c2.cout
        self.head2.weight, self.head2.bias = w2, b2

    def conv_modules(self):
        return self.backbone.modules() + self.neck + [self.head1, self.head2]

    def to(self, device):
        for m in self.conv_modules():
            m.to(device)
        _packs_rebuilt(self)
        return self

    def features(self, images):
        x = self.backbone(images)[-1]
        for m in self.neck:
            x = m(x)
        return self.head2(self.head1(x))  # [B, H/4, W/4, roundup8(num_classes + 4)]

    def forward(self, images, return_aux=False):
        head = self.features(images)
        nc = self.num_classes
        # heat map: NHWC bf16 -> NCHW fp32, sigmoid + clip and the 3x3 peak test in one launch (md_heat_peaks)
        heat, hm = det_ops.heat_peaks(head, 0, nc, with_hm=return_aux)
        wh = nn_ops.nhwc_to_nchw_f32(head, nc, 2)
        reg = nn_ops.nhwc_to_nchw_f32(head, nc + 2, 2)
        det, inds, cls = self.decode({"hm": hm, "heat": heat, "wh": wh, "reg": reg}, return_indices=True)
        if return_aux:
            return det, dict(head=head, hm=hm, wh=wh, reg=reg, inds=inds, cls=cls)
        return det

    __call__ = forward


# ----------------------------------------------------------------------------- CenterPoint RPN neck
@NECKS.register_module
class RPN:
    """centerpoint/det3d_ms/models/necks/rpn.py:9-154: per stage [Pad1 + Conv3x3(stride) + BN(eps 1e-3) + ReLU]
    + n x [Conv3x3 + BN + ReLU]; deblocks = Conv2dTranspose(k=s, stride s) or strided Conv2d(k, stride k)
    + BN + ReLU; channel concat -> [B, H/ds, W/ds, sum(us_num_filters)].  The concat is free: every deblock
    writes its channel slice of the output directly (md_conv2d c_off)."""

    def __init__(self, layer_nums=(3, 5, 5), ds_layer_strides=(2, 2, 2), ds_num_filters=(64, 128, 256),
                 us_layer_strides=(0.5, 1, 2), us_num_filters=(128, 128, 128), num_input_features=64, norm_cfg=None,
                 seed=7, **kwargs):
        init = ParamInit(seed)
        eps = (norm_cfg or {}).get("eps", 1e-3)
        self.blocks, self.deblocks = [], []
        cin = num_input_features
        self.up_start = len(layer_nums) - len(us_layer_strides)
        for i, n in enumerate(layer_nums):
            blk = [ConvModule(init, cin, ds_num_filters[i], 3, ds_layer_strides[i], 1, bn_eps=eps)]
            blk += [ConvModule(init, ds_num_filters[i], ds_num_filters[i], 3, 1, 1, bn_eps=eps) for _ in range(n)]
            self.blocks.append(blk)
            cin = ds_num_filters[i]
            if i - self.up_start >= 0:
                s = us_layer_strides[i - self.up_start]
                cout = us_num_filters[i - self.up_start]
                if s > 1:
                    self.deblocks.append(DeconvModule(init, cin, cout, int(s), int(s), 0, bn_eps=eps))
                else:
                    k = int(round(1 / s))
                    self.deblocks.append(ConvModule(init, cin, cout, k, k, 0, bn_eps=eps))
        self.out_channels = sum(us_num_filters)

    def modules(self):
        return [m for b in self.blocks for m in b] + self.deblocks

    def to(self, device):
        for m in self.modules():
            m.to(device)
        return self

    def __call__(self, x):
        out, c_off = None, 0
        for i, blk in enumerate(self.blocks):
            for m in blk:
                x = m(x)
            if i - self.up_start >= 0:
                d = self.deblocks[i - self.up_start]
                if out is None:
                    if isinstance(d, DeconvModule):
                        oh, ow = x.shape[1] * d.stride, x.shape[2] * d.stride
                    else:
                        oh, ow = nn_ops.conv_out_hw(x.shape[1], x.shape[2], d.packed)
                    out = torch.empty((x.shape[0], oh, ow, self.out_channels), dtype=torch.bfloat16, device=x.device)
                if isinstance(d, DeconvModule):
                    d(x, out=out, c_off=c_off)
                else:
                    nn_ops.conv2d(x, d.packed, out=out, c_off=c_off)
                c_off += d.cout
        return out


# ----------------------------------------------------------------------------- YOLOv5 (build-authored; parity unpinned)
SPPF_FUSED = os.environ.get("MD_SPPF_FUSED", "1") != "0"    # A/B knob: 0 = three md_maxpool2d launches + four concat copies per SPPF block
C3_PAIR_FUSED = os.environ.get("MD_C3_PAIR", "1") != "0"   # A/B knob (tools/ab_env_bench.sh): 0 = two md_conv2d launches per C3 bottleneck


def _yconv(init, cin, cout, k=1, s=1, p=None):
    return ConvModule(init, cin, cout, k, s, k // 2 if p is None else p, bn=True, bn_eps=1e-3, act="silu")


class C3:
    """CSP bottleneck with 3 convs: cv3(concat(m(cv1(x)), cv2(x))), m = n x [x (+) cv2_3x3(cv1_1x1(x))]."""

    def __init__(self, init, c1, c2, n=1, shortcut=True):
        c_ = c2 // 2
        self.cv1, self.cv2, self.cv3 = _yconv(init, c1, c_), _yconv(init, c1, c_), _yconv(init, 2 * c_, c2)
        self.m = [(_yconv(init, c_, c_, 1), _yconv(init, c_, c_, 3)) for _ in range(n)]
        self.shortcut, self.c_ = shortcut, c_
        self._cv12 = None
        self._pairs = None   # md_c3_pair packs of self.m (False: this width is not fused)

    def modules(self):
        return [self.cv1, self.cv2, self.cv3] + [m for pair in self.m for m in pair]

    def __call__(self, x, x_c_off=None, out=None, c_off=0):
        """No copies: one launch computes [cv1(x) | cv2(x)] into the concat buffer; every bottleneck then updates channels [0, c_).
        With 64 or 128 channels a bottleneck is ONE md_c3_pair launch (its 1x1 output stays in LDS; bit-identical): it cannot run in
        place (a tile's halo pixels are other tiles' outputs), so the blocks alternate between two concat buffers and the last one
        carries the cv2(x) half along when it ends in the second.  Other widths: two launches per bottleneck, in place (the 1x1 reads the
        slice, the 3x3 adds the slice as residual and writes it back: each element is read and written by the same thread)."""
        n, h, w, _ = x.shape
        if self._cv12 is None:
            self._cv12 = merged_conv([self.cv1, self.cv2], x.device)
        cat = nn_ops.conv2d(x, self._cv12, x_c_off=x_c_off)
        if self._pairs is None:
            pk = [nn_ops.pack_c3_pair(a.packed, b.packed) for a, b in self.m] if C3_PAIR_FUSED else [None]
            self._pairs = pk if all(p is not None for p in pk) else False
        if self._pairs:
            bufs = (cat, torch.empty_like(cat))   # block i reads bufs[i % 2], writes bufs[(i + 1) % 2]; cv2(x) sits in bufs[0]
            npairs = len(self._pairs)
            for i, pk in enumerate(self._pairs):
                # an odd chain ends in bufs[1]: its last block (which reads bufs[0]) carries the cv2(x) half along -- cv3 reads one buffer
                nn_ops.c3_pair(bufs[i % 2], pk, bufs[(i + 1) % 2], 0, 0, self.shortcut, pass_through=(i + 1 == npairs and npairs % 2 == 1))
            return self.cv3(bufs[npairs % 2], out=out, c_off=c_off)
        for a, b in self.m:
            t = a(cat, x_c_off=0)
            if self.shortcut:
                b(t, residual=cat, res_c_off=0, out=cat, c_off=0)     # SiLU first, then the shortcut add (md_conv2d relu code 2)
            else:
                b(t, out=cat, c_off=0)
        return self.cv3(cat, out=out, c_off=c_off)


class SPPF:
    def __init__(self, init, c1, c2, k=5):
        c_ = c1 // 2
        self.cv1, self.cv2, self.k, self.c_ = _yconv(init, c1, c_), _yconv(init, 4 * c_, c2), k, c_

    def modules(self):
        return [self.cv1, self.cv2]

    def __call__(self, x, out=None, c_off=0):
        n, h, w, _ = x.shape
        cat = torch.empty((n, h, w, 4 * self.c_), dtype=torch.bfloat16, device=x.device)
        if SPPF_FUSED and nn_ops.sppf_pool_fits(h, w, self.c_):
            # cv1 writes the first slice of the concat buffer; ONE launch fills the other three (md_sppf_pool; bit-identical)
            self.cv1(x, out=cat, c_off=0)
            nn_ops.sppf_pool(cat, self.c_, self.k)
            return self.cv2(cat, out=out, c_off=c_off)
        y = self.cv1(x)
        nn_ops.concat_copy(y, cat, 0)
        for i in range(1, 4):
            y = nn_ops.maxpool2d(y, self.k, 1, self.k // 2, zero_pad=False)
            nn_ops.concat_copy(y, cat, i * self.c_)
        return self.cv2(cat, out=out, c_off=c_off)


@DETECTORS.register_module
class YOLOv5:
    """YOLOv5 (Ultralytics v6/v7 layout): CSPDarknet backbone + PANet head + Detect, single-label decode and
    class-aware NMS (conf 0.25 / IoU 0.45 / max_det 300 by default).  depth_multiple/width_multiple 0.33/0.5 = yolov5s."""
    ANCHORS = ((10, 13, 16, 30, 33, 23), (30, 61, 62, 45, 59, 119), (116, 90, 156, 198, 373, 326))

    def __init__(self, depth_multiple=0.33, width_multiple=0.5, num_classes=80, conf_thres=0.25, iou_thres=0.45,
                 max_det=300, nms_pre=4096, seed=7, train_cfg=None, test_cfg=None):
        init = ParamInit(seed)
        _split_forward_of(self, test_cfg)
        ch = lambda c: max(8, int(math.ceil(c * width_multiple / 8) * 8))
        d = lambda n: max(1, round(n * depth_multiple))
        c64, c128, c256, c512, c1024 = ch(64), ch(128), ch(256), ch(512), ch(1024)
        self.b0 = _yconv(init, 3, c64, 6, 2, 2)
        self.b1 = _yconv(init, c64, c128, 3, 2)
        self.b2 = C3(init, c128, c128, d(3))
        self.b3 = _yconv(init, c128, c256, 3, 2)
        self.b4 = C3(init, c256, c256, d(6))
        self.b5 = _yconv(init, c256, c512, 3, 2)
        self.b6 = C3(init, c512, c512, d(9))
        self.b7 = _yconv(init, c512, c1024, 3, 2)
        self.b8 = C3(init, c1024, c1024, d(3))
        self.b9 = SPPF(init, c1024, c1024)
        self.h10 = _yconv(init, c1024, c512, 1)
        self.h13 = C3(init, 2 * c512, c512, d(3), False)
        self.h14 = _yconv(init, c512, c256, 1)
        self.h17 = C3(init, 2 * c256, c256, d(3), False)
        self.h18 = _yconv(init, c256, c256, 3, 2)
        self.h20 = C3(init, 2 * c256, c512, d(3), False)
        self.h21 = _yconv(init, c512, c512, 3, 2)
        self.h23 = C3(init, 2 * c512, c1024, d(3), False)
        self.nc, self.na = num_classes, 3
        no = self.na * (5 + num_classes)
        self.detect = [ConvModule(init, c, no, 1, bn=False, relu=False, bias=True, std=0.02) for c in (c256, c512, c1024)]
        self.strides = (8, 16, 32)
        self.conf_thres, self.iou_thres, self.max_det, self.nms_pre = conf_thres, iou_thres, max_det, nms_pre
        self.prefix_status = det_ops.PrefixStatus()   # sticky flags of the top-nms_pre cut (see StandardRoIHead)
        self.c = (c256, c512)
        self._seg = {}

    def conv_modules(self):
        out = [self.b0, self.b1, self.b3, self.b5, self.b7, self.h10, self.h14, self.h18, self.h21] + self.detect
        for blk in (self.b2, self.b4, self.b6, self.b8, self.b9, self.h13, self.h17, self.h20, self.h23):
            out += blk.modules()
        return out

    def to(self, device):
        for m in self.conv_modules():
            m.to(device)
        for blk in (self.b2, self.b4, self.b6, self.b8, self.h13, self.h17, self.h20, self.h23):
            blk._cv12 = blk._pairs = None     # merged / fused packs are derived copies: rebuilt from the (possibly re-loaded) weights on next use
        _packs_rebuilt(self)
        # the 3-channel stride-2 stem conv on the 4-channel stem layout (md_stem_conv) when the batch arrives in it
        self.stem = nn_ops.pack_stem_conv(self.b0.weight, bn=self.b0.bn, bias=self.b0.bias, act=self.b0.act)
        if self.stem is not None:
            self.stem.to(device)
        return self

    def features(self, x):
        """x: [N,H,W,8] bf16 NHWC (3 real channels), or the batch in the stem layout ([N,H+16,W+16,4], nn_ops.to_stem_layout)."""
        if x.shape[3] == 4:
            if self.stem is None:
                raise nn_ops._lib.MindDetHipError("stem-layout input needs a 6x6 / 3x3 stride-2 stem conv with 32 or 64 output channels")
            x = nn_ops.stem_conv(x, self.stem)
        else:
            x = self.b0(x)
        x = self.b3(self.b2(self.b1(x)))
        # p3 / p4 are produced straight into the second half of the PAN concat buffers that consume them later
        # ([upsampled top | skip]); the next backbone conv reads them as a channel slice
        n, h, w, _ = x.shape
        c = self.b4.cv3.cout
        cat17 = torch.empty((n, h, w, 2 * c), dtype=torch.bfloat16, device=x.device)
        self.b4(x, out=cat17, c_off=c)
        x = self.b5(cat17, x_c_off=c)
        n, h, w, _ = x.shape
        c = self.b6.cv3.cout
        cat13 = torch.empty((n, h, w, 2 * c), dtype=torch.bfloat16, device=x.device)
        self.b6(x, out=cat13, c_off=c)
        x = self.b9(self.b8(self.b7(cat13, x_c_off=c)))
        # h10 / h14 are written straight into the second half of the bottom-up concat buffers that consume them later ([downsampled | top]);
        # the top-down upsample reads them there as a channel slice (r04: two concat copies less)
        n, h, w, _ = x.shape
        c = self.h10.cout
        cat23 = torch.empty((n, h, w, 2 * c), dtype=torch.bfloat16, device=x.device)
        self.h10(x, out=cat23, c_off=c)
        nn_ops.upsample2x(cat23, cat13, 0, src_c0=c, width=c)
        n, h, w, _ = cat13.shape
        c = self.h14.cout
        cat20 = torch.empty((n, h, w, 2 * c), dtype=torch.bfloat16, device=x.device)
        self.h14(self.h13(cat13), out=cat20, c_off=c)
        nn_ops.upsample2x(cat20, cat17, 0, src_c0=c, width=c)
        o3 = self.h17(cat17)
        self.h18(o3, out=cat20, c_off=0)
        o4 = self.h20(cat20)
        self.h21(o4, out=cat23, c_off=0)
        o5 = self.h23(cat23)
        return [d(o) for d, o in zip(self.detect, (o3, o4, o5))]

    def forward(self, images, return_aux=False):
        heads = self.features(images)
        B = images.shape[0]
        dev = images.device
        total = sum(h.shape[1] * h.shape[2] * self.na for h in heads)
        boxes = torch.empty((B, total, 4), dtype=torch.float32, device=dev)
        scores = torch.empty((B, total), dtype=torch.float32, device=dev)
        labels = torch.empty((B, total), dtype=torch.int32, device=dev)
        off = 0
        for h, s, anc in zip(heads, self.strides, self.ANCHORS):
            det_ops.yolo_decode(h, boxes, scores, labels, self.nc, self.na, s, anc, self.conf_thres, off, total)
            off += h.shape[1] * h.shape[2] * self.na
        if (B, total) not in self._seg:
            self._seg[(B, total)] = torch.arange(0, (B + 1) * total, total, dtype=torch.int32, device=dev)
        sv, si, sc = det_ops.topk_segmented(scores, self._seg[(B, total)], self.nms_pre, max_segment=total)
        sb = det_ops.gather_rows(boxes, si, sc)
        sl = torch.gather(labels, 1, si.long())
        keep, kidx, num = det_ops.nms_aligned(sb, self.iou_thres, mode=det_ops.NMS_MODE_STRICT, count=sc, group=sl,
                                              max_output=self.max_det)
        dets, count = det_ops.pack_detections(sb, sv, sl, kidx, num, self.max_det, sel_cnt=sc, status=self.prefix_status.tensor(B, dev))
        if return_aux:
            return dets, count, dict(heads=heads, boxes=boxes, scores=scores, labels=labels, sel_idx=si, sel_cnt=sc,
                                     sel_boxes=sb, sel_labels=sl, keep=keep)
        return dets, count

    __call__ = forward


# ----------------------------------------------------------------------------- YOLOv8 (BASELINE configs[3]; parity unpinned)
class C2f:
    """Ultralytics C2f: cv2(concat(y0, y1, m1(y1), m2(m1(y1)), ...)) with (y0, y1) = chunk(cv1(x), 2) and m = n x
    [x (+) 3x3(3x3(x))].  cv1 is stored as its two output halves (cv1a, cv1b: the same arithmetic as one 2c-channel 1x1
    conv followed by chunk), so that y0 is written straight into the concat buffer and y1 is a compact tensor."""

    def __init__(self, init, c1, c2, n=1, shortcut=False):
        c = c2 // 2
        self.c, self.n, self.shortcut = c, n, shortcut
        self.cv1a, self.cv1b = _yconv(init, c1, c, 1), _yconv(init, c1, c, 1)
        self.cv2 = _yconv(init, (2 + n) * c, c2, 1)
        self.m = [(_yconv(init, c, c, 3), _yconv(init, c, c, 3)) for _ in range(n)]
        self._cv1 = None

    def modules(self):
        return [self.cv1a, self.cv1b, self.cv2] + [m for pair in self.m for m in pair]

    def __call__(self, x, x_c_off=None, out=None, c_off=0):
        """No copies: one launch writes (y0 | y1) = cv1(x) into channels [0, 2c) of the concat buffer; bottleneck i reads its
        input (and its shortcut) as channels [(1+i)c, (2+i)c) of that buffer and writes [(2+i)c, (3+i)c)."""
        n, h, w, _ = x.shape
        c = self.c
        if self._cv1 is None:
            self._cv1 = merged_conv([self.cv1a, self.cv1b], x.device)
        cat = torch.empty((n, h, w, (2 + self.n) * c), dtype=torch.bfloat16, device=x.device)
        nn_ops.conv2d(x, self._cv1, out=cat, c_off=0, x_c_off=x_c_off)
        for i, (a, b) in enumerate(self.m):
            t = a(cat, x_c_off=(1 + i) * c)
            if self.shortcut:
                b(t, residual=cat, res_c_off=(1 + i) * c, out=cat, c_off=(2 + i) * c)   # SiLU first, then the shortcut add
            else:
                b(t, out=cat, c_off=(2 + i) * c)
        return self.cv2(cat, out=out, c_off=c_off)


@DETECTORS.register_module
class YOLOv8:
    """YOLOv8 (Ultralytics v8.0 yolov8.yaml): CSP backbone with C2f blocks + SPPF, PAN head, anchor-free Detect with DFL
    (reg_max 16); decode + class-aware NMS (conf 0.25 / IoU 0.7 / max_det 300).  depth / width / max_channels 1.0 / 1.0 / 512
    = yolov8l."""

    def __init__(self, depth_multiple=1.0, width_multiple=1.0, max_channels=512, num_classes=80, reg_max=16, conf_thres=0.25,
                 iou_thres=0.7, max_det=300, nms_pre=4096, seed=7, train_cfg=None, test_cfg=None):
        init = ParamInit(seed)
        _split_forward_of(self, test_cfg)
        ch = lambda c: max(8, int(math.ceil(min(c, max_channels) * width_multiple / 8) * 8))
        d = lambda n: max(1, round(n * depth_multiple))
        c64, c128, c256, c512, c1024 = ch(64), ch(128), ch(256), ch(512), ch(1024)
        self.b0 = _yconv(init, 3, c64, 3, 2)
        self.b1 = _yconv(init, c64, c128, 3, 2)
        self.b2 = C2f(init, c128, c128, d(3), True)
        self.b3 = _yconv(init, c128, c256, 3, 2)
        self.b4 = C2f(init, c256, c256, d(6), True)
        self.b5 = _yconv(init, c256, c512, 3, 2)
        self.b6 = C2f(init, c512, c512, d(6), True)
        self.b7 = _yconv(init, c512, c1024, 3, 2)
        self.b8 = C2f(init, c1024, c1024, d(3), True)
        self.b9 = SPPF(init, c1024, c1024)
        self.h12 = C2f(init, c1024 + c512, c512, d(3))
        self.h15 = C2f(init, c512 + c256, c256, d(3))
        self.h16 = _yconv(init, c256, c256, 3, 2)
        self.h18 = C2f(init, c256 + c512, c512, d(3))
        self.h19 = _yconv(init, c512, c512, 3, 2)
        self.h21 = C2f(init, c512 + c1024, c1024, d(3))
        self.nc, self.reg_max = num_classes, reg_max
        c2 = max(16, c256 // 4, 4 * reg_max)
        c3 = max(c256, min(num_classes, 100))
        self.head_c = (4 * reg_max + num_classes + 7) // 8 * 8
        self.box, self.cls = [], []
        for c in (c256, c512, c1024):
            self.box.append([_yconv(init, c, c2, 3), _yconv(init, c2, c2, 3),
                             ConvModule(init, c2, 4 * reg_max, 1, bn=False, relu=False, bias=True, std=0.05, bias_value=1.0)])
            self.cls.append([_yconv(init, c, c3, 3), _yconv(init, c3, c3, 3),
                             ConvModule(init, c3, num_classes, 1, bn=False, relu=False, bias=True, std=0.05)])
        self.strides = (8, 16, 32)
        self.conf_thres, self.iou_thres, self.max_det, self.nms_pre = conf_thres, iou_thres, max_det, nms_pre
        self.prefix_status = det_ops.PrefixStatus()   # sticky flags of the top-nms_pre cut (see StandardRoIHead)
        self._seg = {}
        self._stem2 = {}
        _packs_rebuilt(self)

    def conv_modules(self):
        out = [self.b0, self.b1, self.b3, self.b5, self.b7, self.h16, self.h19]
        for blk in (self.b2, self.b4, self.b6, self.b8, self.b9, self.h12, self.h15, self.h18, self.h21):
            out += blk.modules()
        for br in self.box + self.cls:
            out += br
        return out

    def to(self, device):
        for m in self.conv_modules():
            m.to(device)
        for blk in (self.b2, self.b4, self.b6, self.b8, self.h12, self.h15, self.h18, self.h21):
            blk._cv1 = None      # merged packs are derived copies: rebuilt from the (possibly re-loaded) weights on next use
        self._stem2 = {}
        _packs_rebuilt(self)
        # the 3-channel stride-2 stem conv on the 4-channel stem layout (md_stem_conv) when the batch arrives in it
        self.stem = nn_ops.pack_stem_conv(self.b0.weight, bn=self.b0.bn, bias=self.b0.bias, act=self.b0.act)
        if self.stem is not None:
            self.stem.to(device)
        return self

    def features(self, x):
        """x: [N,H,W,8] bf16 NHWC (3 real channels), or the batch in the stem layout ([N,H+16,W+16,4], nn_ops.to_stem_layout)."""
        if x.shape[3] == 4:
            if self.stem is None:
                raise nn_ops._lib.MindDetHipError("stem-layout input needs a 6x6 / 3x3 stride-2 stem conv with 32 or 64 output channels")
            x = nn_ops.stem_conv(x, self.stem)
        else:
            x = self.b0(x)
        x = self.b3(self.b2(self.b1(x)))
        # p3 / p4 are produced straight into the PAN concat buffers that consume them later ([upsampled top | skip]); the next
        # backbone conv reads them as a channel slice
        n, h, w, _ = x.shape
        c_top = self.h12.cv2.cout
        cat15 = torch.empty((n, h, w, c_top + self.b4.cv2.cout), dtype=torch.bfloat16, device=x.device)
        self.b4(x, out=cat15, c_off=c_top)
        x = self.b5(cat15, x_c_off=c_top)
        n, h, w, _ = x.shape
        c_top = self.b9.cv2.cout
        cat12 = torch.empty((n, h, w, c_top + self.b6.cv2.cout), dtype=torch.bfloat16, device=x.device)
        self.b6(x, out=cat12, c_off=c_top)
        # p5 / h12 are written straight into the second part of the bottom-up concat buffers that consume them later ([downsampled | top]);
        # the top-down upsample reads them there as a channel slice (r04: two concat copies less)
        x = self.b8(self.b7(cat12, x_c_off=c_top))
        n, h, w, _ = x.shape
        c19, c5 = self.h19.cout, self.b9.cv2.cout
        cat21 = torch.empty((n, h, w, c19 + c5), dtype=torch.bfloat16, device=x.device)
        self.b9(x, out=cat21, c_off=c19)
        nn_ops.upsample2x(cat21, cat12, 0, src_c0=c19, width=c5)
        n, h, w, _ = cat12.shape
        c16, c12 = self.h16.cout, self.h12.cv2.cout
        cat18 = torch.empty((n, h, w, c16 + c12), dtype=torch.bfloat16, device=x.device)
        self.h12(cat12, out=cat18, c_off=c16)
        nn_ops.upsample2x(cat18, cat15, 0, src_c0=c16, width=c12)
        o3 = self.h15(cat15)
        self.h16(o3, out=cat18, c_off=0)
        o4 = self.h18(cat18)
        self.h19(o4, out=cat21, c_off=0)
        o5 = self.h21(cat21)
        heads = []
        for f, bx, cl in zip((o3, o4, o5), self.box, self.cls):
            n, h, w, _ = f.shape
            # (the two branch convs write every channel unless the class count needed padding to a multiple of 8: only then a zero fill)
            full = self.head_c == 4 * self.reg_max + cl[2].cout
            head = (torch.empty if full else torch.zeros)((n, h, w, self.head_c), dtype=torch.bfloat16, device=f.device)
            key = id(bx[0])
            if key not in self._stem2:                                      # the two branch stems read f once: one launch
                self._stem2[key] = merged_conv([bx[0], cl[0]], f.device)
            t = nn_ops.conv2d(f, self._stem2[key])
            bx[2](bx[1](t, x_c_off=0), out=head, c_off=0)                   # 4 * reg_max distribution logits
            cl[2](cl[1](t, x_c_off=bx[0].cout), out=head, c_off=4 * self.reg_max)   # class logits
            heads.append(head)
        return heads

    def forward(self, images, return_aux=False):
        heads = self.features(images)
        B, dev = images.shape[0], images.device
        total = sum(h.shape[1] * h.shape[2] for h in heads)
        boxes = torch.empty((B, total, 4), dtype=torch.float32, device=dev)
        scores = torch.empty((B, total), dtype=torch.float32, device=dev)
        labels = torch.empty((B, total), dtype=torch.int32, device=dev)
        off = 0
        for h, s in zip(heads, self.strides):
            det_ops.yolov8_decode(h, boxes, scores, labels, self.nc, self.reg_max, s, self.conf_thres, off, total)
            off += h.shape[1] * h.shape[2]
        if (B, total) not in self._seg:
            self._seg[(B, total)] = torch.arange(0, (B + 1) * total, total, dtype=torch.int32, device=dev)
        sv, si, sc = det_ops.topk_segmented(scores, self._seg[(B, total)], self.nms_pre, max_segment=total)
        sb = det_ops.gather_rows(boxes, si, sc)
        sl = torch.gather(labels, 1, si.long())
        keep, kidx, num = det_ops.nms_aligned(sb, self.iou_thres, mode=det_ops.NMS_MODE_STRICT, count=sc, group=sl, max_output=self.max_det)
        dets, count = det_ops.pack_detections(sb, sv, sl, kidx, num, self.max_det, sel_cnt=sc, status=self.prefix_status.tensor(B, dev))
        if return_aux:
            return dets, count, dict(heads=heads, boxes=boxes, scores=scores, labels=labels, sel_idx=si, sel_cnt=sc, sel_boxes=sb,
                                     sel_labels=sl, keep=keep)
        return dets, count

    __call__ = forward

