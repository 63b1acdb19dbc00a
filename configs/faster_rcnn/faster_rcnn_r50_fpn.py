# Faster R-CNN ResNet-50 FPN, inference (BASELINE.json configs[2]); python-file config in the
# reference's Config.fromfile style (top-level keys model / train_cfg / test_cfg / data).
num_classes = 80
model = dict(
    type="FasterRCNN",
    backbone=dict(type="ResNet", depth=50),
    neck=dict(type="FPN", out_channels=256, num_outs=5),
    rpn_head=dict(type="RPNHead", in_channels=256, feat_channels=256, strides=(4, 8, 16, 32, 64), scale=8.0,
                  ratios=(0.5, 1.0, 2.0), nms_pre=1000, max_per_img=1000, nms_thr=0.7),
    roi_head=dict(type="StandardRoIHead", in_channels=256, fc_channels=1024, num_classes=num_classes, roi_size=7,
                  sampling_ratio=2, featmap_strides=(4, 8, 16, 32), score_thr=0.05, nms_thr=0.5, max_per_img=100,
                  nms_pre=2048),
)
train_cfg = None
# streams=2 (r04): the batch runs as two halves on two HIP streams (graphs.SplitForward, bit-identical): +1.0...2.6 % images/s on four boxes in
# r03 (`two_streams` of the bench line); bench.py measures the roofline block in a serialized one-stream pass after the timed region
test_cfg = dict(max_per_img=100, streams=2)
data = dict(img_scale=(1333, 800), pad_divisor=32, input_hw=(800, 1344),
            mean=[0.408, 0.447, 0.470], std=[0.289, 0.274, 0.278])
