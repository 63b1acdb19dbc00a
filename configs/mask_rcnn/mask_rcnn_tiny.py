# Same graph as mask_rcnn_r101_fpn.py at a size the CPU oracle finishes in seconds.
num_classes = 5
model = dict(
    type="MaskRCNN",
    backbone=dict(type="ResNet", depth=50, base_width=16, layers=[1, 1, 1, 1]),
    neck=dict(type="FPN", out_channels=32, num_outs=5),
    rpn_head=dict(type="RPNHead", in_channels=32, feat_channels=32, strides=(4, 8, 16, 32, 64), scale=4.0,
                  ratios=(0.5, 1.0, 2.0), nms_pre=200, max_per_img=100, nms_thr=0.7),
    roi_head=dict(type="StandardRoIHead", in_channels=32, fc_channels=64, num_classes=num_classes, roi_size=7,
                  sampling_ratio=2, featmap_strides=(4, 8, 16, 32), score_thr=0.05, nms_thr=0.5, max_per_img=20,
                  nms_pre=256),
    mask_head=dict(type="FCNMaskHead", in_channels=32, conv_channels=32, num_convs=4, num_classes=num_classes,
                   roi_size=14, sampling_ratio=2, featmap_strides=(4, 8, 16, 32)),
)
train_cfg = None
test_cfg = dict(max_per_img=20)
data = dict(input_hw=(128, 192))
