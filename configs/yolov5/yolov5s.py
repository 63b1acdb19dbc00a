# YOLOv5s 640x640 (BASELINE.json configs[1]): CSPDarknet conv + class-aware NMS, one-stage.
num_classes = 80
model = dict(type="YOLOv5", depth_multiple=0.33, width_multiple=0.5, num_classes=num_classes, conf_thres=0.25,
             iou_thres=0.45, max_det=300, nms_pre=4096)
train_cfg = None
test_cfg = dict(max_per_img=300)
data = dict(input_hw=(640, 640), batch=32)
