# YOLOv8l 640x640 inference (BASELINE.json configs[3]); Ultralytics v8.0 yolov8.yaml scale "l" (depth 1.0, width 1.0,
# max_channels 512).  Absent from the reference (README bullet): standard public definition, parity unpinned.
num_classes = 80
model = dict(type="YOLOv8", depth_multiple=1.0, width_multiple=1.0, max_channels=512, num_classes=num_classes, reg_max=16,
             conf_thres=0.25, iou_thres=0.7, max_det=300, nms_pre=4096)
train_cfg = None
# streams=2: the batch runs as two halves on two HIP streams (graphs.SplitForward: -2.9 % ms/step at the 32-image shard, bit-identical)
test_cfg = dict(max_per_img=300, streams=2)
data = dict(input_hw=(640, 640))
