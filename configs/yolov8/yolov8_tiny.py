# The YOLOv8 graph at a size the CPU oracle finishes in seconds.
num_classes = 5
model = dict(type="YOLOv8", depth_multiple=0.33, width_multiple=0.25, max_channels=1024, num_classes=num_classes, reg_max=16,
             conf_thres=0.25, iou_thres=0.7, max_det=50, nms_pre=512)
train_cfg = None
test_cfg = dict(max_per_img=50)
data = dict(input_hw=(128, 160))
