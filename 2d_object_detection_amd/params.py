"""Post-processing configuration with the names the reference's params.py uses
(params.py:8-22,27-33).  The DetectNet functions read these attributes at call
time, so assigning to them (``params.object_conf = 0.3``) takes effect on the next call,
exactly as with the reference module."""
import numpy as np

grid_stride = 32          # params.py:8
object_conf = .6          # params.py:9
nms_conf = .5             # params.py:10
coord_loss_weight = 1     # params.py:11
obj_loss_weight = 5       # params.py:12
noobj_loss_weight = 1     # params.py:13
num_anchors = 5           # params.py:14
max_boxes = 20            # params.py:15
scaled_height = 416       # params.py:16
scaled_width = 416        # params.py:17

# anchors of the 13 x 13 grid, (w, h) in grid cells             params.py:20-22
YOLO_ANCHORS = np.array(
    ((0.57273, 0.677385), (1.87446, 2.06253), (3.33843, 5.47434),
     (7.88282, 3.52778), (9.77052, 9.16828)))

im_height = 1280          # params.py:27  rows of the original frame
im_width = 1920           # params.py:28  columns of the original frame
grid_height = int(scaled_height // grid_stride)   # params.py:29
grid_width = int(scaled_width // grid_stride)     # params.py:30
num_classes = 5           # params.py:31
pred_vec_len = 5 + num_classes                    # params.py:32
true_vec_len = 5                                  # params.py:33

# IoU above which a prediction counts as having found its object (the literal at train.py:167)
detection_iou = 0.6

# set True to reproduce the reference's "max score in batch" print (detect_net.py:110);
# it costs a device-to-host synchronisation inside filter_boxes, so it is off by default.
print_max_score = False
