"""MI355X-native YOLO detection post-processing (decode -> confidence filter -> NMS).

Drop-in call surface of lbadams2/2D_object_detection's hot path:

* ``detect_net.DetectNet`` -- ``predict_transform`` / ``filter_boxes`` / ``get_anchors``
  with the reference's signatures (detect_net.py:75-183), configured through ``params``.
* ``util`` -- the YOLOv3 three-stride functions ``predict_transform`` / ``write_results`` /
  ``bbox_iou`` and the fused ``YoloV3Post`` engine.
* ``create_dataset`` -- the ground-truth grid codec ``create_true_box_grid`` /
  ``gather_grid_boxes`` (create_dataset.py:47-115, :278-329).
* ``train`` -- ``loss_custom`` / ``loss_keras`` (one fused loss + gradient kernel),
  ``get_metrics``, ``run_validation`` (train.py:110-353).
* ``distributed`` -- batch sharding over the GPUs of a node and the variable-length
  all-gather of detection rows.

The directory name starts with a digit, so import it with
``importlib.import_module("2d_object_detection_amd")``.
"""
from . import params  # noqa: F401
from . import _lib  # noqa: F401
from . import synthetic  # noqa: F401
from . import detect_net  # noqa: F401
from . import util  # noqa: F401
from . import create_dataset  # noqa: F401
from . import train  # noqa: F401
from . import distributed  # noqa: F401
from .detect_net import DetectNet  # noqa: F401
from .util import YoloV3Post, bbox_iou, predict_transform, write_results  # noqa: F401
