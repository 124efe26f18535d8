"""Known answers of tf.image.non_max_suppression, the third-party routine the reference calls
for its NMS (detect_net.py:136; TensorFlow is not vendored in the reference and not installed
here -- README.md:11 implies TensorFlow 2.0.0).  These are the cases of TensorFlow's own unit test
of the op (NonMaxSuppressionOpTest in tensorflow/core/kernels/non_max_suppression_op_test.cc:
boxes [y1, x1, y2, x2], scores, max_output_size, iou_threshold -> selected indices), restated
here as data because the library cannot be run: they pin the restatement of its algorithm in
oracle/ (greedy by descending score, IoU without the +1 term on corner-normalised boxes,
suppression when IoU > threshold, at most max_output_size boxes) to the published behaviour."""
import numpy as np

_THREE_CLUSTERS = [[0, 0, 1, 1], [0, 0.1, 1, 1.1], [0, -0.1, 1, 0.9],
                   [0, 10, 1, 11], [0, 10.1, 1, 11.1], [0, 100, 1, 101]]
_FLIPPED = [[1, 1, 0, 0], [0, 0.1, 1, 1.1], [0, 0.9, 1, -0.1],
            [0, 10, 1, 11], [1, 10.1, 0, 11.1], [1, 101, 0, 100]]
_SCORES = [0.9, 0.75, 0.6, 0.95, 0.5, 0.3]

# name, boxes, scores, max_output_size, iou_threshold, expected selected indices
CASES = [
    ("SelectFromThreeClusters", _THREE_CLUSTERS, _SCORES, 3, 0.5, [3, 0, 5]),
    ("SelectFromThreeClustersFlippedCoordinates", _FLIPPED, _SCORES, 3, 0.5, [3, 0, 5]),
    ("SelectAtMostTwoBoxesFromThreeClusters", _THREE_CLUSTERS, _SCORES, 2, 0.5, [3, 0]),
    ("SelectAtMostThirtyBoxesFromThreeClusters", _THREE_CLUSTERS, _SCORES, 30, 0.5, [3, 0, 5]),
    ("SelectWithNegativeScores", _THREE_CLUSTERS, [s - 10.0 for s in _SCORES], 6, 0.5, [3, 0, 5]),
    ("SelectSingleBox", [[0, 0, 1, 1]], [0.9], 3, 0.5, [0]),
    ("SelectFromTenIdenticalBoxes", [[0, 0, 1, 1]] * 10, [0.9] * 10, 3, 0.5, [0]),
    ("EmptyInput", [], [], 30, 0.5, []),
]


def arrays(case):
    name, boxes, scores, max_out, thr, want = case
    return (np.asarray(boxes, np.float32).reshape(-1, 4), np.asarray(scores, np.float32), max_out, thr,
            np.asarray(want, np.int32))


def as_decoded_tuple(boxes, scores, num_classes=5):
    """The case as the 4-tuple DetectNet.filter_boxes takes (one image): centre and size such that
    centre -/+ size/2 give the case's corners ([y1, x1, y2, x2]; a flipped box has a negative
    size), objectness = the score, every class probability 1, and image dimensions of 1 so that
    the scaling of detect_net.py:133 changes nothing."""
    n = len(scores)
    center = np.zeros((1, n, 2), np.float32)
    wh = np.zeros((1, n, 2), np.float32)
    center[0, :, 0] = (boxes[:, 1] + boxes[:, 3]) / 2       # x
    center[0, :, 1] = (boxes[:, 0] + boxes[:, 2]) / 2       # y
    wh[0, :, 0] = boxes[:, 3] - boxes[:, 1]
    wh[0, :, 1] = boxes[:, 2] - boxes[:, 0]
    obj = scores.reshape(1, n, 1).astype(np.float32)
    # the box score is max over classes of objectness x class probability (detect_net.py:107-109):
    # with every class at 1 it is the objectness itself, also when that is negative
    cls = np.ones((1, n, num_classes), np.float32)
    return center, wh, obj, cls
