"""Keep-set differences between two runs of write_results on (nearly) the same decoded boxes,
each traced to the threshold tie that causes it -- or reported as unexplained.

Two evaluations of the path that differ only in the last bits of exp() (the library's fixed
operation sequence against libm's expf) produce boxes and scores that agree to ~1e-7 relative.
Their kept sets can then differ only where a decision sits at a threshold:
  * an objectness within TIE of the confidence threshold (the row passes the filter on one side),
  * a pair whose IoU is within TIE of the NMS threshold (the later box survives on one side),
  * or a consequence of such a row: it is present on one side only, and there it clearly
    suppresses (IoU well above the threshold) a row the other side keeps.
`differences` matches the rows of the two outputs, lists every row that is on one side only, and
says which of these explains it; anything else is an unexplained difference, which the tests
treat as a failure.
"""
import numpy as np

TIE = 1e-5      # relative closeness to a threshold that counts as a tie
MATCH = 2e-5    # rows of the two sides are the same detection when all values agree this closely


def iou_plus1(a, b):
    """IoU, pixel-inclusive (+1) form, of box a (4,) with boxes b (M, 4), in float64."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64).reshape(-1, 4)
    iw = np.clip(np.minimum(a[2], b[:, 2]) - np.maximum(a[0], b[:, 0]) + 1.0, 0.0, None)
    ih = np.clip(np.minimum(a[3], b[:, 3]) - np.maximum(a[1], b[:, 1]) + 1.0, 0.0, None)
    inter = iw * ih
    area_a = (a[2] - a[0] + 1.0) * (a[3] - a[1] + 1.0)
    area_b = (b[:, 2] - b[:, 0] + 1.0) * (b[:, 3] - b[:, 1] + 1.0)
    return inter / (area_a + area_b - inter)


def _match(a, b, pairs=None):
    """Indices of the rows of a without a counterpart in b and vice versa (rows (n, 8) of one
    image and class, both sorted by objectness descending as write_results emits them).  Only
    rows of b whose objectness lies within MATCH of the row's are candidates: a window found by
    bisection, since b is sorted.  `pairs` (a list) receives the (row of a, row of b) matches."""
    used = np.zeros(len(b), bool)
    only_a = []
    if pairs is None:
        pairs = []
    neg_obj_b = -b[:, 5].astype(np.float64)          # ascending
    for i, r in enumerate(a):
        lo = int(np.searchsorted(neg_obj_b, -float(r[5]) - MATCH, side="left"))
        hi = int(np.searchsorted(neg_obj_b, -float(r[5]) + MATCH, side="right"))
        found = -1
        if hi > lo:
            scale = max(1.0, float(np.abs(r[1:5]).max()))
            close = (~used[lo:hi]) & (np.abs(b[lo:hi, 1:5] - r[1:5]).max(1) <= MATCH * scale)
            j = np.flatnonzero(close)
            if len(j):
                found = lo + int(j[0])
        if found >= 0:
            used[found] = True
            pairs.append((i, found))
        else:
            only_a.append(i)
    return only_a, list(np.flatnonzero(~used))


def align(rows_a, rows_b):
    """Permutation p with rows_b[p[i]] the counterpart of rows_a[i], for two outputs that hold the
    same detections (no one-sided rows): inside an (image, class) group the order may differ where
    two objectness values are equal to the last bits.  Raises ValueError when a row has no
    counterpart."""
    if rows_a.shape != rows_b.shape:
        raise ValueError("different numbers of rows")
    perm = np.arange(len(rows_a))
    key_a = rows_a[:, 0].astype(np.int64) * 4096 + rows_a[:, 7].astype(np.int64)
    key_b = rows_b[:, 0].astype(np.int64) * 4096 + rows_b[:, 7].astype(np.int64)
    for key in np.union1d(key_a, key_b):
        ia, ib = np.flatnonzero(key_a == key), np.flatnonzero(key_b == key)
        if len(ia) != len(ib):
            raise ValueError("group %d holds %d and %d rows" % (key, len(ia), len(ib)))
        a, b = rows_a[ia], rows_b[ib]
        scale = np.maximum(1.0, np.abs(a[:, 1:5]).max(1))
        if (np.abs(a[:, 1:5] - b[:, 1:5]).max(1) <= MATCH * scale).all():
            perm[ia] = ib
            continue
        pairs = []
        only_a, only_b = _match(a, b, pairs)
        if only_a or only_b:
            raise ValueError("group %d: rows without a counterpart" % key)
        for i, j in pairs:
            perm[ia[i]] = ib[j]
    return perm


def differences(rows_a, rows_b, conf, thr, name_a="A", name_b="B"):
    """Every row kept on one side only, with its explanation.

    rows_*: (D, 8) [image, x1, y1, x2, y2, objectness, class score, class id].
    Returns a list of dicts {side, image, class, box, objectness, reason, partner, iou,
    distance} (reason None = unexplained)."""
    out = []
    key_a = rows_a[:, 0].astype(np.int64) * 4096 + rows_a[:, 7].astype(np.int64)
    key_b = rows_b[:, 0].astype(np.int64) * 4096 + rows_b[:, 7].astype(np.int64)
    for key in np.union1d(key_a, key_b):
        a, b = rows_a[key_a == key], rows_b[key_b == key]
        if len(a) == len(b):
            # same rows in the same order up to last-bit noise: nothing to match one by one
            scale = np.maximum(1.0, np.abs(a[:, 1:5]).max(1))
            if (np.abs(a[:, 1:5] - b[:, 1:5]).max(1) <= MATCH * scale).all() and \
                    (np.abs(a[:, 5] - b[:, 5]) <= MATCH).all():
                continue
        only_a, only_b = _match(a, b)
        if not only_a and not only_b:
            continue
        # the rows that are on one side only, by descending objectness: a row is explained by a
        # tie of its own, or by an (already listed) one-sided row of the other side above it
        pending = [(float(a[i, 5]), name_a, a[i], b) for i in only_a] + \
                  [(float(b[i, 5]), name_b, b[i], a) for i in only_b]
        pending.sort(key=lambda t: -t[0])
        one_sided = {name_a: [], name_b: []}
        for obj, side, row, other in pending:
            rec = {"side": side, "image": int(row[0]), "class": int(row[7]),
                   "box": [float(v) for v in row[1:5]], "objectness": obj,
                   "reason": None, "partner": None, "iou": None, "distance": None}
            other_side = name_b if side == name_a else name_a
            if abs(obj - conf) <= TIE * max(conf, 1e-30):
                rec.update(reason="objectness at the confidence threshold", distance=abs(obj - conf))
            else:
                # a row of the other side's output, ranked at or above this one, whose IoU with
                # it sits at the threshold: it suppressed this row over there and not here
                above = other[other[:, 5] >= obj - MATCH]
                if len(above):
                    iou = iou_plus1(row[1:5], above[:, 1:5])
                    k = int(np.argmin(np.abs(iou - thr)))
                    if abs(iou[k] - thr) <= TIE * max(thr, 1e-30):
                        rec.update(reason="IoU at the NMS threshold", partner=[float(v) for v in above[k, 1:5]],
                                   iou=float(iou[k]), distance=float(abs(iou[k] - thr)))
                if rec["reason"] is None:
                    # a one-sided row of the other side (a tie's consequence) that suppresses this
                    # row there beyond doubt
                    for prow in one_sided[other_side]:
                        if prow[5] >= obj - MATCH:
                            iou = float(iou_plus1(row[1:5], prow[1:5])[0])
                            if iou >= thr:
                                rec.update(reason="suppressed on the other side by a row that a tie put there",
                                           partner=[float(v) for v in prow[1:5]], iou=iou,
                                           distance=float(iou - thr))
                                break
            one_sided[side].append(row)
            out.append(rec)
    return out


def format_report(diffs):
    lines = []
    for d in diffs:
        lines.append("%s only: image %d class %d box (%.3f, %.3f, %.3f, %.3f) objectness %.7f -- %s%s"
                     % (d["side"], d["image"], d["class"], *d["box"], d["objectness"],
                        d["reason"] or "UNEXPLAINED",
                        "" if d["distance"] is None else " (|value - threshold| = %.3g)" % d["distance"]))
    return "\n".join(lines)
