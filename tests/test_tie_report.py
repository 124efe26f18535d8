"""The tie report (tests/tie_report.py) on constructed cases, CPU only: differences between two
write_results outputs are traced to the threshold tie behind them, anything else is flagged."""
import numpy as np

import oracle
from tie_report import differences, format_report, iou_plus1


def _row(cx, cy, w, h, obj, cls=0, C=2):
    r = [cx, cy, w, h, obj] + [0.1] * C
    r[5 + cls] = 0.9
    return r


def test_identical_outputs_have_no_differences():
    rows = np.array([[_row(10, 10, 8, 8, 0.9), _row(11, 10, 8, 8, 0.8), _row(60, 60, 5, 5, 0.7, 1)]], np.float32)
    a, _, _ = oracle.n_write_results(rows, 0.5, 2, 0.4)
    assert differences(a, a.copy(), 0.5, 0.4) == []
    b = a.copy()
    b[:, 1:5] *= np.float32(1.0 + 1e-7)            # last-bit noise is not a difference
    assert differences(a, b, 0.5, 0.4) == []


def test_iou_tie_is_traced_and_its_consequence_too():
    # boxes [0,0,1,0] and [1,0,2,0]: IoU(+1) = 1/3 exactly; a third box that only the second one
    # suppresses (it overlaps the second heavily, the first not at all)
    base = [_row(0.5, 0.0, 1, 0, 0.9), _row(1.5, 0.0, 1, 0, 0.8), _row(2.45, 0.0, 1.1, 0, 0.7),
            _row(50, 50, 4, 4, 0.6)]
    rows = np.array([base], np.float32)
    thr = np.float32(1.0 / 3.0)
    a, _, _ = oracle.n_write_results(rows, 0.5, 2, float(thr))            # IoU >= thr: second row goes
    nudged = rows.copy()
    nudged[0, 1, 0] += np.float32(2e-6)                                   # IoU a hair below thr: it stays
    b, _, _ = oracle.n_write_results(nudged, 0.5, 2, float(thr))
    assert len(b) != len(a) or not np.array_equal(a, b)
    d = differences(a, b, 0.5, float(thr), "exact", "nudged")
    assert d, "the two outputs differ"
    assert all(x["reason"] for x in d), format_report(d)
    reasons = {x["reason"] for x in d}
    assert "IoU at the NMS threshold" in reasons
    tie = [x for x in d if x["reason"] == "IoU at the NMS threshold"][0]
    assert tie["side"] == "nudged" and tie["distance"] < 1e-5
    # the third row: kept where the second one is gone, suppressed where it stayed
    assert any(x["reason"].startswith("suppressed on the other side") for x in d) or len(d) == 1
    assert "nudged only" in format_report(d)


def test_confidence_tie_is_traced():
    rows = np.array([[_row(10, 10, 8, 8, 0.9), _row(40, 40, 8, 8, 0.5)]], np.float32)
    a, _, _ = oracle.n_write_results(rows, 0.5, 2, 0.4)                   # 0.5 > 0.5 is false: dropped
    rows2 = rows.copy()
    rows2[0, 1, 4] = np.nextafter(np.float32(0.5), np.float32(1.0))
    b, _, _ = oracle.n_write_results(rows2, 0.5, 2, 0.4)
    d = differences(a, b, 0.5, 0.4)
    assert len(d) == 1 and d[0]["reason"] == "objectness at the confidence threshold" and d[0]["side"] == "B"


def test_a_difference_without_a_tie_is_flagged():
    rows = np.array([[_row(10, 10, 8, 8, 0.9), _row(40, 40, 8, 8, 0.8), _row(70, 70, 8, 8, 0.7)]], np.float32)
    a, _, _ = oracle.n_write_results(rows, 0.5, 2, 0.4)
    b = np.delete(a, 1, axis=0)
    d = differences(a, b, 0.5, 0.4)
    assert len(d) == 1 and d[0]["reason"] is None and "UNEXPLAINED" in format_report(d)
    assert abs(float(iou_plus1([0, 0, 1, 0], [[1, 0, 2, 0]])[0]) - 1.0 / 3.0) < 1e-12


def test_align_pairs_rows_whose_order_flipped():
    """Two outputs with the same detections where two rows of one class swap places (objectness
    equal to the last bits): align() finds the permutation, and refuses a row without counterpart."""
    from tie_report import align
    rs = np.random.RandomState(5)
    n = 40
    a = np.zeros((n, 8), np.float32)
    a[:, 0] = np.repeat([0, 1], n // 2)
    a[:, 7] = np.tile(np.repeat([0, 2], n // 4), 2)
    a[:, 1:3] = rs.uniform(0, 300, (n, 2))
    a[:, 3:5] = a[:, 1:3] + rs.uniform(5, 80, (n, 2))
    for g in range(4):                                   # objectness descending inside each group
        a[g * 10:(g + 1) * 10, 5] = np.sort(rs.uniform(0.5, 1, 10))[::-1]
    a[:, 6] = rs.uniform(0.3, 1, n)
    a[13, 5] = a[12, 5]                                  # a tie in objectness
    b = a * np.float32(1 + 3e-7)
    b[:, [0, 7]] = a[:, [0, 7]]
    b[[12, 13]] = b[[13, 12]]                            # the other side orders the tie the other way
    p = align(a, b)
    assert sorted(p) == list(range(n)) and p[12] == 13 and p[13] == 12
    assert np.abs(b[p] - a).max() < 1e-3
    c = b.copy()
    c[5, 1:5] += 50.0
    import pytest
    with pytest.raises(ValueError):
        align(a, c)
