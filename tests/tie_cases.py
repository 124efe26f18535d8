"""Constructed inputs with equal objectness values (shared by the GPU and the emulation tests)."""
import numpy as np


def few_ties_rows(B=3, N=900, C=3, seed=33):
    """Rows with distinct objectness values except for a few groups of candidates of one class that
    share a value: image 0 holds groups of 2, 2, 3 and 5 (two of them with adjacent float values, so
    their ranks touch) plus a pair split over two classes (no tie inside a segment); image 1 holds
    35 pairs; image 2 none."""
    rs = np.random.RandomState(seed)
    rows = np.zeros((B, N, 5 + C), np.float32)
    rows[:, :, 0], rows[:, :, 1] = rs.uniform(0, 400, (B, N)), rs.uniform(0, 400, (B, N))
    rows[:, :, 2], rows[:, :, 3] = rs.uniform(8, 90, (B, N)), rs.uniform(8, 90, (B, N))
    for b in range(B):
        rows[b, :, 4] = rs.permutation(np.linspace(0.55, 0.99, N).astype(np.float32))
    cls = rs.randint(0, C, (B, N))
    v = np.float32(0.7512345)
    groups = [([5, 17], 0, v), ([N // 2 - 50, 3], 0, np.nextafter(v, np.float32(1))),
              ([100, N - 1, N // 2], 1, np.float32(0.9)), ([7, 8, 9, N - 100, N - 99], 2, np.float32(0.6000001))]
    for idx, c, val in groups:
        rows[0, idx, 4] = val
        cls[0, idx] = c
    rows[0, [50, 51], 4] = np.float32(0.8123)
    cls[0, 50], cls[0, 51] = 0, 1
    pairs = rs.permutation(N)[:70].reshape(35, 2)
    for a, b2 in pairs:
        rows[1, b2, 4] = rows[1, a, 4]
        cls[1, b2] = cls[1, a]
    rows[np.arange(B)[:, None], np.arange(N)[None, :], 5 + cls] = 0.9
    return np.ascontiguousarray(rows)
