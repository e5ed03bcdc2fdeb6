"""TEST INFRASTRUCTURE -- banded Levenshtein distance between two sequences (numpy, one anti-diagonal-free row at a time):
how far a polished contig is from the simulated truth.  Exact when the true distance is below `band`."""
import numpy as np


def banded_edit_distance(a, b, band=256):
    a = np.frombuffer(a.encode() if isinstance(a, str) else a, np.uint8)
    b = np.frombuffer(b.encode() if isinstance(b, str) else b, np.uint8)
    n, m = len(a), len(b)
    if abs(n - m) >= band:
        return None
    big = 1 << 30
    width = 2 * band + 1
    # row i holds D[i, i - band .. i + band]
    prev = np.full(width, big, np.int64)
    j = np.arange(-band, band + 1)
    prev[band:] = np.arange(0, band + 1)
    prev[(j > m) | (j < 0)] = big
    for i in range(1, n + 1):
        cols = i + j
        ok = (cols >= 0) & (cols <= m)
        sub = np.full(width, big, np.int64)
        inside = ok & (cols >= 1)
        idx = np.clip(cols - 1, 0, max(0, m - 1))
        sub[inside] = prev[inside] + (b[idx[inside]] != a[i - 1]) if m else big
        dele = np.full(width, big, np.int64)
        dele[:-1] = prev[1:] + 1                         # D[i-1, col] + 1: one slot to the right in the previous row
        cur = np.minimum(sub, dele)
        cur[~ok] = big
        cur[cols == 0] = i
        # insertions propagate left to right within the row: D[i, c] = min(D[i, c], D[i, c-1] + 1)
        run = cur - np.arange(width)
        run = np.minimum.accumulate(run)
        cur = np.minimum(cur, run + np.arange(width))
        cur[~ok] = big
        prev = cur
    k = m - n + band
    return int(prev[k]) if 0 <= k < width and prev[k] < big else None
