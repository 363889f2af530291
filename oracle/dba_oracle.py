"""CPU oracle (TEST INFRASTRUCTURE ONLY) of the DTW-barycentre front-end, SURVEY 8(f)-3.

Restates the semantics of the reference's own DBA, ensembles/dtwa.py (performDBA :6-20, approximate_medoid_index
:23-37, squared_DTW :48-78, DBA_update :87-143): squared-difference local cost, the three-way minimum with the
reference's tie order (diagonal, then left when the diagonal loses to it, else top), medoid start, and centre update
= mean of the aligned samples accumulated in back-tracking order.  Pinned bit-for-bit by tests/golden/dba_*.npz,
which oracle/make_dba_golden.py produced by running the reference module itself.
The dynamic programme is swept by anti-diagonals (vectorised); floating-point operations per cell are the
reference's (one subtraction, one square, one addition), so results are bit-identical, not merely close."""
from __future__ import annotations

import numpy as np

INF = float("inf")


def _dp(x, y, want_path):
    """cost (N,M) of aligning x (rows) with y (columns); path codes 0 diag, 1 left, 2 top, -1 origin."""
    n, m = len(x), len(y)
    delta = np.square(np.subtract.outer(x, y))
    cost = np.full((n + 1, m + 1), INF)          # one guard row / column of +inf
    path = np.zeros((n, m), dtype=np.int8) if want_path else None
    cost[1, 1] = delta[0, 0]
    if want_path:
        path[0, 0] = -1
    for d in range(1, n + m - 1):
        i = np.arange(max(0, d - m + 1), min(d, n - 1) + 1)
        j = d - i
        diag, left, top = cost[i, j], cost[i + 1, j], cost[i, j + 1]
        take_diag_branch = diag <= left
        res = np.where(take_diag_branch, np.where(diag <= top, diag, top), np.where(left <= top, left, top))
        cost[i + 1, j + 1] = res + delta[i, j]
        if want_path:
            path[i, j] = np.where(take_diag_branch, np.where(diag <= top, 0, 2), np.where(left <= top, 1, 2))
    return cost[1:, 1:], path


def squared_dtw(x, y):
    """dtwa.py:48-78"""
    return _dp(np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64), False)[0][-1, -1]


def medoid_index(series):
    """dtwa.py:23-37 for at most 50 series (beyond that the reference draws a random subset)."""
    assert len(series) <= 50
    best, idx = None, -1
    for a, cand in enumerate(series):
        ss = 0
        for t in series:
            ss = ss + squared_dtw(cand, t)
        if idx == -1 or ss < best:
            best, idx = ss, a
    return idx


def dba_update(center, series):
    """dtwa.py:87-143"""
    n = len(center)
    acc = np.zeros(n)
    cnt = np.zeros(n, dtype=np.int64)
    for s in series:
        _, path = _dp(center, s, True)
        i, j = n - 1, len(s) - 1
        while path[i, j] != -1:
            acc[i] += s[j]
            cnt[i] += 1
            code = path[i, j]
            if code == 0:
                i, j = i - 1, j - 1
            elif code == 1:
                j -= 1
            else:
                i -= 1
        assert i == 0 and j == 0
        acc[0] += s[0]
        cnt[0] += 1
    return acc / cnt


def perform_dba(series, n_iterations=10):
    """dtwa.py:6-20"""
    series = [np.asarray(s, dtype=np.float64) for s in series]
    center = series[medoid_index(series)]
    for _ in range(n_iterations):
        center = dba_update(center, series)
    return center
