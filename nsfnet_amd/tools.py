"""Collocation-point sampling helpers with the reference's names and results
(NSFnet/tools.py, identical in ev-NSFnet/), vectorised.

The reference draws every stratum with a Python loop over ``np.random.uniform(size=1)``
(tools.py:40-45) and sorts points with an O(N * 2052) pure-Python double loop
(tools.py:59-83; ~1.5 ms per point, 9 minutes at N = 360k).  These restatements consume
the SAME global numpy RNG stream in the same order, so under ``np.random.seed(s)`` they
return bit-identical samples (tests/test_data_prep.py checks that against vectors produced
by the reference), and the sort is a chunked vectorised distance computation.
"""
import numpy as np


def LHSample(D, bounds, N):
    """Latin-hypercube sample of N points in D dimensions (tools.py:30-57)."""
    result = np.empty([N, D])
    d = 1.0 / N
    j = np.arange(N, dtype=np.float64)
    low, high = j * d, (j + 1.0) * d
    for i in range(D):
        u = np.random.random_sample(N)          # == N consecutive np.random.uniform(size=1) draws
        temp = low + (high - low) * u
        np.random.shuffle(temp)
        result[:, i] = temp
    b = np.array(bounds)
    lower_bounds, upper_bounds = b[:, 0], b[:, 1]
    if np.any(lower_bounds > upper_bounds):
        print('Wrong value bound')
        return None
    np.add(np.multiply(result, (upper_bounds - lower_bounds), out=result), lower_bounds, out=result)
    return result


def distance(p1, p2):
    return float(np.sqrt((p2[0] - p1[0]) ** 2 + (p2[1] - p1[1]) ** 2))


def min_distances(pts1, pts2, chunk=4096):
    """min_j |pts1_i - pts2_j| for every i, evaluated with the reference's formula
    sqrt(dx^2 + dy^2) (tools.py:59-67) in row chunks."""
    pts1 = np.asarray(pts1, dtype=np.float64)
    pts2 = np.asarray(pts2, dtype=np.float64)
    out = np.empty(pts1.shape[0])
    for lo in range(0, pts1.shape[0], chunk):
        p = pts1[lo:lo + chunk]
        dx = pts2[None, :, 0] - p[:, None, 0]
        dy = pts2[None, :, 1] - p[:, None, 1]
        out[lo:lo + chunk] = np.sqrt(dx * dx + dy * dy).min(axis=1)
    return out


def minDistance(pt, pts2):
    return float(min_distances(np.asarray(pt, dtype=np.float64).reshape(1, 2), pts2)[0])


def sort_pts(pts1, pts2, flag_reverse=False):
    """Sort pts1 by distance to the point set pts2 (tools.py:69-83); returns
    (sorted points (N,2), sorted distances (N,1))."""
    minDists = min_distances(pts1, pts2).reshape(1, -1)
    dists_sorted = np.sort(minDists).reshape(-1, 1)
    sort_index = np.argsort(minDists)
    if flag_reverse:
        sort_index = sort_index.reshape(-1, 1)[::-1].reshape(1, -1)
        dists_sorted = dists_sorted[::-1]
    pts1_sorted = np.squeeze(np.asarray(pts1)[sort_index, :])
    return pts1_sorted, dists_sorted
