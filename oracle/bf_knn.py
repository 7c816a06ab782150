"""ORACLE (test infrastructure only -- imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg).

CPU restatement of the visual-odometry matcher of GISNav's TwistNode:

    cv2.BFMatcher(crossCheck=False).knnMatch(desc_qry, desc_ref, k=2)       ros/gisnav/gisnav/core/twist_node.py:95,248
    good = [m for m, n in matches if m.distance < 0.7 * n.distance]          twist_node.py:263-267
    compute_pose(camera_info, mkp_qry, mkp_ref, np.zeros_like(qry))          twist_node.py:289

OpenCV (un-vendored `opencv-python-headless`, unpinned in ros/gisnav/setup.py:116-119; absent here -> PARITY UNPINNED)
`BFMatcher::knnMatchImpl` -> `batchDistance(..., NORM_L2, K=2)`: per query, dist[j] = sqrt(sum_k (q_k - r_k)^2) in
float32 (`batchDistL2_32f` -> `normL2Sqr<float,float>` then `std::sqrt`), and the K best are kept by insertion in
train-index order with strict `<` (modules/core/src/batch_distance.cpp, `BatchDistInvoker`): the result is the first K
of a STABLE ascending sort -- ties go to the lower train index.  cv2.SIFT descriptors are integer-valued 0..255, for
which every partial sum is an exact integer below 2^24, so the float32 accumulation order is immaterial.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np

CONFIDENCE_THRESHOLD = 0.7   # twist_node.py:54
MIN_MATCHES = 30             # twist_node.py:57


def knn_match2(desc_q: np.ndarray, desc_r: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """knnMatch(k=2): (idx [N, min(2, M)] int32, dist [N, min(2, M)] float32), best first."""
    q = np.asarray(desc_q, np.float32)
    r = np.asarray(desc_r, np.float32)
    n, m = len(q), len(r)
    k = min(2, m)
    idx = np.zeros((n, k), np.int32)
    dist = np.zeros((n, k), np.float32)
    for i in range(n):
        diff = r - q[i]                                        # float32
        d2 = np.zeros(m, np.float32)
        for c in range(0, q.shape[1], 4):                      # normL2Sqr's 4-way unrolled float accumulation
            v = diff[:, c:c + 4]
            d2 += (v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1] + v[:, 2] * v[:, 2] + v[:, 3] * v[:, 3]).astype(np.float32)
        d = np.sqrt(d2).astype(np.float32)                     # std::sqrt(float)
        order = np.argsort(d, kind="stable")[:k]               # strict-less insertion in index order == stable sort
        idx[i], dist[i] = order, d[order]
    return idx, dist


def ratio_test(idx: np.ndarray, dist: np.ndarray, ratio: float = CONFIDENCE_THRESHOLD) -> Tuple[np.ndarray, np.ndarray]:
    """`m.distance < ratio * n.distance` with Python-float (f64) arithmetic; returns (pairs [K,2] int64, m.distance [K])."""
    if idx.shape[1] < 2:
        raise ValueError("fewer than two train descriptors: `for m, n in matches` cannot unpack (reference raises)")
    keep = [i for i in range(len(idx)) if float(dist[i, 0]) < ratio * float(dist[i, 1])]
    pairs = np.array([[i, idx[i, 0]] for i in keep], np.int64).reshape(-1, 2)
    return pairs, dist[keep, 0].astype(np.float32)


def twist_pose(k_matrix: np.ndarray, kp_qry: np.ndarray, desc_qry: np.ndarray, kp_ref: np.ndarray, desc_ref: np.ndarray,
               ratio: float = CONFIDENCE_THRESHOLD, min_matches: int = MIN_MATCHES) -> Optional[Tuple[np.ndarray, np.ndarray]]:
    """TwistNode._pose lines 248-289 from the descriptors on: (r, t) or None."""
    from . import pnp_ransac as pr
    if len(kp_qry) < min_matches or len(kp_ref) < 2:
        return None
    idx, dist = knn_match2(desc_qry, desc_ref)
    pairs, _ = ratio_test(idx, dist, ratio)
    if len(pairs) < min_matches:
        return None
    mkp_qry = np.asarray(kp_qry, np.float64)[pairs[:, 0]]
    mkp_ref = np.asarray(kp_ref, np.float64)[pairs[:, 1]]
    return pr.compute_pose(np.asarray(k_matrix, np.float64).reshape(-1), mkp_qry, mkp_ref, None)   # zeros raster -> z = 0
