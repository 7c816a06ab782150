"""Host-side mirror of TwistNode's visual-odometry matcher.

`ros/gisnav/gisnav/core/twist_node.py:95,227-289`:

    self._bf = cv2.BFMatcher(crossCheck=False)
    matches = self._bf.knnMatch(desc_qry, desc_ref, k=2)
    good = [m for m, n in matches if m.distance < self.CONFIDENCE_THRESHOLD * n.distance]
    pose = compute_pose(camera_info, mkp_qry, mkp_ref, np.zeros_like(qry))

Marshalling only: the arithmetic runs in libgisnav_amd.so (`gn_vo_match`, `gn_vo_estimate`); there is no CPU path.
"""
from __future__ import annotations

from typing import List, NamedTuple, Optional, Tuple

import numpy as np
import torch

from . import _lib
from .engine import PoseEngine

CONFIDENCE_THRESHOLD = 0.7   # twist_node.py:54
MIN_MATCHES = 30             # twist_node.py:57


class DMatch(NamedTuple):
    """The fields of cv2.DMatch the reference reads."""
    queryIdx: int
    trainIdx: int
    distance: float


class BFMatcher:
    """`cv2.BFMatcher(normType=NORM_L2, crossCheck=False)` with the one method TwistNode calls."""

    def __init__(self, engine: Optional[PoseEngine] = None, device: int = 0, max_kpts: int = 2048):
        self._eng = engine if engine is not None else PoseEngine(device, max_batch=1, max_kpts=max_kpts, precision="f32")

    @property
    def engine(self) -> PoseEngine:
        return self._eng

    def _run(self, desc_qry: np.ndarray, desc_ref: np.ndarray, ratio: float):
        dev = self._eng.device
        dq = self._eng.to_device("vo_desc_q", np.asarray(desc_qry)[None])     # (pinned staging, gisnav_amd/upload.py)
        dr = self._eng.to_device("vo_desc_r", np.asarray(desc_ref)[None])
        nq = torch.full((1,), dq.shape[1], dtype=torch.int32, device=dev)
        nr = torch.full((1,), dr.shape[1], dtype=torch.int32, device=dev)
        return self._eng.vo_match(dq, nq, dr, nr, ratio, want_knn=True)

    def knnMatch(self, desc_qry: np.ndarray, desc_ref: np.ndarray, k: int = 2) -> List[Tuple[DMatch, ...]]:
        if k != 2:
            raise ValueError("only k=2 is implemented (the reference's call)")
        n_q, n_r = len(desc_qry), len(desc_ref)
        if n_q == 0:
            return []
        _, _, _, nn_idx, nn_dist = self._run(desc_qry, desc_ref, CONFIDENCE_THRESHOLD)
        ii, dd = self._eng.to_host(nn_idx[0, :n_q], nn_dist[0, :n_q])
        il, dl = ii.tolist(), dd.tolist()            # (one pass to Python numbers instead of 4 n numpy scalar reads)
        if n_r >= 2:
            return [(DMatch(q, i[0], d[0]), DMatch(q, i[1], d[1])) for q, (i, d) in enumerate(zip(il, dl))]
        return [(DMatch(q, i[0], d[0]),) for q, (i, d) in enumerate(zip(il, dl))]

    def ratio_matches(self, desc_qry: np.ndarray, desc_ref: np.ndarray, ratio: float = CONFIDENCE_THRESHOLD) -> List[DMatch]:
        """knnMatch(k=2) + `m.distance < ratio * n.distance` in one device pass (twist_node.py:263-267)."""
        idx, dist, n_good = self._run(desc_qry, desc_ref, ratio)[:3]
        k = int(self._eng.to_host(n_good)[0][0])
        ii, dd = self._eng.to_host(idx[0, :k], dist[0, :k])
        return [DMatch(i[0], i[1], d) for i, d in zip(ii.tolist(), dd.tolist())]


def twist_pose(engine: PoseEngine, k_matrix: np.ndarray, kp_qry: np.ndarray, desc_qry: np.ndarray,
               kp_ref: np.ndarray, desc_ref: np.ndarray, ratio: float = CONFIDENCE_THRESHOLD,
               min_matches: int = MIN_MATCHES) -> Optional[Tuple[np.ndarray, np.ndarray]]:
    """`TwistNode._pose` from the descriptors on (twist_node.py:248-289): (r, t) or None.  kp: [N,2] pixel coordinates."""
    n_q, n_r = len(kp_qry), len(kp_ref)
    if n_q < min_matches or n_r < 2:      # `len(matches) < MIN_MATCHES` -> None (twist_node.py:256)
        return None
    dev = engine.device

    def pack(kp, desc, side):
        k4 = np.zeros((1, len(kp), 4), np.float32)
        k4[0, :, :2] = kp
        return (engine.to_device("vo_desc_" + side, np.asarray(desc)[None]), engine.to_device("vo_kpt_" + side, k4),
                torch.full((1,), len(kp), dtype=torch.int32, device=dev))

    dq, kq, nq = pack(kp_qry, desc_qry, "q")
    dr, kr, nr = pack(kp_ref, desc_ref, "r")
    out = engine.vo_estimate(dict(desc_q=dq, kpt_q=kq, n_q=nq, desc_r=dr, kpt_r=kr, n_r=nr, kpt_format=_lib.GN_KPT_XYSA),
                             k_matrix, ratio, min_matches)
    ok, R, t = engine.to_host(out["ok"], out["R"], out["t"])
    if not bool(ok[0]):
        return None
    return R[0], t[0]
