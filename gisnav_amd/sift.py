"""Host-side mirror of `cv2.SIFT_create()` as GISNav uses it: `detectAndCompute(image, None)` for
the reference tile (pose_node.py:122,230-232) and for every camera frame (twist_node.py:93,227-245).  Marshalling only:
scale space, keypoints and descriptors are produced by libgisnav_amd.so (`gn_sift_detect_and_compute`, `gn_sift_detect_and_compute_batch`)."""
from __future__ import annotations

import ctypes as C
from typing import List, NamedTuple, Optional, Tuple

import numpy as np
import torch

from . import _lib
from .engine import PoseEngine, _ptr


class KeyPoint(NamedTuple):
    """The cv2.KeyPoint fields GISNav reads (pt, size, angle) plus response / octave."""
    pt: Tuple[float, float]
    size: float
    angle: float
    response: float
    octave: int


class SIFT:
    """`cv2.SIFT_create()` stand-in.  `detectAndCompute(image, None)` returns (list of KeyPoint, descriptors [N,128] float32)
    like cv2; `detect_and_compute_device` keeps everything in HBM in the layout `gn_match` / `gn_estimate` consume."""

    def __init__(self, engine: Optional[PoseEngine] = None, device: int = 0, max_keypoints: int = 8192):
        self._eng = engine if engine is not None else PoseEngine(device, max_batch=1, max_kpts=128, precision="f32")
        self._max = int(max_keypoints)

    def detect_and_compute_device(self, image):
        """image: (H, W) uint8 numpy array or device tensor.  Returns (kpt_xysa [N,4] f32, response [N], octave [N] i32,
        desc [N,128] f32) as device tensors, N keypoints in OpenCV's order."""
        eng = self._eng
        t = eng.to_device("sift_image", image, torch.uint8)       # (pinned staging: pageable uploads stall on this platform, gisnav_amd/upload.py)
        assert t.dtype == torch.uint8 and t.dim() == 2, "expected a single-channel uint8 image"
        H, W = int(t.shape[0]), int(t.shape[1])
        kpt = torch.empty((self._max, 4), dtype=torch.float32, device=eng.device)
        resp = torch.empty((self._max,), dtype=torch.float32, device=eng.device)
        octv = torch.empty((self._max,), dtype=torch.int32, device=eng.device)
        desc = torch.empty((self._max, 128), dtype=torch.float32, device=eng.device)
        n = C.c_int32(0)
        rc = eng.lib.gn_sift_detect_and_compute(eng.ctx, _ptr(t.contiguous()), H, W, self._max, _ptr(kpt), _ptr(resp), _ptr(octv), _ptr(desc),
                                                C.byref(n), eng._stream())
        _lib.check(eng.ctx, rc, "gn_sift_detect_and_compute")
        k = int(n.value)
        return kpt[:k], resp[:k], octv[:k], desc[:k]

    def detect_and_compute_batch_device(self, images):
        """images: (B, H, W) uint8 numpy array or device tensor, all of one size.  One pass over the whole batch
        (`gn_sift_detect_and_compute_batch`).  Returns (kpt_xysa [B,max,4], response [B,max], octave [B,max], desc [B,max,128],
        n [B] int32 on the host): image b's keypoints are rows [0, n[b]) -- the padded layout `PoseEngine.estimate` takes."""
        eng = self._eng
        t = eng.to_device("sift_images", images, torch.uint8)
        assert t.dtype == torch.uint8 and t.dim() == 3, "expected a (B, H, W) uint8 stack"
        B, H, W = (int(v) for v in t.shape)
        kpt = torch.empty((B, self._max, 4), dtype=torch.float32, device=eng.device)
        resp = torch.empty((B, self._max), dtype=torch.float32, device=eng.device)
        octv = torch.empty((B, self._max), dtype=torch.int32, device=eng.device)
        desc = torch.empty((B, self._max, 128), dtype=torch.float32, device=eng.device)
        n = (C.c_int32 * B)()
        rc = eng.lib.gn_sift_detect_and_compute_batch(eng.ctx, _ptr(t.contiguous()), B, H, W, self._max, _ptr(kpt), _ptr(resp), _ptr(octv), _ptr(desc),
                                                      C.cast(n, C.POINTER(C.c_int32)), eng._stream())
        _lib.check(eng.ctx, rc, "gn_sift_detect_and_compute_batch")
        return kpt, resp, octv, desc, np.frombuffer(n, dtype=np.int32).copy()

    def last_totals(self, B: int = 1) -> np.ndarray:
        """Distinct keypoints per image found by the last call BEFORE the max_keypoints cap (the cap keeps the strongest by
        response, like cv2's nfeatures); a caller that wants them all re-creates SIFT with a larger max_keypoints."""
        n = (C.c_int32 * B)()
        _lib.check(self._eng.ctx, self._eng.lib.gn_sift_last_totals(self._eng.ctx, B, C.cast(n, C.POINTER(C.c_int32))), "gn_sift_last_totals")
        return np.frombuffer(n, dtype=np.int32).copy()

    def detectAndCompute(self, image, mask=None):
        if mask is not None:
            raise ValueError("masks are not supported (GISNav passes None)")
        kpt, resp, octv, desc = self.detect_and_compute_device(image)
        k, r, o, d = self._eng.to_host(kpt, resp, octv, desc)
        # (.tolist() converts every field to a Python number in one pass: three times faster than per-element float() on a few hundred keypoints)
        kps: List[KeyPoint] = [KeyPoint((x, y), sz, an, rs, oc) for (x, y, sz, an), rs, oc in zip(k.tolist(), r.tolist(), o.tolist())]
        return kps, d

    def as_extractor(self):
        """The `extractor(ref_u8) -> (kp, desc, size, angle)` callable `gisnav_amd.pose_node.PoseNode` takes."""
        def extractor(ref_u8):
            kpt, _, _, desc = self.detect_and_compute_device(ref_u8)
            total = int(self.last_totals(1)[0])
            if total > self._max:                      # cv2.SIFT_create() is unbounded (pose_node.py:122): enlarge the buffers, extract again
                self._max = ((total + 1023) // 1024) * 1024
                kpt, _, _, desc = self.detect_and_compute_device(ref_u8)
            k, d = self._eng.to_host(kpt, desc)
            return k[:, :2], d, k[:, 2], k[:, 3]
        return extractor
