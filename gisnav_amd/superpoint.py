"""Host-side wrapper of the SuperPoint extractor in libgisnav_amd.so (`gn_sp_load_tensor`, `gn_sp_detect_and_describe`) -- the
feature extractor of BASELINE.json configs[4] (SuperPoint + LightGlue), which the reference tree does not contain (its extractor
is cv2.SIFT, pose_node.py:122).  Marshalling only: every convolution, the score map, NMS, top-k and the descriptor sampling run in
hand-written gfx950 kernels (csrc/gn_superpoint.hip)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from .engine import PoseEngine, _ptr


class SuperPoint:
    """`detect_and_describe_device(images)` -> keypoints in the GN_KPT_XYSA layout + 256-d descriptors, ready for a
    `PoseEngine(feature="superpoint")`; weights under transformers' SuperPointForKeypointDetection key names."""

    ARITHMETIC = {"exact_f32": 0, "split_fp16": 1, "fp16": 2}

    def __init__(self, engine: Optional[PoseEngine] = None, device: int = 0, max_keypoints: int = 1024, state_dict: Optional[Dict] = None,
                 arithmetic: Optional[str] = None):
        """arithmetic: None keeps the context's default (split_fp16 in f16x2 contexts, exact_f32 elsewhere); "fp16" = one fp16 product per
        block, the 16-bit-operand arithmetic BASELINE.json configs[4] names (not f32-accurate, see include/gisnav_amd.h)."""
        self._eng = engine if engine is not None else PoseEngine(device, max_batch=1, max_kpts=128, precision="f32", feature="superpoint")
        self._max = int(max_keypoints)
        if arithmetic is not None:
            self._eng.sp_set_arithmetic(self.ARITHMETIC[arithmetic])
        if state_dict is not None:
            self.load_state_dict(state_dict)

    def load_state_dict(self, sd) -> None:
        self._eng.sp_load_state_dict(sd)     # lives in the context (gn_resize keeps it)

    def detect_and_describe_device(self, images):
        """images: (B, H, W) uint8 (scaled by 1/255 like the published pre-processing) or float32 in [0, 1], numpy or device tensor;
        H and W multiples of 8.  Returns (kpt_xysa [B,max,4], score [B,max], desc [B,max,256] device tensors, n [B] int32 host)."""
        eng = self._eng
        t = images if isinstance(images, torch.Tensor) else eng.to_device("sp_images", images, torch.float32 if np.asarray(images).dtype != np.uint8 else torch.uint8)
        if t.dtype == torch.uint8:
            t = t.to(torch.float32) * (1.0 / 255.0)
        t = t.to(device=eng.device, dtype=torch.float32).contiguous()
        assert t.dim() == 3, "expected a (B, H, W) stack"
        B, H, W = (int(v) for v in t.shape)
        kpt = torch.zeros((B, self._max, 4), dtype=torch.float32, device=eng.device)
        score = torch.zeros((B, self._max), dtype=torch.float32, device=eng.device)
        desc = torch.zeros((B, self._max, 256), dtype=torch.float32, device=eng.device)
        n = (C.c_int32 * B)()
        rc = eng.lib.gn_sp_detect_and_describe(eng.ctx, _ptr(t), B, H, W, self._max, _ptr(kpt), _ptr(score), _ptr(desc),
                                               C.cast(n, C.POINTER(C.c_int32)), eng._stream())
        _lib.check(eng.ctx, rc, "gn_sp_detect_and_describe")
        return kpt, score, desc, np.frombuffer(n, dtype=np.int32).copy()
