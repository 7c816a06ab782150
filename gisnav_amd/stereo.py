"""Host-side mirror of StereoNode's reference-raster preparation.

`ros/gisnav/gisnav/core/stereo_node.py:229-262, 292-335`: BGR orthoimage -> gray, stacked with the DEM raster, rotated
about the centre (`cv2.getRotationMatrix2D` + `cv2.warpAffine`) and centre-cropped to the camera resolution.
Marshalling only: the pixels are produced by libgisnav_amd.so (`gn_rotate_crop_center`, `gn_stereo_reference`).
"""
from __future__ import annotations

import ctypes as C
from typing import Tuple

import numpy as np
import torch

from . import _lib
from .engine import PoseEngine, _ptr


def rotate_and_crop_center(engine: PoseEngine, image, angle_degrees: float, shape: Tuple[int, int]):
    """`StereoNode._rotate_and_crop_center(image, angle_degrees, shape)` for an (H, W, 2) u8 stack (numpy array or device
    tensor).  Returns (cropped (h, w, 2) u8 device tensor, 3x3 f64 matrix back to the original frame)."""
    t = engine.to_device("stereo_image", image, torch.uint8)      # (pinned staging, gisnav_amd/upload.py)
    assert t.dtype == torch.uint8 and t.dim() == 3 and t.shape[2] == 2, "expected an (H, W, 2) uint8 stack"
    H, W = int(t.shape[0]), int(t.shape[1])
    out = torch.empty((shape[0], shape[1], 2), dtype=torch.uint8, device=engine.device)
    back = np.zeros(9, np.float64)
    rc = engine.lib.gn_rotate_crop_center(engine.ctx, _ptr(t.contiguous()), H, W, float(angle_degrees), int(shape[0]), int(shape[1]),
                                          _ptr(out), back.ctypes.data_as(_lib.c_f64p), engine._stream())
    _lib.check(engine.ctx, rc, "gn_rotate_crop_center")
    return out, back.reshape(3, 3)


def stereo_reference(engine: PoseEngine, orthoimage_bgr, dem, map_rotation: float, crop_shape: Tuple[int, int]):
    """stereo_node.py:229-262 in one device pass: (reference (h, w) u8, dem (h, w) u8, 3x3 f64 matrix)."""
    dev = engine.device
    b = engine.to_device("stereo_bgr", orthoimage_bgr, torch.uint8)
    d = engine.to_device("stereo_dem", dem, torch.uint8)
    assert b.dtype == torch.uint8 and b.dim() == 3 and b.shape[2] == 3 and d.shape == b.shape[:2]
    H, W = int(b.shape[0]), int(b.shape[1])
    ref = torch.empty(crop_shape, dtype=torch.uint8, device=dev)
    dm = torch.empty(crop_shape, dtype=torch.uint8, device=dev)
    back = np.zeros(9, np.float64)
    rc = engine.lib.gn_stereo_reference(engine.ctx, _ptr(b.contiguous()), _ptr(d.contiguous()), H, W, float(map_rotation),
                                        int(crop_shape[0]), int(crop_shape[1]), _ptr(ref), _ptr(dm),
                                        back.ctypes.data_as(_lib.c_f64p), engine._stream())
    _lib.check(engine.ctx, rc, "gn_stereo_reference")
    return ref, dm, back.reshape(3, 3)
