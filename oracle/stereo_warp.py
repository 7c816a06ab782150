"""ORACLE (test infrastructure only -- imported by tests/ and the measurement tools, never by gisnav_amd/).

CPU restatement of StereoNode's reference-raster preparation:

    orthoimage_arr = cv2.cvtColor(orthoimage_arr, cv2.COLOR_BGR2GRAY)          ros/gisnav/gisnav/core/stereo_node.py:234
    orthoimage_stack = np.dstack((orthoimage_arr, dem_arr))                     stereo_node.py:235
    rotated, M = self._rotate_and_crop_center(orthoimage_stack, map_rotation, crop_shape)   stereo_node.py:246-248, 292-335

OpenCV is un-vendored and absent here (PARITY UNPINNED); this follows OpenCV 4.x imgproc as published:
* `cvtColor(BGR2GRAY)` on u8: (B*1868 + G*9617 + R*4899 + (1 << 13)) >> 14                     (color_rgb.simd.hpp, RGB2Gray<uchar>)
* `getRotationMatrix2D`: alpha = cos(a) * s, beta = sin(a) * s, [[alpha, beta, (1-alpha)cx - beta cy], [-beta, alpha, beta cx + (1-alpha) cy]]
* `warpAffine(INTER_LINEAR, BORDER_CONSTANT 0)`: invert M in f64; X = (cvRound((M1 y + M2) * 1024) + 16 + cvRound(M0 x * 1024)) >> 5
  (same for Y); integer pixel (X >> 5, Y >> 5), 1/32 sub-pixel weights (X & 31, Y & 31) as 15-bit fixed-point bilinear
  taps, result (sum + (1 << 14)) >> 15; taps outside the source read 0                                   (imgwarp.cpp WarpAffineInvoker, remapBilinear)
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np

AB_BITS, INTER_BITS, COEF_BITS = 10, 5, 15


def bgr2gray_u8(bgr: np.ndarray) -> np.ndarray:
    b, g, r = (bgr[..., c].astype(np.int64) for c in range(3))
    return ((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14).astype(np.uint8)


def get_rotation_matrix_2d(center: Tuple[float, float], angle_degrees: float, scale: float) -> np.ndarray:
    a = angle_degrees * (math.pi / 180.0)
    alpha, beta = math.cos(a) * scale, math.sin(a) * scale
    cx, cy = float(np.float32(center[0])), float(np.float32(center[1]))           # Point2f
    return np.array([[alpha, beta, (1 - alpha) * cx - beta * cy], [-beta, alpha, beta * cx + (1 - alpha) * cy]], np.float64)


def invert_affine(m: np.ndarray) -> np.ndarray:
    """The in-place inversion at the top of cv::warpAffine (f64, this exact operation order)."""
    M = [float(v) for v in np.asarray(m, np.float64).reshape(6)]
    D = M[0] * M[4] - M[1] * M[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[4] * D, M[0] * D
    M[0] = A11; M[1] *= -D; M[3] *= -D; M[4] = A22
    b1 = -M[0] * M[2] - M[1] * M[5]
    b2 = -M[3] * M[2] - M[4] * M[5]
    M[2], M[5] = b1, b2
    return np.array(M, np.float64).reshape(2, 3)


def bilinear_tab() -> np.ndarray:
    """BilinearTab_i[32*32][4] of initInterTab2D(INTER_LINEAR, fixpt): exact for every entry except (0, 0), whose
    32768 saturates to 32767 and is compensated on tap 3 (no effect on any u8 result; kept for fidelity)."""
    fy, fx = np.meshgrid(np.arange(32), np.arange(32), indexing="ij")
    tab = np.stack([(32 - fx) * (32 - fy) * 32, fx * (32 - fy) * 32, (32 - fx) * fy * 32, fx * fy * 32], -1).reshape(1024, 4).astype(np.int64)
    tab[0] = (32767, 0, 0, 1)
    return tab


def warp_affine_u8(src: np.ndarray, m: np.ndarray, dsize: Tuple[int, int], rows: slice = slice(None), cols: slice = slice(None)) -> np.ndarray:
    """cv2.warpAffine(src, m, dsize) for u8 images with 1..4 channels (optionally only a sub-rectangle of the output)."""
    src = np.asarray(src, np.uint8)
    if src.ndim == 2:
        return warp_affine_u8(src[..., None], m, dsize, rows, cols)[..., 0]
    h, w = src.shape[:2]
    dw, dh = dsize
    M = invert_affine(m).reshape(6)
    xs = np.arange(dw, dtype=np.float64)[cols]
    ys = np.arange(dh, dtype=np.float64)[rows]
    adelta = np.rint(M[0] * xs * (1 << AB_BITS)).astype(np.int64)
    bdelta = np.rint(M[3] * xs * (1 << AB_BITS)).astype(np.int64)
    rd = (1 << AB_BITS) // 32 // 2
    X0 = np.rint((M[1] * ys + M[2]) * (1 << AB_BITS)).astype(np.int64) + rd
    Y0 = np.rint((M[4] * ys + M[5]) * (1 << AB_BITS)).astype(np.int64) + rd
    X = (X0[:, None] + adelta[None, :]) >> (AB_BITS - INTER_BITS)
    Y = (Y0[:, None] + bdelta[None, :]) >> (AB_BITS - INTER_BITS)
    sx = np.clip(X >> INTER_BITS, -32768, 32767)                                   # saturate_cast<short>
    sy = np.clip(Y >> INTER_BITS, -32768, 32767)
    wt = bilinear_tab()[(Y & 31) * 32 + (X & 31)]                                  # [H, W, 4]
    acc = np.zeros(X.shape + (src.shape[2],), np.int64)
    for k, (oy, ox) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
        yy, xx = sy + oy, sx + ox
        inside = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
        tap = src[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)].astype(np.int64) * inside[..., None]
        acc += tap * wt[..., k, None]
    return np.clip((acc + (1 << (COEF_BITS - 1))) >> COEF_BITS, 0, 255).astype(np.uint8)


def rotate_and_crop_center(image: np.ndarray, angle_degrees: float, shape: Tuple[int, int]) -> Tuple[np.ndarray, np.ndarray]:
    """StereoNode._rotate_and_crop_center (stereo_node.py:292-335): (cropped rotated image, matrix back to the original frame)."""
    h, w = image.shape[:2]
    center = (w // 2, h // 2)
    rotation_matrix = get_rotation_matrix_2d(center, angle_degrees, 1.0)
    dx = center[0] - shape[1] // 2
    dy = center[1] - shape[0] // 2
    cropped = warp_affine_u8(image, rotation_matrix, (w, h), slice(dy, dy + shape[0]), slice(dx, dx + shape[1]))
    extended = np.vstack([rotation_matrix, [0, 0, 1]])
    inverse = np.linalg.inv(extended)
    T = np.array([[1, 0, dx], [0, 1, dy], [0, 0, 1]])
    return cropped, inverse @ T


def stereo_reference(bgr: np.ndarray, dem: np.ndarray, angle_degrees: float, crop_shape: Tuple[int, int]):
    """stereo_node.py:229-262: gray + DEM stack -> rotate and crop -> (reference u8, dem u8, inverse matrix)."""
    stack = np.dstack((bgr2gray_u8(bgr), dem))
    out, minv = rotate_and_crop_center(stack, angle_degrees, crop_shape)
    return out[:, :, 0], out[:, :, 1], minv
