"""ORACLE (test infrastructure only -- imported by tests/ and measurement tools, never by gisnav_amd/).

CPU restatement of `cv2.SIFT_create().detectAndCompute(image, None)` as GISNav calls it for the reference tile
(ros/gisnav/gisnav/core/pose_node.py:122,230-232) and for every camera frame (core/twist_node.py:93,227-245) --
SURVEY.md §8(f) row 1.  OpenCV is un-vendored and absent here: this follows OpenCV 4.x `features2d/src/sift.dispatch.cpp`
and `sift.simd.hpp` as published (defaults nfeatures=0, nOctaveLayers=3, contrastThreshold=0.04, edgeThreshold=10,
sigma=1.6, float descriptors) -- PARITY UNPINNED against a real cv2.

All image arithmetic is float32 with the operation order written here (numpy never fuses a multiply-add); the HIP
kernels are compiled with -ffp-contract=off and follow the same order, so the two agree bit for bit.  Where OpenCV
calls vectorised math (`hal::exp32f`, `hal::fastAtan2`, `cosf`, `powf`) the restatement pins ONE definition used on both
sides: `exp32` below (float32 range reduction + degree-6 polynomial), the per-keypoint cos / sin / pow in float64
rounded to float32, `fastAtan2` as OpenCV's degree-7 polynomial in float32.
"""
from __future__ import annotations

import math
from typing import List, Tuple

import numpy as np

F = np.float32
SIFT_IMG_BORDER = 5
SIFT_MAX_INTERP_STEPS = 5
SIFT_ORI_HIST_BINS = 36
SIFT_ORI_SIG_FCTR = F(1.5)
SIFT_ORI_RADIUS = F(4.5)            # 3 * SIFT_ORI_SIG_FCTR
SIFT_ORI_PEAK_RATIO = F(0.8)
SIFT_DESCR_WIDTH = 4
SIFT_DESCR_HIST_BINS = 8
SIFT_DESCR_SCL_FCTR = F(3.0)
SIFT_DESCR_MAG_THR = F(0.2)
SIFT_INT_DESCR_FCTR = F(512.0)
SIFT_INIT_SIGMA = 0.5
FLT_EPSILON = F(1.1920929e-07)


# ------------------------------------------------------------------------------------------------ scale space
def gaussian_kernel(sigma: float) -> np.ndarray:
    """cv::getGaussianKernel(ksize, sigma, CV_32F), ksize = cvRound(sigma * 8 + 1) | 1 (float images)."""
    n = int(np.rint(sigma * 8 + 1)) | 1
    scale2x = -0.5 / (sigma * sigma)
    k = [math.exp(scale2x * (i - (n - 1) * 0.5) * (i - (n - 1) * 0.5)) for i in range(n)]      # libm exp, as the C++ side
    total = 0.0
    for v in k:
        total += v
    inv = 1.0 / total
    return np.array([v * inv for v in k], np.float64).astype(F)


def _reflect101(i: np.ndarray, n: int) -> np.ndarray:
    """cv::borderInterpolate(p, len, BORDER_REFLECT_101), repeated until inside (kernels wider than a tiny octave)."""
    if n == 1:
        return np.zeros_like(i)
    i = i.copy()
    while True:
        bad = (i < 0) | (i >= n)
        if not bad.any():
            return i
        i = np.where(i < 0, -i, i)
        i = np.where(i >= n, 2 * n - 2 - i, i)


def gaussian_blur(img: np.ndarray, sigma: float) -> np.ndarray:
    """cv::GaussianBlur(img, Size(), sigma, sigma) on float32, BORDER_REFLECT_101: row pass (taps left to right), then
    the symmetric column pass (centre tap, then pairs)."""
    k = gaussian_kernel(sigma)
    n, r = len(k), len(k) // 2
    h, w = img.shape
    cols = _reflect101(np.arange(-r, w + r), w)
    p = img[:, cols]
    row = k[0] * p[:, 0:w]
    for t in range(1, n):
        row = row + k[t] * p[:, t:t + w]
    rows = _reflect101(np.arange(-r, h + r), h)
    q = row[rows, :]
    out = k[r] * q[r:r + h]
    for t in range(1, r + 1):
        out = out + k[r + t] * (q[r + t:r + t + h] + q[r - t:r - t + h])
    return out.astype(F)


def resize_linear_2x(img: np.ndarray) -> np.ndarray:
    """cv::resize(img, Size(2w, 2h), INTER_LINEAR) on float32."""
    h, w = img.shape

    def taps(n_dst, n_src):
        f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * 0.5 - 0.5).astype(F)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(F)).astype(F)
        lo = s < 0
        f[lo] = 0; s[lo] = 0
        hi = s + 1 >= n_src
        f[hi] = 0; s[hi] = n_src - 1
        return s, np.minimum(s + 1, n_src - 1), (F(1) - f).astype(F), f

    sx, sx1, a0, a1 = taps(2 * w, w)
    sy, sy1, b0, b1 = taps(2 * h, h)
    hor = img[:, sx] * a0[None, :] + img[:, sx1] * a1[None, :]
    return (hor[sy] * b0[:, None] + hor[sy1] * b1[:, None]).astype(F)


def build_pyramids(gray_u8: np.ndarray, n_layers: int = 3, sigma: float = 1.6):
    """createInitialImage (doubled) + buildGaussianPyramid + buildDoGPyramid.  Returns (gauss[o][0..n+2], dog[o][0..n+1])."""
    base = resize_linear_2x(gray_u8.astype(F))
    # createInitialImage takes `float sigma` and evaluates this line in float: sqrtf(max(sigma * sigma - SIFT_INIT_SIGMA^2 * 4, 0.01f))
    sf, s0 = F(sigma), F(SIFT_INIT_SIGMA)
    sig_diff = float(np.sqrt(np.maximum(F(sf * sf) - F(F(s0 * s0) * F(4)), F(0.01))).astype(F))
    base = gaussian_blur(base, sig_diff)
    n_octaves = int(np.rint(math.log(min(base.shape)) / math.log(2.0) - 2)) + 1
    sig = [sigma]
    k = 2.0 ** (1.0 / n_layers)
    for i in range(1, n_layers + 3):
        sp = (k ** (i - 1)) * sigma
        st = sp * k
        sig.append(math.sqrt(st * st - sp * sp))
    gauss, dog = [], []
    for o in range(n_octaves):
        layers = []
        for i in range(n_layers + 3):
            if o == 0 and i == 0:
                layers.append(base)
            elif i == 0:
                src = gauss[o - 1][n_layers]
                layers.append(np.ascontiguousarray(src[0:2 * (src.shape[0] // 2):2, 0:2 * (src.shape[1] // 2):2]))     # INTER_NEAREST, half size
            else:
                layers.append(gaussian_blur(layers[i - 1], sig[i]))
        gauss.append(layers)
        dog.append([(layers[i + 1] - layers[i]).astype(F) for i in range(n_layers + 2)])
    return gauss, dog


# ------------------------------------------------------------------------------------------------ restated math helpers
def fast_atan2_deg(y: np.ndarray, x: np.ndarray) -> np.ndarray:
    """cv::fastAtan2 (degrees, 0..360), OpenCV's polynomial in float32."""
    p1, p3, p5, p7 = F(0.9997878412794807 * 57.29577951308232), F(-0.3258083974640975 * 57.29577951308232), \
        F(0.1555786518463281 * 57.29577951308232), F(-0.04432655554792128 * 57.29577951308232)
    y = np.asarray(y, F); x = np.asarray(x, F)
    ax, ay = np.abs(x), np.abs(y)
    eps = F(2.220446049250313e-16)
    swap = ax < ay
    num = np.where(swap, ax, ay); den = np.where(swap, ay, ax)
    c = (num / (den + eps)).astype(F)
    c2 = (c * c).astype(F)
    a = ((((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c).astype(F)
    a = np.where(swap, F(90.0) - a, a).astype(F)
    a = np.where(x < 0, F(180.0) - a, a).astype(F)
    a = np.where(y < 0, F(360.0) - a, a).astype(F)
    return a


def exp32(x: np.ndarray) -> np.ndarray:
    """The ONE float32 exponential both sides use for the Gaussian sample weights (OpenCV's hal::exp32f is likewise an
    ~1e-7-accurate approximation, not libm): 2^n * P(f), n = rint(x log2 e), f = x log2 e - n, P = degree-6 Taylor of 2^f,
    every operation a separate float32 multiply or add."""
    x = np.asarray(x, F)
    t = (x * F(1.4426950408889634)).astype(F)
    n = np.rint(t).astype(F)
    f = (t - n).astype(F)
    p = F(0.00015403530393381608)
    for c in (F(0.0013333558146428443), F(0.009618129107628477), F(0.05550410866482158), F(0.2402265069591007), F(0.6931471805599453), F(1.0)):
        p = ((p * f).astype(F) + c).astype(F)
    return np.ldexp(p, n.astype(np.int32)).astype(F)


def cv_round(v) -> np.ndarray:
    return np.rint(v).astype(np.int64)


def _lu_solve3(H: np.ndarray, b: np.ndarray):
    """Matx33f::solve(b, DECOMP_LU): Gaussian elimination with partial pivoting in float32; None when singular."""
    A = H.astype(F).copy(); x = b.astype(F).copy()
    for i in range(3):
        k = i + int(np.argmax(np.abs(A[i:, i])))
        if abs(A[k, i]) < FLT_EPSILON:
            return None
        if k != i:
            A[[i, k]] = A[[k, i]]; x[[i, k]] = x[[k, i]]
        d = F(-1.0) / A[i, i]
        for j in range(i + 1, 3):
            alpha = F(A[j, i] * d)
            for c in range(i + 1, 3):
                A[j, c] = F(A[j, c] + alpha * A[i, c])
            x[j] = F(x[j] + alpha * x[i])
    for i in (2, 1, 0):
        s = x[i]
        for c in range(i + 1, 3):
            s = F(s - A[i, c] * x[c])
        x[i] = F(s / A[i, i])
    return x


# ------------------------------------------------------------------------------------------------ detection
def _adjust_local_extrema(dogs, octv, layer, r, c, n_layers, contrast_thr, edge_thr, sigma):
    img_scale = F(1.0 / 255.0)
    deriv_scale, second_scale, cross_scale = F(img_scale * F(0.5)), img_scale, F(img_scale * F(0.25))
    xi = xr = xc = F(0)
    i = 0
    while i < SIFT_MAX_INTERP_STEPS:
        img, prv, nxt = dogs[layer], dogs[layer - 1], dogs[layer + 1]
        dD = np.array([(img[r, c + 1] - img[r, c - 1]) * deriv_scale, (img[r + 1, c] - img[r - 1, c]) * deriv_scale,
                       (nxt[r, c] - prv[r, c]) * deriv_scale], F)
        v2 = F(img[r, c] * F(2))
        dxx = F((img[r, c + 1] + img[r, c - 1] - v2) * second_scale)
        dyy = F((img[r + 1, c] + img[r - 1, c] - v2) * second_scale)
        dss = F((nxt[r, c] + prv[r, c] - v2) * second_scale)
        dxy = F((img[r + 1, c + 1] - img[r + 1, c - 1] - img[r - 1, c + 1] + img[r - 1, c - 1]) * cross_scale)
        dxs = F((nxt[r, c + 1] - nxt[r, c - 1] - prv[r, c + 1] + prv[r, c - 1]) * cross_scale)
        dys = F((nxt[r + 1, c] - nxt[r - 1, c] - prv[r + 1, c] + prv[r - 1, c]) * cross_scale)
        X = _lu_solve3(np.array([[dxx, dxy, dxs], [dxy, dyy, dys], [dxs, dys, dss]], F), dD)
        if X is None:
            X = np.zeros(3, F)
        xi, xr, xc = F(-X[2]), F(-X[1]), F(-X[0])
        if abs(xi) < 0.5 and abs(xr) < 0.5 and abs(xc) < 0.5:
            break
        if abs(xi) > 715827882.0 or abs(xr) > 715827882.0 or abs(xc) > 715827882.0:      # INT_MAX / 3
            return None
        c += int(np.rint(xc)); r += int(np.rint(xr)); layer += int(np.rint(xi))
        rows, cols = dogs[0].shape
        if layer < 1 or layer > n_layers or c < SIFT_IMG_BORDER or c >= cols - SIFT_IMG_BORDER or r < SIFT_IMG_BORDER or r >= rows - SIFT_IMG_BORDER:
            return None
        i += 1
    if i >= SIFT_MAX_INTERP_STEPS:
        return None
    img, prv, nxt = dogs[layer], dogs[layer - 1], dogs[layer + 1]
    dD = np.array([(img[r, c + 1] - img[r, c - 1]) * deriv_scale, (img[r + 1, c] - img[r - 1, c]) * deriv_scale,
                   (nxt[r, c] - prv[r, c]) * deriv_scale], F)
    t = F(F(F(dD[0] * xc) + F(dD[1] * xr)) + F(dD[2] * xi))
    contr = F(F(img[r, c] * img_scale) + F(t * F(0.5)))
    if F(abs(contr) * F(n_layers)) < F(contrast_thr):
        return None
    v2 = F(img[r, c] * F(2))
    dxx = F((img[r, c + 1] + img[r, c - 1] - v2) * second_scale)
    dyy = F((img[r + 1, c] + img[r - 1, c] - v2) * second_scale)
    dxy = F((img[r + 1, c + 1] - img[r + 1, c - 1] - img[r - 1, c + 1] + img[r - 1, c - 1]) * cross_scale)
    tr = F(dxx + dyy)
    det = F(F(dxx * dyy) - F(dxy * dxy))
    if det <= 0 or F(F(tr * tr) * F(edge_thr)) >= F(F(F(edge_thr + 1) * F(edge_thr + 1)) * det):
        return None
    scale = F(1 << octv)
    x = F(F(F(c) + xc) * scale); y = F(F(F(r) + xr) * scale)
    octave = octv + (layer << 8) + (int(np.rint(F(F(xi + F(0.5)) * F(255)))) << 16)
    size = F(F(F(sigma) * F(math.pow(2.0, float(F(F(layer) + xi) / F(n_layers))))) * scale) * F(2)
    return dict(x=x, y=y, octave=octave, size=F(size), response=F(abs(contr)), r=r, c=c, layer=layer)


def _orientation_hist(img, px, py, radius, sigma_w):
    n = SIFT_ORI_HIST_BINS
    expf_scale = F(F(-1.0) / F(F(2.0) * F(sigma_w * sigma_w)))
    rows, cols = img.shape
    ii, jj = np.meshgrid(np.arange(-radius, radius + 1), np.arange(-radius, radius + 1), indexing="ij")
    yy, xx = py + ii, px + jj
    ok = (yy > 0) & (yy < rows - 1) & (xx > 0) & (xx < cols - 1)
    yy, xx, ii, jj = yy[ok], xx[ok], ii[ok], jj[ok]                          # raster order of the double loop
    dx = (img[yy, xx + 1] - img[yy, xx - 1]).astype(F)
    dy = (img[yy - 1, xx] - img[yy + 1, xx]).astype(F)
    W = exp32(((ii * ii + jj * jj).astype(F) * expf_scale).astype(F))
    ori = fast_atan2_deg(dy, dx)
    mag = np.sqrt((dx * dx + dy * dy).astype(F)).astype(F)
    b = cv_round(F(n / 360.0) * ori)
    b = np.where(b >= n, b - n, b); b = np.where(b < 0, b + n, b)
    temph = np.zeros(n, F)
    np.add.at(temph, b, (W * mag).astype(F))                                 # in-order float32 accumulation
    ext = np.concatenate([temph[-2:], temph, temph[:2]])
    hist = ((ext[0:n] + ext[4:n + 4]) * F(1.0 / 16.0) + (ext[1:n + 1] + ext[3:n + 3]) * F(4.0 / 16.0) + ext[2:n + 2] * F(6.0 / 16.0)).astype(F)
    return hist, F(hist.max())


def detect(gray_u8: np.ndarray, n_layers=3, contrast_thr=0.04, edge_thr=10.0, sigma=1.6, pyramids=None) -> List[dict]:
    """findScaleSpaceExtrema: keypoints (before sorting / duplicate removal / the first-octave rescale)."""
    gauss, dog = pyramids if pyramids is not None else build_pyramids(gray_u8, n_layers, sigma)
    threshold = int(math.floor(0.5 * contrast_thr / n_layers * 255))
    kpts = []
    for o in range(len(dog)):
        dogs = dog[o]
        rows, cols = dogs[0].shape
        if rows <= 2 * SIFT_IMG_BORDER or cols <= 2 * SIFT_IMG_BORDER:
            continue
        for i in range(1, n_layers + 1):
            img, prv, nxt = dogs[i], dogs[i - 1], dogs[i + 1]
            B = SIFT_IMG_BORDER
            ctr = img[B:rows - B, B:cols - B]
            cand = np.abs(ctr) > threshold
            ismax = cand & (ctr > 0); ismin = cand & (ctr < 0)
            for lay in (prv, img, nxt):
                for dy in (-1, 0, 1):
                    for dx in (-1, 0, 1):
                        nb = lay[B + dy:rows - B + dy, B + dx:cols - B + dx]
                        ismax &= ctr >= nb; ismin &= ctr <= nb
            rr, cc = np.nonzero(ismax | ismin)
            for r, c in zip(rr + B, cc + B):                                  # row-major, as the row loop visits them
                k = _adjust_local_extrema(dogs, o, i, int(r), int(c), n_layers, contrast_thr, edge_thr, sigma)
                if k is None:
                    continue
                scl_octv = F(F(k["size"] * F(0.5)) / F(1 << o))
                hist, omax = _orientation_hist(gauss[o][k["layer"]], k["c"], k["r"], int(np.rint(F(SIFT_ORI_RADIUS * scl_octv))),
                                               F(SIFT_ORI_SIG_FCTR * scl_octv))
                n = SIFT_ORI_HIST_BINS
                mag_thr = F(omax * SIFT_ORI_PEAK_RATIO)
                for j in range(n):
                    l, r2 = (j - 1) % n, (j + 1) % n
                    if hist[j] > hist[l] and hist[j] > hist[r2] and hist[j] >= mag_thr:
                        b = F(F(j) + F(F(F(0.5) * F(hist[l] - hist[r2])) / F(F(hist[l] - F(F(2) * hist[j])) + hist[r2])))
                        b = F(n + b) if b < 0 else (F(b - n) if b >= n else b)
                        ang = F(F(360.0) - F(F(360.0 / n) * b))
                        if abs(ang - F(360.0)) < FLT_EPSILON:
                            ang = F(0)
                        kpts.append(dict(x=k["x"], y=k["y"], size=k["size"], angle=ang, response=k["response"], octave=k["octave"]))
    return kpts


def sort_and_dedup(kpts: List[dict]) -> List[dict]:
    """KeyPointsFilter::removeDuplicatedSorted: order by (x, y, size desc, angle, response desc, octave desc), drop repeats
    of (x, y, size, angle); then the first-octave rescale of SIFT_Impl::detectAndCompute (pt, size *= 0.5, octave - 1)."""
    ks = sorted(kpts, key=lambda k: (float(k["x"]), float(k["y"]), -float(k["size"]), float(k["angle"]), -float(k["response"]), -k["octave"]))
    out = []
    for k in ks:
        if out and (out[-1]["x"], out[-1]["y"], out[-1]["size"], out[-1]["angle"]) == (k["x"], k["y"], k["size"], k["angle"]):
            continue
        out.append(k)
    res = []
    for k in out:
        octave = (k["octave"] & ~255) | ((k["octave"] + (-1 & 255)) & 255)      # firstOctave = -1: (octave & ~255) | ((octave + firstOctave) & 255)
        res.append(dict(x=F(k["x"] * F(0.5)), y=F(k["y"] * F(0.5)), size=F(k["size"] * F(0.5)), angle=k["angle"], response=k["response"], octave=octave))
    return res


def unpack_octave(octave: int) -> Tuple[int, int, float]:
    o = octave & 255
    layer = (octave >> 8) & 255
    o = o if o < 128 else (-128 | o)
    scale = 1.0 / (1 << o) if o >= 0 else float(1 << -o)
    return o, layer, scale


# ------------------------------------------------------------------------------------------------ descriptors
def _descriptor(img, ptx, pty, ori, scl):
    d, n = SIFT_DESCR_WIDTH, SIFT_DESCR_HIST_BINS
    rows, cols = img.shape
    px, py = int(np.rint(ptx)), int(np.rint(pty))
    cos_t = F(math.cos(float(ori) * (math.pi / 180.0))); sin_t = F(math.sin(float(ori) * (math.pi / 180.0)))
    bins_per_deg = F(n / 360.0)
    exp_scale = F(F(-1.0) / F(d * d * 0.5))
    hist_width = F(SIFT_DESCR_SCL_FCTR * scl)
    radius = int(np.rint(F(F(F(hist_width * F(1.4142135623730951)) * F(d + 1)) * F(0.5))))
    radius = min(radius, int(math.sqrt(float(cols) * cols + float(rows) * rows)))
    cos_t = F(cos_t / hist_width); sin_t = F(sin_t / hist_width)
    ii, jj = np.meshgrid(np.arange(-radius, radius + 1), np.arange(-radius, radius + 1), indexing="ij")
    fi, fj = ii.astype(F), jj.astype(F)
    c_rot = (fj * cos_t - fi * sin_t).astype(F)
    r_rot = (fj * sin_t + fi * cos_t).astype(F)
    rbin = (r_rot + F(d // 2) - F(0.5)).astype(F)
    cbin = (c_rot + F(d // 2) - F(0.5)).astype(F)
    r, c = py + ii, px + jj
    ok = (rbin > -1) & (rbin < d) & (cbin > -1) & (cbin < d) & (r > 0) & (r < rows - 1) & (c > 0) & (c < cols - 1)
    r, c, rbin, cbin, c_rot, r_rot = r[ok], c[ok], rbin[ok], cbin[ok], c_rot[ok], r_rot[ok]
    dx = (img[r, c + 1] - img[r, c - 1]).astype(F)
    dy = (img[r - 1, c] - img[r + 1, c]).astype(F)
    W = exp32(((c_rot * c_rot + r_rot * r_rot).astype(F) * exp_scale).astype(F))
    Ori = fast_atan2_deg(dy, dx)
    Mag = np.sqrt((dx * dx + dy * dy).astype(F)).astype(F)
    obin = ((Ori - F(ori)).astype(F) * bins_per_deg).astype(F)
    mag = (Mag * W).astype(F)
    r0, c0, o0 = np.floor(rbin).astype(np.int64), np.floor(cbin).astype(np.int64), np.floor(obin).astype(np.int64)
    rb, cb, ob = (rbin - r0.astype(F)).astype(F), (cbin - c0.astype(F)).astype(F), (obin - o0.astype(F)).astype(F)
    o0 = np.where(o0 < 0, o0 + n, o0); o0 = np.where(o0 >= n, o0 - n, o0)
    v_r1 = (mag * rb).astype(F); v_r0 = (mag - v_r1).astype(F)
    v_rc11 = (v_r1 * cb).astype(F); v_rc10 = (v_r1 - v_rc11).astype(F)
    v_rc01 = (v_r0 * cb).astype(F); v_rc00 = (v_r0 - v_rc01).astype(F)
    v111 = (v_rc11 * ob).astype(F); v110 = (v_rc11 - v111).astype(F)
    v101 = (v_rc10 * ob).astype(F); v100 = (v_rc10 - v101).astype(F)
    v011 = (v_rc01 * ob).astype(F); v010 = (v_rc01 - v011).astype(F)
    v001 = (v_rc00 * ob).astype(F); v000 = (v_rc00 - v001).astype(F)
    idx = ((r0 + 1) * (d + 2) + c0 + 1) * (n + 2) + o0
    hist = np.zeros((d + 2) * (d + 2) * (n + 2), F)
    # one sample adds its 8 shares in this order before the next sample is visited
    order = np.stack([idx, idx + 1, idx + (n + 2), idx + (n + 3), idx + (d + 2) * (n + 2), idx + (d + 2) * (n + 2) + 1,
                      idx + (d + 3) * (n + 2), idx + (d + 3) * (n + 2) + 1], 1).reshape(-1)
    vals = np.stack([v000, v001, v010, v011, v100, v101, v110, v111], 1).reshape(-1)
    np.add.at(hist, order, vals)
    h3 = hist.reshape(d + 2, d + 2, n + 2)
    dst = np.zeros(d * d * n, F)
    for i in range(d):
        for j in range(d):
            cell = h3[i + 1, j + 1].copy()
            cell[0] = F(cell[0] + cell[n]); cell[1] = F(cell[1] + cell[n + 1])
            dst[(i * d + j) * n:(i * d + j) * n + n] = cell[:n]
    nrm2 = F(0)
    for v in dst:
        nrm2 = F(nrm2 + F(v * v))
    thr = F(np.sqrt(nrm2) * SIFT_DESCR_MAG_THR)
    nrm2 = F(0)
    for k in range(len(dst)):
        dst[k] = min(dst[k], thr)
        nrm2 = F(nrm2 + F(dst[k] * dst[k]))
    nrm2 = F(SIFT_INT_DESCR_FCTR / max(F(np.sqrt(nrm2)), FLT_EPSILON))
    q = np.rint((dst * nrm2).astype(F))                                       # saturate_cast<uchar>
    return np.clip(q, 0, 255).astype(F)


def compute(gauss, kpts: List[dict]) -> np.ndarray:
    """calcDescriptors on the final keypoints (first octave = -1)."""
    out = np.zeros((len(kpts), 128), F)
    for k_i, k in enumerate(kpts):
        o, layer, scale = unpack_octave(k["octave"])
        size = F(k["size"] * F(scale))
        img = gauss[o + 1][layer]
        angle = F(F(360.0) - k["angle"])
        if abs(angle - F(360.0)) < FLT_EPSILON:
            angle = F(0)
        out[k_i] = _descriptor(img, F(k["x"] * F(scale)), F(k["y"] * F(scale)), angle, F(size * F(0.5)))
    return out


def detect_and_compute(gray_u8: np.ndarray):
    """cv2.SIFT_create().detectAndCompute(gray, None) -> (kp [N,2] f32, size [N], angle [N], response [N], octave [N], desc [N,128])."""
    pyr = build_pyramids(gray_u8)
    kpts = sort_and_dedup(detect(gray_u8, pyramids=pyr))
    desc = compute(pyr[0], kpts)
    g = lambda name, dt: np.array([k[name] for k in kpts], dt)  # noqa: E731
    return np.stack([g("x", F), g("y", F)], 1) if kpts else np.zeros((0, 2), F), g("size", F), g("angle", F), g("response", F), g("octave", np.int32), desc


def retain_best(kp, size, angle, response, octave, desc, n: int):
    """The keypoint cap of gn_sift_detect_and_compute (include/gisnav_amd.h): the n keypoints of largest response, ties to the
    earlier one in KeyPoint_LessThan order, listed in that order.  cv2's own cap (`SIFT_create(nfeatures)` ->
    KeyPointsFilter::retainBest, pose_node.py:108) selects by the same criterion but leaves the survivors in std::nth_element
    order and keeps EVERY keypoint that ties with the n-th response -- a documented difference in order / tie handling."""
    if len(kp) <= n:
        return kp, size, angle, response, octave, desc
    order = np.lexsort((np.arange(len(kp)), -response.astype(np.float64)))[:n]
    keep = np.sort(order)
    return kp[keep], size[keep], angle[keep], response[keep], octave[keep], desc[keep]
