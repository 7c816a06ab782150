"""ORACLE -- test infrastructure only, never imported by the product path.

CPU (numpy float64) restatement of the pose-solver half of GISNav's PoseNode hot path:

  * ``compute_pose``                     ros/gisnav/gisnav/core/_shared.py:89-125
      - DEM lookup -> 3-D points          _shared.py:95-102
      - ``cv2.solvePnPRansac(obj, img, K, zeros(4,1), useExtrinsicGuess=False,
        iterationsCount=10)``             _shared.py:104-116
      - ``cv2.Rodrigues``                 _shared.py:117
  * OpenCV 4.x calib3d [EXT, un-vendored, UNPINNED version in ros/gisnav/setup.py:116]:
      solvepnp.cpp (solvePnPRansac, PnPRansacCallback), ptsetreg.cpp
      (RANSACPointSetRegistrator, RANSACUpdateNumIters), epnp.cpp, calibration.cpp
      (cvFindExtrinsicCameraParams2, cvRodrigues2, cvProjectPoints2), compat_ptsetreg.cpp
      (CvLevMarq), fundam.cpp (findHomography method 0), levmarq.cpp (LMSolver),
      core rand.cpp (cv::RNG).

PARITY UNPINNED: cv2 is absent from this container and the reference's tests hold no
golden vector for this call.  The restatement follows the published OpenCV algorithm
(SURVEY.md Appendix B); linear algebra uses LAPACK via numpy where OpenCV uses its own
Jacobi SVD / eigen.  Where OpenCV's own result is implementation-defined the restatement picks a
deterministic representative and says so at the spot: the basis of the exactly 2-D null space of EPnP's
M^T M for 5 points (`canonical_nullspace`) and the third singular pair of a rank-deficient 3x3 alignment.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may
import this module.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import numpy as np

DBL_EPSILON = float(np.finfo(np.float64).eps)
FLT_EPSILON = float(np.finfo(np.float32).eps)
DBL_MIN = float(np.finfo(np.float64).tiny)


# ----------------------------------------------------------------------------- cv::RNG (core/rand.cpp)
class CvRNG:
    """Multiply-with-carry generator; solvePnPRansac's registrator seeds it with (uint64)-1."""
    COEFF = 4164903690

    def __init__(self, state: int = 0xFFFFFFFFFFFFFFFF):
        self.state = state if state else 0xFFFFFFFF

    def next(self) -> int:
        self.state = ((self.state & 0xFFFFFFFF) * self.COEFF + (self.state >> 32)) & 0xFFFFFFFFFFFFFFFF
        return self.state & 0xFFFFFFFF

    def uniform(self, a: int, b: int) -> int:
        return a if a == b else int(self.next() % (b - a) + a)


def get_subset(rng: CvRNG, count: int, model_points: int = 5):
    """RANSACPointSetRegistrator::getSubset -- draw distinct indices, redraw on duplicates."""
    idx = []
    for _ in range(model_points):
        v = rng.uniform(0, count)
        while v in idx:
            v = rng.uniform(0, count)
        idx.append(v)
    return idx


def ransac_update_num_iters(p: float, ep: float, model_points: int, max_iters: int) -> int:
    p = min(max(p, 0.0), 1.0)
    ep = min(max(ep, 0.0), 1.0)
    num = max(1.0 - p, DBL_MIN)
    denom = 1.0 - (1.0 - ep) ** model_points
    if denom < DBL_MIN:
        return 0
    num, denom = math.log(num), math.log(denom)
    if denom >= 0 or -num >= max_iters * (-denom):
        return max_iters
    return int(np.rint(num / denom))  # cvRound: round-half-even


# ----------------------------------------------------------------------------- Rodrigues (calibration.cpp)
def rodrigues_vec2mat(r: np.ndarray, jac: bool = False):
    r = np.asarray(r, np.float64).reshape(3)
    theta = float(np.linalg.norm(r))
    if theta < DBL_EPSILON:
        R = np.eye(3)
        if not jac:
            return R
        J = np.zeros((3, 9))
        J[0, 5], J[0, 7], J[1, 2], J[1, 6], J[2, 1], J[2, 3] = -1, 1, 1, -1, -1, 1
        return R, J
    c, s = math.cos(theta), math.sin(theta)
    c1, itheta = 1.0 - c, 1.0 / theta
    rx, ry, rz = r * itheta
    rrt = np.array([[rx * rx, rx * ry, rx * rz], [rx * ry, ry * ry, ry * rz], [rx * rz, ry * rz, rz * rz]])
    r_x = np.array([[0, -rz, ry], [rz, 0, -rx], [-ry, rx, 0]])
    R = c * np.eye(3) + c1 * rrt + s * r_x
    if not jac:
        return R
    eye = np.eye(3).reshape(9)
    drrt = np.array([[rx + rx, ry, rz, ry, 0, 0, rz, 0, 0],
                     [0, rx, 0, rx, ry + ry, rz, 0, rz, 0],
                     [0, 0, rx, 0, 0, ry, rx, ry, rz + rz]], np.float64)
    d_r_x = np.array([[0, 0, 0, 0, 0, -1, 0, 1, 0],
                      [0, 0, 1, 0, 0, 0, -1, 0, 0],
                      [0, -1, 0, 1, 0, 0, 0, 0, 0]], np.float64)
    J = np.empty((3, 9))
    for i, ri in enumerate((rx, ry, rz)):
        a0, a1, a2 = -s * ri, (s - 2 * c1 * itheta) * ri, c1 * itheta
        a3, a4 = (c - s * itheta) * ri, s * itheta
        J[i] = a0 * eye + a1 * rrt.reshape(9) + a2 * drrt[i] + a3 * r_x.reshape(9) + a4 * d_r_x[i]
    return R, J


def rodrigues_mat2vec(Rm: np.ndarray) -> np.ndarray:
    U, _, Vt = np.linalg.svd(np.asarray(Rm, np.float64).reshape(3, 3))
    R = U @ Vt
    r = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = math.sqrt(float(r @ r) * 0.25)
    c = (R[0, 0] + R[1, 1] + R[2, 2] - 1) * 0.5
    c = min(max(c, -1.0), 1.0)
    theta = math.acos(c)
    if s < 1e-5:
        if c > 0:
            return np.zeros(3)
        rx = math.sqrt(max((R[0, 0] + 1) * 0.5, 0.0))
        ry = math.sqrt(max((R[1, 1] + 1) * 0.5, 0.0)) * (-1.0 if R[0, 1] < 0 else 1.0)
        rz = math.sqrt(max((R[2, 2] + 1) * 0.5, 0.0)) * (-1.0 if R[0, 2] < 0 else 1.0)
        if abs(rx) < abs(ry) and abs(rx) < abs(rz) and ((R[1, 2] > 0) != (ry * rz > 0)):
            rz = -rz
        v = np.array([rx, ry, rz])
        return v * (theta / np.linalg.norm(v))
    return r * (theta / (2 * s))


# ----------------------------------------------------------------------------- projectPoints (zero distortion)
def project_points(obj: np.ndarray, rvec, tvec, A: np.ndarray, jac: bool = False):
    """cvProjectPoints2 with distCoeffs = 0.  obj (n,3) f64 -> (n,2) f64 [, d/d(r,t) (2n,6)]."""
    t = np.asarray(tvec, np.float64).reshape(3)
    if jac:
        R, dRdr = rodrigues_vec2mat(rvec, True)
    else:
        R = rodrigues_vec2mat(rvec)
    fx, fy, cx, cy = A[0, 0], A[1, 1], A[0, 2], A[1, 2]
    X = obj @ R.T + t
    with np.errstate(divide="ignore"):
        z = np.where(X[:, 2] != 0, 1.0 / X[:, 2], 1.0)
    x, y = X[:, 0] * z, X[:, 1] * z
    proj = np.column_stack([x * fx + cx, y * fy + cy])
    if not jac:
        return proj
    n = obj.shape[0]
    J = np.empty((2 * n, 6))
    # d(x,y)/dX
    dxdX = np.column_stack([z, np.zeros(n), -x * z])
    dydX = np.column_stack([np.zeros(n), z, -y * z])
    J[0::2, 3:6] = fx * dxdX
    J[1::2, 3:6] = fy * dydX
    for i in range(3):
        dR = dRdr[i].reshape(3, 3)
        dX = obj @ dR.T  # (n,3)
        J[0::2, i] = fx * np.einsum("nk,nk->n", dxdX, dX)
        J[1::2, i] = fy * np.einsum("nk,nk->n", dydX, dX)
    return proj, J


# ----------------------------------------------------------------------------- EPnP (epnp.cpp)
_PAIRS = ((0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3))


def canonical_nullspace(v0: np.ndarray, v1: np.ndarray):
    """With 5 points M is 10 x 12, so the two smallest right-singular vectors span an exactly
    2-dimensional null space and their individual directions are an artefact of the SVD code path
    (OpenCV: the sweep order of its one-sided Jacobi).  EPnP's approx_1 / approx_3 initialisations
    depend on that artefact, so the restatement fixes the basis deterministically: v0 is the normalised
    projection of the coordinate axis e_11 onto the null space, v1 its in-plane orthogonal complement
    (sign: component 10 non-negative).  Any orthonormal basis is an equally valid stand-in for OpenCV's."""
    a, b = v0[11], v1[11]
    nrm = math.hypot(a, b)
    if nrm < 1e-12:
        return v0, v1
    w0 = (a * v0 + b * v1) / nrm
    w1 = (-b * v0 + a * v1) / nrm
    if w1[10] < 0:
        w1 = -w1
    return w0, w1


_NULLSPACE_BASIS = canonical_nullspace


def canonical_axis_signs(u: np.ndarray) -> np.ndarray:
    """The sign of each principal axis returned by an SVD is implementation-defined, and it changes
    where EPnP puts its control points (c_i = c_0 +- k_i u_i), i.e. the parametrisation in which the
    basis-dependent beta initialisations are computed.  Fix it: every axis (column) gets the sign that
    makes its largest-magnitude component positive (first such component on ties)."""
    u = u.copy()
    for i in range(u.shape[1]):
        k = int(np.argmax(np.abs(u[:, i])))
        if u[k, i] < 0:
            u[:, i] = -u[:, i]
    return u


def _svd_solve(A, b):
    """cvSolve(A, b, x, CV_SVD) as epnp.cpp's find_betas_approx_{1,2,3} call it: minimum-norm least squares through the SVD,
    singular values <= 2 * DBL_EPSILON * sum(w) treated as zero (cv::SVD::backSubst)."""
    U, w, Vt = np.linalg.svd(A, full_matrices=False)
    thr = 2.0 * DBL_EPSILON * float(np.sum(w))
    x = np.zeros(A.shape[1])
    for k in range(len(w)):
        if abs(w[k]) > thr:
            x += Vt[k] * (float(U[:, k] @ b) / w[k])
    return x


def _qr_solve(A, b, x_prev):
    """epnp.cpp's own qr_solve (used by gauss_newton): Householder QR WITHOUT pivoting and without any rank test, columns scaled by
    their largest entry (the scan that finds it skips the last row, as in the original).  A column that is exactly zero makes
    the routine return early and leaves X as it was (the caller then adds the previous step again)."""
    A = np.array(A, np.float64)
    b = np.array(b, np.float64)
    nr, nc = A.shape
    A1, A2 = np.zeros(nc), np.zeros(nc)
    for k in range(nc):
        eta = abs(A[k, k])
        for i in range(k + 1, nr):           # original: compares rows k .. nr-2 (the pointer is advanced after the compare)
            eta = max(eta, abs(A[i - 1, k]))
        if eta == 0:
            return x_prev.copy()
        sum2 = 0.0
        for i in range(k, nr):
            A[i, k] *= 1.0 / eta
            sum2 += A[i, k] * A[i, k]
        sigma = math.sqrt(sum2)
        if A[k, k] < 0:
            sigma = -sigma
        A[k, k] += sigma
        A1[k] = sigma * A[k, k]
        A2[k] = -eta * sigma
        for j in range(k + 1, nc):
            tau = float(A[k:, k] @ A[k:, j]) / A1[k]
            A[k:, j] -= tau * A[k:, k]
    for j in range(nc):
        tau = float(A[j:, j] @ b[j:]) / A1[j]
        b[j:] -= tau * A[j:, j]
    x = np.zeros(nc)
    with np.errstate(all="ignore"):
        x[nc - 1] = b[nc - 1] / A2[nc - 1]
        for i in range(nc - 2, -1, -1):
            x[i] = (b[i] - float(A[i, i + 1:] @ x[i + 1:])) / A2[i]
    return x


def epnp(pws: np.ndarray, us: np.ndarray, fu=1.0, fv=1.0, uc=0.0, vc=0.0, dbg: Optional[dict] = None) -> Tuple[np.ndarray, np.ndarray]:
    """EPnP (Lepetit et al.) as in OpenCV's epnp.cpp.  pws (n,3), us (n,2) -> R (3,3), t (3,)."""
    n = pws.shape[0]
    # choose_control_points
    cws = np.empty((4, 3))
    cws[0] = pws.mean(axis=0)
    pw0 = pws - cws[0]
    u_, dc, _ = np.linalg.svd(pw0.T @ pw0)
    u_ = canonical_axis_signs(u_)
    for i in range(3):
        cws[i + 1] = cws[0] + math.sqrt(dc[i] / n) * u_[:, i]
    # barycentric coordinates
    cc = (cws[1:] - cws[0]).T
    cc_inv = np.linalg.pinv(cc)
    al = (pws - cws[0]) @ cc_inv.T
    alphas = np.column_stack([1.0 - al.sum(axis=1), al])
    # M (2n x 12)
    M = np.zeros((2 * n, 12))
    for j in range(4):
        M[0::2, 3 * j] = alphas[:, j] * fu
        M[0::2, 3 * j + 2] = alphas[:, j] * (uc - us[:, 0])
        M[1::2, 3 * j + 1] = alphas[:, j] * fv
        M[1::2, 3 * j + 2] = alphas[:, j] * (vc - us[:, 1])
    _, _, ut = np.linalg.svd(M.T @ M)  # rows of ut: right singular vectors, descending
    v = [ut[11 - k] for k in range(4)]
    if n == 5 and _NULLSPACE_BASIS is not None:
        v[0], v[1] = _NULLSPACE_BASIS(v[0], v[1])
    # L (6x10) and rho
    dv = np.array([[v[i][3 * a:3 * a + 3] - v[i][3 * b:3 * b + 3] for (a, b) in _PAIRS] for i in range(4)])
    L = np.empty((6, 10))
    for j in range(6):
        d0, d1, d2, d3 = dv[0][j], dv[1][j], dv[2][j], dv[3][j]
        L[j] = [d0 @ d0, 2 * (d0 @ d1), d1 @ d1, 2 * (d0 @ d2), 2 * (d1 @ d2), d2 @ d2,
                2 * (d0 @ d3), 2 * (d1 @ d3), 2 * (d2 @ d3), d3 @ d3]
    rho = np.array([np.sum((cws[a] - cws[b]) ** 2) for (a, b) in _PAIRS])

    def approx1():
        b4 = _svd_solve(L[:, [0, 1, 3, 6]], rho)
        if b4[0] < 0:
            b0 = math.sqrt(-b4[0])
            return np.array([b0, -b4[1] / b0, -b4[2] / b0, -b4[3] / b0])
        b0 = math.sqrt(b4[0])
        return np.array([b0, b4[1] / b0, b4[2] / b0, b4[3] / b0])

    def approx2():
        b3 = _svd_solve(L[:, [0, 1, 2]], rho)
        if b3[0] < 0:
            b0 = math.sqrt(-b3[0])
            b1 = math.sqrt(-b3[2]) if b3[2] < 0 else 0.0
        else:
            b0 = math.sqrt(b3[0])
            b1 = math.sqrt(b3[2]) if b3[2] > 0 else 0.0
        if b3[1] < 0:
            b0 = -b0
        return np.array([b0, b1, 0.0, 0.0])

    def approx3():
        b5 = _svd_solve(L[:, [0, 1, 2, 3, 4]], rho)
        if b5[0] < 0:
            b0 = math.sqrt(-b5[0])
            b1 = math.sqrt(-b5[2]) if b5[2] < 0 else 0.0
        else:
            b0 = math.sqrt(b5[0])
            b1 = math.sqrt(b5[2]) if b5[2] > 0 else 0.0
        if b5[1] < 0:
            b0 = -b0
        return np.array([b0, b1, b5[3] / b0, 0.0])

    def gauss_newton(betas):
        betas = betas.copy()
        step = np.zeros(4)
        for _ in range(5):
            b0, b1, b2, b3 = betas
            A = np.column_stack([
                2 * L[:, 0] * b0 + L[:, 1] * b1 + L[:, 3] * b2 + L[:, 6] * b3,
                L[:, 1] * b0 + 2 * L[:, 2] * b1 + L[:, 4] * b2 + L[:, 7] * b3,
                L[:, 3] * b0 + L[:, 4] * b1 + 2 * L[:, 5] * b2 + L[:, 8] * b3,
                L[:, 6] * b0 + L[:, 7] * b1 + L[:, 8] * b2 + 2 * L[:, 9] * b3])
            bb = rho - (L[:, 0] * b0 * b0 + L[:, 1] * b0 * b1 + L[:, 2] * b1 * b1 + L[:, 3] * b0 * b2
                        + L[:, 4] * b1 * b2 + L[:, 5] * b2 * b2 + L[:, 6] * b0 * b3 + L[:, 7] * b1 * b3
                        + L[:, 8] * b2 * b3 + L[:, 9] * b3 * b3)
            step = _qr_solve(A, bb, step)
            betas = betas + step
        return betas

    def r_and_t(betas):
        ccs = np.zeros((4, 3))
        for k in range(4):
            ccs += betas[k] * v[k].reshape(4, 3)
        pcs = alphas @ ccs
        if pcs[0, 2] < 0:
            ccs, pcs = -ccs, -pcs
        pc0, pw0_ = pcs.mean(axis=0), pws.mean(axis=0)
        abt = (pcs - pc0).T @ (pws - pw0_)
        U, sv, Vt = np.linalg.svd(abt)
        if sv[2] <= 1e-9 * sv[0]:
            # rank-deficient alignment (coplanar world points): OpenCV fills the third singular pair from
            # a seeded random vector; the restatement completes it as u0 x u1, v0 x v1 (proper rotation)
            U = np.column_stack([U[:, 0], U[:, 1], np.cross(U[:, 0], U[:, 1])])
            Vt = np.vstack([Vt[0], Vt[1], np.cross(Vt[0], Vt[1])])
        R = U @ Vt
        if np.linalg.det(R) < 0:
            R[2] = -R[2]
        t = pc0 - R @ pw0_
        Xc = pws @ R.T + t
        ue = uc + fu * Xc[:, 0] / Xc[:, 2]
        ve = vc + fv * Xc[:, 1] / Xc[:, 2]
        err = np.sqrt((us[:, 0] - ue) ** 2 + (us[:, 1] - ve) ** 2).sum() / n
        return err, R, t

    if dbg is not None:
        dbg.update(L=L, rho=rho, v=v, alphas=alphas, cws=cws, cands=[])
    best = None
    for f in (approx1, approx2, approx3):
        with np.errstate(all="ignore"):
            b_init = f()
            b_fin = gauss_newton(b_init)
            cand = r_and_t(b_fin)
        if dbg is not None:
            dbg["cands"].append((b_init, b_fin, cand[0]))
        if best is None or cand[0] < best[0]:  # strict '<': first wins ties (epnp.cpp compute_pose)
            best = cand
    return best[1], best[2]


# ----------------------------------------------------------------------------- CvLevMarq pose refinement
def _levmarq_pose(obj, img, A, r0, t0, max_iter=20, eps=FLT_EPSILON):
    """cvFindExtrinsicCameraParams2's refinement loop on CvLevMarq(6, 2n, eps=FLT_EPSILON, 20 iters)."""
    param = np.concatenate([r0, t0]).astype(np.float64)
    prev_param = param.copy()
    lambda_lg10 = -3
    iters = 0
    prev_err_norm = 0.0
    img_flat = img.reshape(-1)

    def residual(p, jac):
        if jac:
            proj, J = project_points(obj, p[:3], p[3:], A, True)
            return proj.reshape(-1) - img_flat, J
        return project_points(obj, p[:3], p[3:], A).reshape(-1) - img_flat, None

    def step(JtJ, JtErr):
        lam = math.exp(lambda_lg10 * math.log(10.0))
        M = JtJ.copy()
        M[np.diag_indices(6)] *= 1.0 + lam
        return prev_param - np.linalg.lstsq(M, JtErr, rcond=None)[0]  # solve(..., DECOMP_SVD)

    # STARTED -> CALC_J
    err, J = residual(param, True)
    while True:
        # CALC_J
        JtJ, JtErr = J.T @ J, J.T @ err
        prev_param = param.copy()
        param = step(JtJ, JtErr)
        if iters == 0:
            prev_err_norm = float(np.linalg.norm(err))
        # CHECK_ERR (possibly repeated with growing lambda)
        while True:
            err, _ = residual(param, False)
            err_norm = float(np.linalg.norm(err))
            if err_norm > prev_err_norm:
                lambda_lg10 += 1
                if lambda_lg10 <= 16:
                    param = step(JtJ, JtErr)
                    continue
            break
        lambda_lg10 = max(lambda_lg10 - 1, -16)
        iters += 1
        denom = float(np.linalg.norm(prev_param))
        rel = float(np.linalg.norm(param - prev_param)) / (denom if denom > 0 else 1.0)  # CV_RELATIVE_L2
        if iters >= max_iter or rel < eps:
            break
        prev_err_norm = err_norm
        err, J = residual(param, True)
    return param[:3], param[3:]


# ----------------------------------------------------------------------------- findHomography(method=0)
def _homography_dlt(M: np.ndarray, m: np.ndarray) -> Optional[np.ndarray]:
    """HomographyEstimatorCallback::runKernel (normalised DLT, eigen of L^T L)."""
    n = len(M)
    cM, cm = M.mean(axis=0), m.mean(axis=0)
    sM, sm = np.abs(M - cM).sum(axis=0), np.abs(m - cm).sum(axis=0)
    if np.any(np.abs(sM) < DBL_EPSILON) or np.any(np.abs(sm) < DBL_EPSILON):
        return None
    sm, sM = n / sm, n / sM
    inv_hnorm = np.array([[1 / sm[0], 0, cm[0]], [0, 1 / sm[1], cm[1]], [0, 0, 1]])
    hnorm2 = np.array([[sM[0], 0, -cM[0] * sM[0]], [0, sM[1], -cM[1] * sM[1]], [0, 0, 1]])
    x, y = (m[:, 0] - cm[0]) * sm[0], (m[:, 1] - cm[1]) * sm[1]
    X, Y = (M[:, 0] - cM[0]) * sM[0], (M[:, 1] - cM[1]) * sM[1]
    one, zero = np.ones(n), np.zeros(n)
    Lx = np.column_stack([X, Y, one, zero, zero, zero, -x * X, -x * Y, -x])
    Ly = np.column_stack([zero, zero, zero, X, Y, one, -y * X, -y * Y, -y])
    LtL = Lx.T @ Lx + Ly.T @ Ly
    w, V = np.linalg.eigh(LtL)
    H0 = V[:, 0].reshape(3, 3)  # smallest eigenvalue
    H = inv_hnorm @ H0 @ hnorm2
    return H / H[2, 2]


def _homography_refine(H: np.ndarray, M: np.ndarray, m: np.ndarray, max_iters: int = 10) -> np.ndarray:
    """LMSolver (levmarq.cpp) on HomographyRefineCallback: 8 free parameters (h[8] fixed = 1)."""
    def compute(h, jac):
        Mx, My = M[:, 0], M[:, 1]
        ww = h[6] * Mx + h[7] * My + 1.0
        ww = np.where(np.abs(ww) > DBL_EPSILON, 1.0 / ww, 0.0)
        xi = (h[0] * Mx + h[1] * My + h[2]) * ww
        yi = (h[3] * Mx + h[4] * My + h[5]) * ww
        r = np.empty(2 * len(M))
        r[0::2], r[1::2] = xi - m[:, 0], yi - m[:, 1]
        if not jac:
            return r, None
        J = np.zeros((2 * len(M), 8))
        J[0::2, 0], J[0::2, 1], J[0::2, 2] = Mx * ww, My * ww, ww
        J[0::2, 6], J[0::2, 7] = -Mx * ww * xi, -My * ww * xi
        J[1::2, 3], J[1::2, 4], J[1::2, 5] = Mx * ww, My * ww, ww
        J[1::2, 6], J[1::2, 7] = -Mx * ww * yi, -My * ww * yi
        return r, J

    x = H.reshape(9)[:8].copy()
    r, J = compute(x, True)
    S = float(r @ r)
    A, v = J.T @ J, J.T @ r
    D = np.diag(A).copy()
    Rlo, Rhi = 0.25, 0.75
    lam, lc = 1.0, 0.75
    it = 0
    while True:
        Ap = A + np.diag(lam * D)
        d = np.linalg.lstsq(Ap, v, rcond=None)[0]
        xd = x - d
        rd, _ = compute(xd, False)
        Sd = float(rd @ rd)
        dS = float(d @ (2 * v - A @ d))
        Rr = (S - Sd) / (dS if abs(dS) > DBL_EPSILON else 1.0)
        if Rr > Rhi:
            lam *= 0.5
            if lam < lc:
                lam = 0.0
        elif Rr < Rlo:
            tt = float(d @ v)
            nu = (Sd - S) / (tt if abs(tt) > DBL_EPSILON else 1.0) + 2
            nu = min(max(nu, 2.0), 10.0)
            if lam == 0:
                Ainv = np.linalg.pinv(A)
                lam = lc = 1.0 / max(DBL_EPSILON, float(np.max(np.abs(np.diag(Ainv)))))
                nu *= 0.5
            lam *= nu
        if Sd < S:
            S, x = Sd, xd
            r, J = compute(x, True)
            A, v = J.T @ J, J.T @ r
        it += 1
        if not (it < max_iters and np.max(np.abs(d)) >= DBL_EPSILON and np.max(np.abs(r)) >= DBL_EPSILON):
            break
    return np.append(x, 1.0).reshape(3, 3)


def find_homography_ls(src: np.ndarray, dst: np.ndarray) -> Optional[np.ndarray]:
    """cv::findHomography(src, dst, method=0): points are converted to float32 first."""
    M = src.astype(np.float32).astype(np.float64)
    m = dst.astype(np.float32).astype(np.float64)
    H = _homography_dlt(M, m)
    if H is None:
        return None
    if len(M) > 4:
        H = _homography_refine(H, M, m)
    return H


# ----------------------------------------------------------------------------- solvePnP(SOLVEPNP_ITERATIVE)
def solve_pnp_iterative(obj: np.ndarray, img: np.ndarray, A: np.ndarray):
    """cvFindExtrinsicCameraParams2 with useExtrinsicGuess = False, zero distortion."""
    obj = np.asarray(obj, np.float64)
    img = np.asarray(img, np.float64)
    n = len(obj)
    mn = np.column_stack([(img[:, 0] - A[0, 2]) / A[0, 0], (img[:, 1] - A[1, 2]) / A[1, 1]])
    Mc = obj.mean(axis=0)
    MM = (obj - Mc).T @ (obj - Mc)
    _, W, Vt = np.linalg.svd(MM)
    if W[2] / W[1] < 1e-3:  # planar structure
        R_tr = Vt.copy()
        if Vt[0, 2] ** 2 + Vt[1, 2] ** 2 < 1e-10:
            R_tr = np.eye(3)
        if np.linalg.det(R_tr) < 0:
            R_tr = -R_tr
        T_tr = -R_tr @ Mc
        Mxy = (obj @ R_tr.T + T_tr)[:, :2]
        H = find_homography_ls(Mxy, mn)
        if H is not None and np.all(np.isfinite(H)):
            h1n, h2n = np.linalg.norm(H[:, 0]), np.linalg.norm(H[:, 1])
            h1 = H[:, 0] / max(h1n, DBL_EPSILON)
            h2 = H[:, 1] / max(h2n, DBL_EPSILON)
            t = H[:, 2] * (2.0 / max(h1n + h2n, DBL_EPSILON))
            Rh = np.column_stack([h1, h2, np.cross(h1, h2)])
            Rh = rodrigues_vec2mat(rodrigues_mat2vec(Rh))
            t = t + Rh @ T_tr
            R = Rh @ R_tr
        else:
            R, t = np.eye(3), np.zeros(3)
        r = rodrigues_mat2vec(R)
    else:  # DLT
        if n < 6:
            raise ValueError("DLT algorithm needs at least 6 points")
        L = np.zeros((2 * n, 12))
        x, y = -mn[:, 0], -mn[:, 1]
        L[0::2, 0:3], L[0::2, 3] = obj, 1.0
        L[0::2, 8:11], L[0::2, 11] = x[:, None] * obj, x
        L[1::2, 4:7], L[1::2, 7] = obj, 1.0
        L[1::2, 8:11], L[1::2, 11] = y[:, None] * obj, y
        _, _, LV = np.linalg.svd(L.T @ L)
        RRt = LV[11].reshape(3, 4).copy()
        if np.linalg.det(RRt[:, :3]) < 0:
            RRt = -RRt
        sc = np.linalg.norm(RRt[:, :3])
        U, _, Vt2 = np.linalg.svd(RRt[:, :3])
        R = U @ Vt2
        t = RRt[:, 3] * (np.linalg.norm(R) / sc)
        r = rodrigues_mat2vec(R)
    return _levmarq_pose(obj, img, A, r, t)


# ----------------------------------------------------------------------------- P3P (npoints == 4)
def _p3p_lengths(distances, cosines):
    """OpenCV p3p.cpp `solve_for_lengths` (Gao, Hou, Tang, Cheng 2003): the quartic in x = |OP0| / |OP2| and the rational y(x).
    distances = (|P1P2|, |P0P2|, |P0P1|), cosines = (cos(f1, f2), cos(f0, f2), cos(f0, f1)) of the unit bearing vectors.
    Returns up to 4 length triples (|OP0|, |OP1|, |OP2|).  The quartic's real roots come from numpy's companion-matrix solver (OpenCV: Ferrari)."""
    p, q, r = 2 * cosines[0], 2 * cosines[1], 2 * cosines[2]
    inv_d22 = 1.0 / (distances[2] * distances[2])
    a = inv_d22 * distances[0] * distances[0]
    b = inv_d22 * distances[1] * distances[1]
    a2, b2, p2, q2, r2 = a * a, b * b, p * p, q * q, r * r
    pr, pqr = p * r, q * p * r
    if p2 + q2 + r2 - pqr - 1 == 0:
        return []
    ab, a_2, a_4 = a * b, 2 * a, 4 * a
    A = -2 * b + b2 + a2 + 1 + ab * (2 - r2) - a_2
    if A == 0:
        return []
    B = q * (-2 * (ab + a2 + 1 - b) + r2 * ab + a_4) + pr * (b - b2 + ab)
    C = q2 + b2 * (r2 + p2 - 2) - b * (p2 + pqr) - ab * (r2 + pqr) + (a2 - a_2) * (2 + q2) + 2
    D = pr * (ab - b2 + b) + q * ((p2 - 2) * b + 2 * (ab - a2) + a_4 - 2)
    E = 1 + 2 * (b - a - ab) + b2 - b * p2 + a2
    temp = p2 * (a - 1 + b) + r2 * (a - 1 - b) + pqr - a * pqr
    b0 = b * temp * temp
    if b0 == 0:
        return []
    roots = np.roots([A, B, C, D, E])
    real = []
    bm, cm, dm, em = B / A, C / A, D / A, E / A
    for z in roots:
        if abs(z.imag) <= 1e-9 * max(1.0, abs(z.real)):
            x = float(z.real)
            for _ in range(4):                                   # Newton polish on the real axis (monic form), as the kernel does
                pv = (((x + bm) * x + cm) * x + dm) * x + em
                dv = ((4.0 * x + 3.0 * bm) * x + 2.0 * cm) * x + dm
                if dv == 0.0:
                    break
                x -= pv / dv
            real.append(x)
    real.sort()
    r3, pr2 = r2 * r, p * r2
    r3q = r3 * q
    out = []
    for x in real:
        if x <= 0:
            continue
        x2 = x * x
        b1 = ((1 - a - b) * x2 + (q * a - q) * x + 1 - a + b) * (
            ((r3 * (a2 + ab * (2 - r2) - a_2 + b2 - 2 * b + 1)) * x
             + (r3q * (2 * (b - a2) + a_4 + ab * (r2 - 2) - 2) + pr2 * (1 + a2 + 2 * (ab - a - b) + r2 * (b - b2) + b2))) * x2
            + (r3 * (q2 * (1 - 2 * a + a2) + r2 * (b2 - ab) - a_4 + 2 * (a2 - b2) + 2) + r * p2 * (b2 + 2 * (ab - b - a) + 1 + a2)
               + pr2 * q * (a_4 + 2 * (b - ab - a2) - 2 - r2 * b)) * x
            + 2 * r3q * (a_2 - b - a2 + ab - 1) + pr2 * (q2 - a_4 + 2 * (a2 - b2) + r2 * b + q2 * (a2 - a_2) + 2)
            + p2 * (p * (2 * (ab - a - b) + a2 + b2 + 1) + 2 * q * r * (b + a_2 - a2 - ab - 1)))
        if b1 <= 0:
            continue
        y = b1 / b0
        v = x2 + y * y - x * y * r
        if v <= 0:
            continue
        Z = distances[2] / np.sqrt(v)
        out.append((x * Z, y * Z, Z))
    return out


def _triad(p0, p1, p2):
    e1 = (p1 - p0) / np.linalg.norm(p1 - p0)
    e3 = np.cross(e1, p2 - p0)
    e3 = e3 / np.linalg.norm(e3)
    return np.stack([e1, np.cross(e3, e1), e3], axis=1)


def solve_p3p(obj: np.ndarray, und: np.ndarray):
    """cv::solvePnP(SOLVEPNP_P3P) on exactly 4 correspondences (what solvePnPRansac does when npoints == 4, _shared.py:109-116): the first
    three points give up to four poses (Gao's P3P), the fourth picks the one with the smallest reprojection error.  obj (4,3) f64,
    und (4,2) normalised image points.  The rigid alignment of the three reconstructed camera-frame points with the object points is done
    with orthonormal triads (OpenCV: Horn's quaternion least squares -- identical for congruent triangles, which these are to round-off).
    Returns (R, t) or None."""
    f = np.column_stack([und[:3], np.ones(3)])
    f = f / np.linalg.norm(f, axis=1, keepdims=True)
    d = (np.linalg.norm(obj[1] - obj[2]), np.linalg.norm(obj[0] - obj[2]), np.linalg.norm(obj[0] - obj[1]))
    c = (float(f[1] @ f[2]), float(f[0] @ f[2]), float(f[0] @ f[1]))
    if min(d) == 0:
        return None
    best = None
    for L in _p3p_lengths(d, c):
        M = f * np.asarray(L)[:, None]
        Fo, Fc = _triad(obj[0], obj[1], obj[2]), _triad(M[0], M[1], M[2])
        R = Fc @ Fo.T
        t = M[0] - R @ obj[0]
        pc = R @ obj[3] + t
        err = (pc[0] / pc[2] - und[3, 0]) ** 2 + (pc[1] / pc[2] - und[3, 1]) ** 2 if pc[2] != 0 else np.inf
        if best is None or err < best[0]:
            best = (err, R, t)
    return None if best is None else (best[1], best[2])


# ----------------------------------------------------------------------------- solvePnPRansac
def solve_pnp_ransac(obj: np.ndarray, img: np.ndarray, A: np.ndarray, iterations_count: int = 10,
                     reproj_error: float = 8.0, confidence: float = 0.99):
    """cv2.solvePnPRansac(obj f32 (K,3), img f32 (K,2), A f64, dist=0, False, iterations_count).

    Returns (ok, rvec (3,1) f64, tvec (3,1) f64, inliers (n,) int or None).
    K == 4: OpenCV makes ONE solvePnP(SOLVEPNP_P3P) call, every point an inlier, no refinement (`solve_p3p`); unreachable from PoseNode
    (MIN_MATCHES = 15, pose_node.py:299-303) and TwistNode (30), reachable through compute_pose.  K < 4: cv2 raises; here (False, ...).
    K == 5: the same `if (model_points == npoints)` block of solvepnp.cpp: ONE solvePnP(SOLVEPNP_EPNP) on all five points, every point an
    inlier, no RANSAC loop and no ITERATIVE refinement.
    EPnP receives the camera matrix: epnp::init_points re-applies the intrinsics to the undistorted (normalised, float32-stored) points,
    us = x fu + uc, so the rows of M carry fu / fv and the candidate-selection error is in pixels -- equivalent to working on normalised
    points only when fx == fy.
    """
    obj = np.asarray(obj, np.float32)
    img = np.asarray(img, np.float32)
    count = len(obj)
    model_points = 5
    A = np.asarray(A, np.float64).reshape(3, 3)
    obj64, img64 = obj.astype(np.float64), img.astype(np.float64)
    if count == 4:
        und4 = np.column_stack([(img64[:, 0] - A[0, 2]) / A[0, 0], (img64[:, 1] - A[1, 2]) / A[1, 1]]).astype(np.float32).astype(np.float64)
        sol = solve_p3p(obj64, und4)
        if sol is None:
            return False, None, None, None
        r = rodrigues_mat2vec(sol[0])
        return True, r.reshape(3, 1), np.asarray(sol[1]).reshape(3, 1), np.arange(4)
    if count < model_points:
        return False, None, None, None
    fu, fv, uc, vc = float(A[0, 0]), float(A[1, 1]), float(A[0, 2]), float(A[1, 2])
    # solvePnP(SOLVEPNP_EPNP) runs cv::undistortPoints on the subset; its output Mat takes the INPUT's depth, so for the float32 image
    # points PoseNode passes, epnp reads normalised coordinates that were computed in double and stored as float32
    und = np.column_stack([(img64[:, 0] - A[0, 2]) / A[0, 0], (img64[:, 1] - A[1, 2]) / A[1, 1]]).astype(np.float32).astype(np.float64)
    us_px = np.column_stack([und[:, 0] * fu + uc, und[:, 1] * fv + vc])      # epnp::init_points
    if count == model_points:      # solvePnPRansac's `model_points == npoints` early return
        try:
            R, t = epnp(obj64, us_px, fu, fv, uc, vc)
            rvec = rodrigues_mat2vec(R)
            ok = bool(np.all(np.isfinite(rvec)) and np.all(np.isfinite(t)))
        except (np.linalg.LinAlgError, ValueError, ZeroDivisionError):
            ok = False
        if not ok:
            return False, None, None, None
        return True, rvec.reshape(3, 1), np.asarray(t).reshape(3, 1), np.arange(count)
    rng = CvRNG(0xFFFFFFFFFFFFFFFF)
    niters = iterations_count
    max_good = 0
    best_mask, best_model = None, None
    thr = np.float32(reproj_error * reproj_error)
    it = 0
    while it < niters:
        idx = get_subset(rng, count, model_points)
        try:
            R, t = epnp(obj64[idx], us_px[idx], fu, fv, uc, vc)
            rvec = rodrigues_mat2vec(R)
            ok = bool(np.all(np.isfinite(rvec)) and np.all(np.isfinite(t)))
        except (np.linalg.LinAlgError, ValueError, ZeroDivisionError):
            ok = False
        if ok:
            proj = project_points(obj64, rvec, t, A).astype(np.float32)
            diff = img - proj
            err = (diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1]).astype(np.float32)
            mask = err <= thr
            good = int(mask.sum())
            if good > max(max_good, model_points - 1):
                best_mask, best_model, max_good = mask, (rvec, t), good
                niters = ransac_update_num_iters(confidence, (count - good) / count, model_points, niters)
        it += 1
    if best_mask is None:
        return False, None, None, None
    inl = np.nonzero(best_mask)[0]
    try:
        r, t = solve_pnp_iterative(obj64[inl], img64[inl], A)
    except ValueError:  # OpenCV >= 4.5 catches the DLT "< 6 points" exception: fall back to RANSAC model
        r, t = best_model
    return True, r.reshape(3, 1), np.asarray(t).reshape(3, 1), inl


# ----------------------------------------------------------------------------- _shared.compute_pose
def compute_pose(k_flat, mkp_qry: np.ndarray, mkp_ref: np.ndarray, elevation: Optional[np.ndarray]):
    """_shared.py:89-125.  Returns (R (3,3) f64, t (3,1) f64) or None when RANSAC fails
    (the reference would raise on cv2.Rodrigues(None) there; the shim reports None)."""
    if elevation is None:
        obj = np.hstack((mkp_ref, np.zeros((len(mkp_ref), 1))))
    else:
        x, y = np.transpose(np.floor(mkp_ref).astype(int))
        obj = np.hstack((mkp_ref, elevation[y, x].reshape(-1, 1)))
    ok, r, t, _ = solve_pnp_ransac(obj.astype(np.float32), mkp_qry, np.asarray(k_flat, np.float64).reshape(3, 3), 10)
    if not ok:
        return None
    return rodrigues_vec2mat(r), t
