"""CPU restatement (numpy, fp64) of the pose solver behind the reference's ``ransac_PnP``
(src/utils/eval_utils.py:18-42): ``cv2.solvePnPRansac(pts_3d, pts_2d, K, dist=0, reprojectionError=5,
iterationsCount=10000, flags=cv2.SOLVEPNP_EPNP)`` followed by ``cv2.Rodrigues``.

TEST INFRASTRUCTURE ONLY (see oracle/gatsspg_oracle.py): imported by tests/ and bench.py's baseline leg, never by the
product path.

**PARITY UNPINNED.**  The algorithm lives in a third-party dependency that is absent from /root/reference and from
this image: OpenCV (``opencv-python``; OnePose's requirements pin no version, its environment used 4.5.x).  No reference
outputs can be generated here, so this file restates the PUBLISHED algorithm of that dependency and is checked only
through domain properties (exact recovery of synthetic poses, robustness to outliers, agreement of independent
formulations), not against cv2 outputs:

* EPnP -- Lepetit, Moreno-Noguer, Fua, "EPnP: An Accurate O(n) Solution to the PnP Problem", IJCV 2009, as implemented in
  OpenCV ``modules/calib3d/src/epnp.cpp``: four control points (centroid + principal directions), barycentric
  coordinates, the 2n x 12 system M, the four right null vectors of M^T M, the three beta approximations (N = 1, 2, 3) each
  refined by 5 Gauss-Newton steps, absolute orientation (Procrustes) and the choice of the N with the smallest
  reprojection error.
* RANSAC -- OpenCV ``solvePnPRansac`` / ``RANSACPointSetRegistrator``: minimal sets of 5 correspondences for EPNP, EPnP on the
  sample, inliers = squared reprojection error <= reprojectionError^2, keep the model with the most inliers, final EPnP
  over the inliers of the best model.  Differences, deliberate and documented: the sample indices come from a
  counter-based hash (OpenCV's cv::RNG stream cannot be reproduced without OpenCV), and ALL `iterations` hypotheses are
  evaluated instead of stopping early at OpenCV's adaptive confidence bound (a superset of its search).
"""
from __future__ import annotations

import numpy as np

F64 = np.float64
PAIRS = ((0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3))
MODEL_POINTS = 5          # solvePnPRansac: minimal set for SOLVEPNP_EPNP


# ---- sampling ------------------------------------------------------------------------------------------
def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


def sample_indices(seed, hyp, n):
    """MODEL_POINTS distinct indices in [0, n) for hypothesis `hyp`: successive hash draws, duplicates rejected
    (the same integer arithmetic runs on the GPU)."""
    out = []
    ctr = 0
    while len(out) < MODEL_POINTS:
        r = _splitmix64((seed << 40) ^ (hyp << 8) ^ ctr)
        ctr += 1
        idx = int((r >> 11) % n)
        if idx not in out:
            out.append(idx)
    return out


# ---- EPnP -----------------------------------------------------------------------------------------------
def _control_points(pw):
    """epnp.cpp choose_control_points: centroid + sqrt(eigenvalue / n) * principal directions."""
    n = len(pw)
    c0 = pw.mean(axis=0)
    d = pw - c0
    w, v = np.linalg.eigh(d.T @ d)            # ascending
    cws = [c0]
    for i in (2, 1, 0):                       # descending eigenvalues, like the SVD OpenCV uses
        d = v[:, i]
        # an eigenvector's sign is arbitrary, and with noisy data the EPnP solution depends (at noise level) on which
        # of the two mirrored control points is used: fix it -- largest-magnitude component positive (OpenCV's own sign
        # is whatever its SVD returns)
        if d[np.argmax(np.abs(d))] < 0:
            d = -d
        cws.append(c0 + np.sqrt(max(w[i], 0.0) / n) * d)
    return np.array(cws)


def _barycentric(pw, cws):
    cc = (cws[1:] - cws[0]).T                 # columns = control directions
    a123 = np.linalg.solve(cc, (pw - cws[0]).T).T
    return np.concatenate([1.0 - a123.sum(axis=1, keepdims=True), a123], axis=1)


def _fill_m(alphas, uv, fu, fv, uc, vc):
    n = len(uv)
    m = np.zeros((2 * n, 12), F64)
    for j in range(4):
        m[0::2, 3 * j] = alphas[:, j] * fu
        m[0::2, 3 * j + 2] = alphas[:, j] * (uc - uv[:, 0])
        m[1::2, 3 * j + 1] = alphas[:, j] * fv
        m[1::2, 3 * j + 2] = alphas[:, j] * (vc - uv[:, 1])
    return m


def _l6x10(v):
    """v: the 4 null vectors (v[0] = smallest eigenvalue), each [12]."""
    dv = np.array([[v[i][3 * a:3 * a + 3] - v[i][3 * b:3 * b + 3] for (a, b) in PAIRS] for i in range(4)])   # [4][6][3]
    dot = lambda i, j: (dv[i] * dv[j]).sum(axis=1)                                                              # [6]
    return np.stack([dot(0, 0), 2 * dot(0, 1), dot(1, 1), 2 * dot(0, 2), 2 * dot(1, 2), dot(2, 2),
                     2 * dot(0, 3), 2 * dot(1, 3), 2 * dot(2, 3), dot(3, 3)], axis=1)


def _lstsq(a, b):
    return np.linalg.lstsq(a, b, rcond=None)[0]


def _betas_approx(n_approx, l, rho):
    b = np.zeros(4, F64)
    if n_approx == 1:                                   # unknowns B11 B12 B13 B14
        x = _lstsq(l[:, [0, 1, 3, 6]], rho)
        if x[0] < 0:
            b[0] = np.sqrt(-x[0]); b[1:] = -x[1:] / b[0]
        else:
            b[0] = np.sqrt(x[0]); b[1:] = x[1:] / b[0]
    elif n_approx == 2:                                 # B11 B12 B22
        x = _lstsq(l[:, [0, 1, 2]], rho)
        if x[0] < 0:
            b[0] = np.sqrt(-x[0]); b[1] = np.sqrt(-x[2]) if x[2] < 0 else 0.0
        else:
            b[0] = np.sqrt(x[0]); b[1] = np.sqrt(x[2]) if x[2] > 0 else 0.0
        if x[1] < 0:
            b[0] = -b[0]
    else:                                               # B11 B12 B22 B13 B23
        x = _lstsq(l[:, [0, 1, 2, 3, 4]], rho)
        if x[0] < 0:
            b[0] = np.sqrt(-x[0]); b[1] = np.sqrt(-x[2]) if x[2] < 0 else 0.0
        else:
            b[0] = np.sqrt(x[0]); b[1] = np.sqrt(x[2]) if x[2] > 0 else 0.0
        if x[1] < 0:
            b[0] = -b[0]
        b[2] = x[3] / b[0]
    return b


def _gauss_newton(l, rho, b, iters=5):
    b = b.copy()
    for _ in range(iters):
        a = np.stack([2 * l[:, 0] * b[0] + l[:, 1] * b[1] + l[:, 3] * b[2] + l[:, 6] * b[3],
                      l[:, 1] * b[0] + 2 * l[:, 2] * b[1] + l[:, 4] * b[2] + l[:, 7] * b[3],
                      l[:, 3] * b[0] + l[:, 4] * b[1] + 2 * l[:, 5] * b[2] + l[:, 8] * b[3],
                      l[:, 6] * b[0] + l[:, 7] * b[1] + l[:, 8] * b[2] + 2 * l[:, 9] * b[3]], axis=1)
        r = rho - (l[:, 0] * b[0] * b[0] + l[:, 1] * b[0] * b[1] + l[:, 2] * b[1] * b[1] + l[:, 3] * b[0] * b[2]
                   + l[:, 4] * b[1] * b[2] + l[:, 5] * b[2] * b[2] + l[:, 6] * b[0] * b[3] + l[:, 7] * b[1] * b[3]
                   + l[:, 8] * b[2] * b[3] + l[:, 9] * b[3] * b[3])
        b = b + _lstsq(a, r)
    return b


def _r_and_t(v, betas, alphas, pw, uv, fu, fv, uc, vc):
    ccs = sum(betas[k] * v[k].reshape(4, 3) for k in range(4))              # control points in the camera frame
    pcs = alphas @ ccs
    if pcs[0, 2] < 0:                                                      # solve_for_sign
        ccs, pcs = -ccs, -pcs
    pc0, pw0 = pcs.mean(axis=0), pw.mean(axis=0)                           # estimate_R_and_t (absolute orientation)
    abt = (pcs - pc0).T @ (pw - pw0)
    u, _, vt = np.linalg.svd(abt)
    r = u @ vt
    if np.linalg.det(r) < 0:
        r[2] = -r[2]
    t = pc0 - r @ pw0
    p = pw @ r.T + t
    with np.errstate(divide="ignore", invalid="ignore"):
        ue = uc + fu * p[:, 0] / p[:, 2]
        ve = vc + fv * p[:, 1] / p[:, 2]
        err = np.sqrt((uv[:, 0] - ue) ** 2 + (uv[:, 1] - ve) ** 2).mean()
    return r, t, err


def epnp(pw, uv, k):
    """pw [n,3] object points, uv [n,2] pixels, k 3x3 intrinsics -> (R [3,3], t [3]); n >= 4."""
    pw = np.asarray(pw, F64); uv = np.asarray(uv, F64); k = np.asarray(k, F64)
    fu, fv, uc, vc = k[0, 0], k[1, 1], k[0, 2], k[1, 2]
    cws = _control_points(pw)
    alphas = _barycentric(pw, cws)
    m = _fill_m(alphas, uv, fu, fv, uc, vc)
    w, vecs = np.linalg.eigh(m.T @ m)                                      # ascending: columns 0..3 span the null space
    v = [vecs[:, i] for i in range(4)]
    l = _l6x10(v)
    rho = np.array([((cws[a] - cws[b]) ** 2).sum() for (a, b) in PAIRS])
    best = None
    for n_approx in (1, 2, 3):
        b = _gauss_newton(l, rho, _betas_approx(n_approx, l, rho))
        r, t, err = _r_and_t(v, b, alphas, pw, uv, fu, fv, uc, vc)
        if np.isfinite(err) and (best is None or err < best[2]):
            best = (r, t, err)
    if best is None:
        return np.full((3, 3), np.nan), np.full(3, np.nan)
    return best[0], best[1]


# ---- RANSAC ---------------------------------------------------------------------------------------------
def reproj_err2(r, t, pw, uv, k):
    p = pw @ r.T + t
    with np.errstate(divide="ignore", invalid="ignore"):
        ue = k[0, 2] + k[0, 0] * p[:, 0] / p[:, 2]
        ve = k[1, 2] + k[1, 1] * p[:, 1] / p[:, 2]
        e = (uv[:, 0] - ue) ** 2 + (uv[:, 1] - ve) ** 2
    return np.where(np.isfinite(e), e, np.inf)


def solve_pnp_ransac(pts_3d, pts_2d, k, reproj_error=5.0, iterations=10000, seed=0, return_debug=False):
    """-> (ok, R, t, inlier indices).  Mirrors cv2.solvePnPRansac(..., flags=SOLVEPNP_EPNP) with the two documented
    differences (hash sampler, no early termination).  Ties in the inlier count keep the lowest hypothesis index."""
    pw = np.asarray(pts_3d, F64); uv = np.asarray(pts_2d, F64); k = np.asarray(k, F64)
    n = len(pw)
    if n < MODEL_POINTS:                      # OpenCV: npoints < model_points -> false (npoints == model_points: one EPnP)
        return (False, np.eye(3), np.zeros(3), np.zeros(0, np.int64)) + ((None,) if return_debug else ())
    thr2 = reproj_error * reproj_error
    best_cnt, best_h, best_mask = -1, -1, None
    counts = np.zeros(iterations, np.int64)
    for h in range(iterations):
        idx = sample_indices(seed, h, n)
        r, t = epnp(pw[idx], uv[idx], k)
        if not np.isfinite(r).all():
            continue
        mask = reproj_err2(r, t, pw, uv, k) <= thr2
        counts[h] = mask.sum()
        if counts[h] > best_cnt:
            best_cnt, best_h, best_mask = int(counts[h]), h, mask
    if best_cnt < MODEL_POINTS:
        return (False, np.eye(3), np.zeros(3), np.zeros(0, np.int64)) + ((None,) if return_debug else ())
    inl = np.nonzero(best_mask)[0]
    r, t = epnp(pw[inl], uv[inl], k)          # final solvePnP(EPNP) over the inliers of the best model
    out = (True, r, t, inl)
    return out + (dict(best_hypothesis=best_h, counts=counts),) if return_debug else out


def ransac_pnp(k, pts_2d, pts_3d, scale=1, iterations=10000, seed=0):
    """eval_utils.ransac_PnP (:18-42): -> (pose [3,4], pose_homo [4,4], inliers [m,1])."""
    pts_3d = np.ascontiguousarray(np.asarray(pts_3d, F64)) * scale
    ok, r, t, inl = solve_pnp_ransac(pts_3d, np.asarray(pts_2d, F64), np.asarray(k, F64), 5.0, iterations, seed)
    if not ok:
        return np.eye(4)[:3], np.eye(4), []
    pose = np.concatenate([r, (t / scale)[:, None]], axis=-1)
    return pose, np.concatenate([pose, np.array([[0, 0, 0, 1.0]])], axis=0), inl[:, None]


def query_pose_error(pose_pred, pose_gt):
    """eval_utils.query_pose_error (:45-63): (angular error in degrees, translation error in cm)."""
    pose_pred, pose_gt = np.asarray(pose_pred)[:3], np.asarray(pose_gt)[:3]
    t_err = np.linalg.norm(pose_pred[:, 3] - pose_gt[:, 3]) * 100
    trace = min(np.trace(pose_pred[:, :3] @ pose_gt[:, :3].T), 3.0)
    return np.rad2deg(np.arccos((trace - 1.0) / 2.0)), t_err
