"""RANSAC-EPnP pose from 2D-3D correspondences on the GPU: host-side mirror of the reference's
``ransac_PnP`` (src/utils/eval_utils.py:18-42) and ``query_pose_error`` (:45-63).

``ransac_PnP(K, pts_2d, pts_3d, scale)`` keeps the reference signature and return convention
(``pose [3,4]``, ``pose_homo [4,4]``, ``inliers [m,1]`` as numpy; identity + ``[]`` when the solve fails, :40-42).
``ransac_pnp_device`` leaves everything in HBM (inputs: the ``mkpts2d`` / ``mkpts3d`` tensors of ``FrameMatcher``).
The solver is the HIP library behind include/pnp.h; there is no cv2 / CPU fallback.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _native_pnp
from ._native import NativeError  # noqa: F401

REPROJ_ERROR = 5.0        # eval_utils.py:30
ITERATIONS = 10000        # eval_utils.py:31


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _k_array(K):
    k = np.ascontiguousarray(np.asarray(K.detach().cpu() if isinstance(K, torch.Tensor) else K, dtype=np.float64)).reshape(9)
    return (ctypes.c_double * 9)(*k.tolist())


@torch.no_grad()
def ransac_pnp_device(K, pts_2d, pts_3d, scale=1.0, reproj_error=REPROJ_ERROR, iterations=ITERATIONS, seed=0):
    """pts_2d [n,2], pts_3d [n,3] GPU tensors -> (pose [3,4] float64, inlier_mask [n] int32, info [4] int32:
    ok, inliers, best hypothesis, its count), all on the GPU, nothing synchronised."""
    if not (pts_2d.is_cuda and pts_3d.is_cuda):
        raise RuntimeError("onepose_amd.pnp runs only on a ROCm GPU (there is no CPU fallback)")
    dev = pts_2d.device
    p2 = pts_2d.to(torch.float32).contiguous()
    p3 = pts_3d.to(torch.float32).contiguous()
    n = p2.shape[0]
    lib = _native_pnp.load()
    nbytes = lib.pnp_workspace_bytes(n, iterations)
    ws = torch.empty(max(nbytes, 256), device=dev, dtype=torch.uint8)
    pose = torch.empty(3, 4, device=dev, dtype=torch.float64)
    mask = torch.zeros(max(n, 1), device=dev, dtype=torch.int32)
    info = torch.zeros(4, device=dev, dtype=torch.int32)
    with torch.cuda.device(dev):   # the C ABI launches on the current device
        _native_pnp.check(lib.pnp_ransac_epnp(p3.data_ptr(), p2.data_ptr(), _k_array(K), float(scale), n, float(reproj_error),
                                              int(iterations), int(seed), pose.data_ptr(), mask.data_ptr(), info.data_ptr(),
                                              ws.data_ptr(), ws.numel(), _stream(dev)), "pnp_ransac_epnp")
    return pose, mask[:n], info


@torch.no_grad()
def ransac_pnp_from_matches(K, kpts2d, kpts3d, matches0, scale=1.0, reproj_error=REPROJ_ERROR, iterations=ITERATIONS, seed=0):
    """inference.py:148-155 without a host round trip: kpts2d [n1,2] (extractor), kpts3d [N3,3] (database), matches0 [n1]
    int64 (-1 = unmatched), all on the GPU -> (pose [3,4] float64, inlier_mask [n1] int32 per query keypoint, info [4])."""
    if not (kpts2d.is_cuda and kpts3d.is_cuda and matches0.is_cuda):
        raise RuntimeError("onepose_amd.pnp runs only on a ROCm GPU (there is no CPU fallback)")
    dev = kpts2d.device
    k2 = kpts2d.to(torch.float32).contiguous()
    k3 = kpts3d.to(torch.float32).contiguous()
    m0 = matches0.to(torch.int64).contiguous()
    n1 = k2.shape[0]
    lib = _native_pnp.load()
    ws = torch.empty(lib.pnp_workspace_bytes(n1, iterations), device=dev, dtype=torch.uint8)
    pose = torch.empty(3, 4, device=dev, dtype=torch.float64)
    mask = torch.empty(n1, device=dev, dtype=torch.int32)
    info = torch.empty(4, device=dev, dtype=torch.int32)
    with torch.cuda.device(dev):
        _native_pnp.check(lib.pnp_ransac_epnp_matches(k2.data_ptr(), k3.data_ptr(), m0.data_ptr(), n1, _k_array(K), float(scale),
                                                      float(reproj_error), int(iterations), int(seed), pose.data_ptr(), mask.data_ptr(),
                                                      info.data_ptr(), ws.data_ptr(), ws.numel(), _stream(dev)), "pnp_ransac_epnp_matches")
    return pose, mask, info


def ransac_PnP(K, pts_2d, pts_3d, scale=1, iterations=ITERATIONS, seed=0):
    """ solve pnp -- drop-in for eval_utils.ransac_PnP (:18-42); numpy or tensor inputs, numpy outputs."""
    dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
    if dev is None:
        raise RuntimeError("onepose_amd.pnp runs only on a ROCm GPU (there is no CPU fallback)")
    to = lambda a: a if isinstance(a, torch.Tensor) and a.is_cuda else torch.as_tensor(np.asarray(a.cpu() if isinstance(a, torch.Tensor) else a, dtype=np.float32)).to(dev)  # noqa: E731
    p2, p3 = to(pts_2d), to(pts_3d)
    if p2.shape[0] < 5:                       # cv2 raises / returns false for fewer than the 5 model points: :40-42
        return np.eye(4)[:3], np.eye(4), []
    pose, mask, info = ransac_pnp_device(K, p2, p3, scale, REPROJ_ERROR, iterations, seed)
    info = info.cpu().numpy()
    if not info[0]:
        return np.eye(4)[:3], np.eye(4), []
    pose = pose.cpu().numpy()
    inliers = np.nonzero(mask.cpu().numpy())[0].astype(np.int32)[:, None]      # cv2 returns an [m,1] int32 index array
    return pose, np.concatenate([pose, np.array([[0, 0, 0, 1.0]])], axis=0), inliers


@torch.no_grad()
def epnp(K, pts_2d, pts_3d, scale=1.0):
    """EPnP over all correspondences (cv2.solvePnP(flags=SOLVEPNP_EPNP)) -> pose [3,4] float64 on the GPU."""
    p2 = pts_2d.to(torch.float32).contiguous()
    p3 = pts_3d.to(torch.float32).contiguous()
    pose = torch.empty(3, 4, device=p2.device, dtype=torch.float64)
    lib = _native_pnp.load()
    with torch.cuda.device(p2.device):
        _native_pnp.check(lib.pnp_epnp(p3.data_ptr(), p2.data_ptr(), _k_array(K), float(scale), p2.shape[0], pose.data_ptr(), None, 0,
                                       _stream(p2.device)), "pnp_epnp")
    return pose


def query_pose_error(pose_pred, pose_gt):
    """eval_utils.query_pose_error (:45-63): (angular error [deg], translation error [cm]) -- host-side, six flops."""
    pose_pred, pose_gt = np.asarray(pose_pred)[:3], np.asarray(pose_gt)[:3]
    translation_distance = np.linalg.norm(pose_pred[:, 3] - pose_gt[:, 3]) * 100
    trace = np.trace(np.dot(pose_pred[:, :3], pose_gt[:, :3].T))
    trace = trace if trace <= 3 else 3
    return np.rad2deg(np.arccos((trace - 1.0) / 2.0)), translation_distance


class Evaluator:
    """cm-degree pose accuracy over a sequence: host-side mirror of src/evaluators/cmd_evaluator.py (the bookkeeping
    inference.py:102,163,166 does around the pose solver).  A frame counts for the k cm / k degree metric when its
    translation error is below k cm AND its rotation error below k degrees (k = 1, 3, 5)."""

    THRESHOLDS = (1, 3, 5)

    def __init__(self):
        self.cmd1, self.cmd3, self.cmd5, self.cmd7, self.add = [], [], [], [], []

    def _hits(self, k):
        return {1: self.cmd1, 3: self.cmd3, 5: self.cmd5}[k]

    def evaluate(self, pose_pred, pose_gt):
        if pose_pred is None:                      # cmd_evaluator.py:36-40 (also feeds the unused 7 cm list)
            for k in self.THRESHOLDS:
                self._hits(k).append(False)
            self.cmd7.append(False)
            return
        ang, trans = query_pose_error(pose_pred, pose_gt)
        for k in self.THRESHOLDS:
            self._hits(k).append(bool(trans < k and ang < k))

    def summarize(self):
        out = {f"cmd{k}": np.mean(self._hits(k)) for k in self.THRESHOLDS}
        for k in self.THRESHOLDS:
            print(f"{k} cm {k} degree metric: {out[f'cmd{k}']}")
        self.cmd1, self.cmd3, self.cmd5, self.cmd7 = [], [], [], []
        return out
