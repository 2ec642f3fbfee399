"""On-disk object annotation -> resident 3D database tensors for the matcher.

Formats written by the reference's SfM post-processing (src/sfm/postprocess/feature_process.py:191-194,
357-363) and consumed at inference.py:113-130:

    anno_3d_average.npz : keypoints3d [N,3], descriptors3d [256,N] (per-point mean descriptor), scores3d [N,1]
    anno_3d_collect.npz : keypoints3d [N,3], descriptors3d [256,K] (all collected per-view descriptors,
                          concatenated point after point), scores3d [K,1]
    idxs.npy            : [N] number of collected descriptors of each point (sum = K)

Leaf selection follows the semantics of data_utils.build_features3d_leaves (src/utils/data_utils.py:163-205):
every 3D point gets exactly ``num_leaf`` leaves -- a random subset of its collected descriptors when it has at
least ``num_leaf``, otherwise all of them plus all-ones "dustbin" columns, in random order.  The reference draws
from numpy's global RNG; here the generator is explicit (``seed``), so a database can be rebuilt reproducibly.
"""
from __future__ import annotations

import numpy as np
import torch


def build_leaves(collect_desc, idxs, num_leaf, rng):
    """collect_desc [256,K], idxs [N] -> leaves [256, N*num_leaf] (dustbin = all ones)."""
    collect_desc = np.asarray(collect_desc, dtype=np.float32)
    idxs = np.asarray(idxs, dtype=np.int64)
    n, k = idxs.shape[0], collect_desc.shape[1]
    if int(idxs.sum()) != k:
        raise ValueError(f"idxs sums to {int(idxs.sum())} but anno_3d_collect holds {k} descriptors")
    starts = np.concatenate([[0], np.cumsum(idxs)[:-1]])
    # one random key per candidate slot; slots beyond a point's count are dustbin candidates.  Real descriptors are
    # preferred (keys < 1) over dustbin ones (keys >= 1) by the first argsort, then the chosen num_leaf are shuffled.
    width = max(int(idxs.max()) if n else 0, num_leaf)
    slot = np.arange(width)[None, :]
    real = slot < idxs[:, None]
    keys = rng.random((n, width)) + (~real)
    chosen = np.argsort(keys, axis=1)[:, :num_leaf]                     # random subset of real slots first
    shuffle = np.argsort(rng.random((n, num_leaf)), axis=1)             # random order, dustbin mixed in
    chosen = np.take_along_axis(chosen, shuffle, axis=1)
    is_real = np.take_along_axis(real, chosen, axis=1)
    src = np.where(is_real, starts[:, None] + chosen, k)                # k = dustbin column
    table = np.concatenate([collect_desc, np.ones((collect_desc.shape[0], 1), np.float32)], axis=1)
    return table[:, src.reshape(-1)]


def load_object_database(avg_anno_path, collect_anno_path, idxs_path, num_leaf=8, seed=0, device="cuda"):
    """-> dict(keypoints3d [1,N,3], descriptors3d_db [1,256,N], descriptors2d_db [1,256,N*num_leaf]) on `device`,
    the three database-side entries of the matcher's input (GATs_SuperGlue.py:181-189)."""
    avg, clt, idxs = np.load(avg_anno_path), np.load(collect_anno_path), np.load(idxs_path)
    kp3d = np.asarray(clt["keypoints3d"], dtype=np.float32)
    n = kp3d.shape[0]
    d3 = np.asarray(avg["descriptors3d"], dtype=np.float32)
    if d3.shape[1] < n:    # pad_features3d_random (data_utils.py:143-160): pad with ones / truncate to num_3d
        d3 = np.concatenate([d3, np.ones((d3.shape[0], n - d3.shape[1]), np.float32)], axis=1)
    d3 = d3[:, :n]
    leaves = build_leaves(clt["descriptors3d"], idxs, num_leaf, np.random.default_rng(seed))
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a))[None].to(device)  # noqa: E731
    return {"keypoints3d": to(kp3d), "descriptors3d_db": to(d3), "descriptors2d_db": to(leaves)}
