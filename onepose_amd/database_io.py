"""On-disk object annotation -> resident 3D database tensors for the matcher.

Formats written by the reference's SfM post-processing (src/sfm/postprocess/feature_process.py:191-194,
357-363) and consumed at inference.py:113-130:

    anno_3d_average.npz : keypoints3d [N,3], descriptors3d [256,N] (per-point mean descriptor), scores3d [N,1]
    anno_3d_collect.npz : keypoints3d [N,3], descriptors3d [256,K] (all collected per-view descriptors,
                          concatenated point after point), scores3d [K,1]
    idxs.npy            : [N] number of collected descriptors of each point (sum = K)

``pad_features3d_random`` and ``build_features3d_leaves`` mirror src/utils/data_utils.py:143-205 (same names, argument
meaning and results): every 3D point gets exactly ``num_leaf`` leaves -- a random subset of its collected descriptors when
it has at least ``num_leaf``, otherwise all of them plus all-ones "dustbin" columns, in random order.  The reference
draws one ``np.random.permutation`` per point from numpy's GLOBAL generator; ``rng=None`` does exactly that (so
``np.random.seed(s)`` before the call reproduces the reference's leaf choice bit for bit -- pinned by
tests/golden/make_db_golden.py), ``rng=int`` uses a private ``RandomState`` with the same stream.
``build_leaves`` is the vectorised variant with an explicit ``numpy.random.Generator`` (same semantics, different stream).
"""
from __future__ import annotations

import numpy as np
import torch


def _legacy_rng(rng):
    if rng is None:
        return np.random           # the global generator, like the reference
    if isinstance(rng, (int, np.integer)):
        return np.random.RandomState(int(rng))
    return rng


def pad_features3d_random(descriptors, scores, n_target_shape):
    """Pad (all-ones descriptors, zero scores) or truncate to n_target_shape points (data_utils.py:143-160)."""
    descriptors = np.asarray(descriptors, dtype=np.float32)
    scores = np.asarray(scores, dtype=np.float32)
    dim, n_pad = descriptors.shape[0], n_target_shape - descriptors.shape[1]
    if n_pad < 0:
        return descriptors[:, :n_target_shape], scores[:n_target_shape, :]
    return (np.concatenate([descriptors, np.ones((dim, n_pad), np.float32)], axis=1),
            np.concatenate([scores, np.zeros((n_pad, 1), np.float32)], axis=0))


def build_features3d_leaves(descriptors, scores, idxs, n_target_shape, num_leaf, rng=None):
    """data_utils.py:163-205 with the reference's own random stream: one ``permutation`` per point, in point order, of
    either its collected columns (truncated to num_leaf) or its columns + dustbin ids.  Returns
    (descriptors [dim, n_target_shape * num_leaf], scores [n_target_shape * num_leaf, 1]) as float32 numpy arrays."""
    rs = _legacy_rng(rng)
    descriptors = np.asarray(descriptors, dtype=np.float32)
    scores = np.asarray(scores, dtype=np.float32)
    idxs = np.asarray(idxs)
    dim, orig_num = descriptors.shape[0], idxs.shape[0]
    n_pad = n_target_shape - orig_num
    table = np.concatenate([descriptors, np.ones((dim, 1), np.float32)], axis=1)
    stable = np.concatenate([scores, np.zeros((1, 1), np.float32)], axis=0)
    dustbin = table.shape[1] - 1
    upper = np.cumsum(idxs, axis=0)
    lower = np.insert(upper[:-1], 0, 0)
    picks = []
    for start, end in zip(lower, upper):
        if num_leaf > end - start:
            ids = np.arange(start, end).tolist() + [dustbin] * int(num_leaf - (end - start))
            picks.append(rs.permutation(np.array(ids)))
        else:
            picks.append(rs.permutation(np.arange(start, end))[:num_leaf])
    sel = np.concatenate(picks, axis=0) if picks else np.zeros((0,), np.int64)
    assert sel.shape[0] == orig_num * num_leaf
    d, s = table[:, sel], stable[sel, :]
    if n_pad < 0:
        return d[:, :num_leaf * n_target_shape], s[:num_leaf * n_target_shape, :]
    return (np.concatenate([d, np.ones((dim, n_pad * num_leaf), np.float32)], axis=1),
            np.concatenate([s, np.zeros((n_pad * num_leaf, 1), np.float32)], axis=0))


def build_leaves(collect_desc, idxs, num_leaf, rng):
    """Vectorised leaf selection, explicit ``numpy.random.Generator``: collect_desc [dim,K], idxs [N] -> leaves
    [dim, N*num_leaf] (dustbin = all ones).  Same semantics as build_features3d_leaves, NOT the same random stream."""
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


def load_object_database(avg_anno_path, collect_anno_path, idxs_path, num_leaf=8, seed=None, device="cuda", reference_rng=True):
    """inference.py:113-130 -> dict(keypoints3d [1,N,3], descriptors3d_db [1,256,N], descriptors2d_db [1,256,N*num_leaf])
    on `device`, the three database-side entries of the matcher's input (GATs_SuperGlue.py:181-189).
    reference_rng=True (default): the reference's leaf choice -- ``seed=None`` draws from numpy's global generator exactly
    like inference.py does, an int seeds a private stream.  reference_rng=False: the vectorised ``build_leaves``."""
    avg, clt, idxs = np.load(avg_anno_path), np.load(collect_anno_path), np.load(idxs_path)
    kp3d = np.asarray(clt["keypoints3d"], dtype=np.float32)
    n = kp3d.shape[0]
    d3, _ = pad_features3d_random(avg["descriptors3d"], avg["scores3d"], n)
    if int(np.asarray(idxs).sum()) != clt["descriptors3d"].shape[1]:
        raise ValueError(f"idxs sums to {int(np.asarray(idxs).sum())} but anno_3d_collect holds {clt['descriptors3d'].shape[1]} descriptors")
    if reference_rng:
        leaves, _ = build_features3d_leaves(clt["descriptors3d"], clt["scores3d"], idxs, n, num_leaf, rng=seed)
    else:
        leaves = build_leaves(clt["descriptors3d"], idxs, num_leaf, np.random.default_rng(0 if seed is None else seed))
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a))[None].to(device)  # noqa: E731
    return {"keypoints3d": to(kp3d), "descriptors3d_db": to(d3), "descriptors2d_db": to(leaves)}
