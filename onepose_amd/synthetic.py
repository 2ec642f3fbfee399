"""Seeded synthetic weights and inputs for the GATsSPG matcher.

There is no checkpoint and no dataset in the build environment, so benchmarks and parity
tests run on random-init weights of the reference architecture and synthetic unit-norm
descriptors (real SuperPoint descriptors are unit-norm).  Everything here is
generated with ``numpy.random.RandomState`` so the very same tensors can be rebuilt bit for
bit on any machine (golden-vector generation in the build container, parity tests on the GPU
box) without shipping 22 MB of weights.

Parameter names / shapes follow the reference ``state_dict`` exactly
(src/models/GATsSPG_architectures/GATs_SuperGlue.py:145-177, GATs.py:25-28).
"""
from __future__ import annotations

import numpy as np

D = 256
GATS_LAYERS = (0, 3, 6, 9)
ATTN_LAYERS = (1, 2, 4, 5, 7, 8, 10, 11)


def _conv(rs, out_c, in_c):
    """PyTorch's default Conv1d init: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and bias."""
    bound = 1.0 / np.sqrt(in_c)
    w = rs.uniform(-bound, bound, size=(out_c, in_c, 1)).astype(np.float32)
    b = rs.uniform(-bound, bound, size=(out_c,)).astype(np.float32)
    return w, b


def make_state_dict(seed=0, keypoints_encoder=(32, 64, 128)):
    """Random-init weights with the distributions of the reference constructors
    (xavier_normal(gain=1.414) for GATs W/a, GATs.py:25-28; Conv1d default elsewhere; the last
    bias of every MLP zero, GATs_SuperGlue.py:109,136)."""
    rs = np.random.RandomState(seed)
    sd = {}
    for name, inp in (("kenc_2d", 3), ("kenc_3d", 4)):
        chans = [inp] + list(keypoints_encoder) + [D]
        for i in range(1, len(chans)):
            w, b = _conv(rs, chans[i], chans[i - 1])
            if i == len(chans) - 1:
                b[:] = 0
            sd[f"{name}.encoder.{3 * (i - 1)}.weight"] = w
            sd[f"{name}.encoder.{3 * (i - 1)}.bias"] = b
    for i in range(12):
        p = f"gnn.layers.{i}"
        if i in GATS_LAYERS:
            sd[f"{p}.W"] = (rs.standard_normal((D, D)) * 1.414 * np.sqrt(2.0 / (D + D))).astype(np.float32)
            sd[f"{p}.a"] = (rs.standard_normal((2 * D, 1)) * 1.414 * np.sqrt(2.0 / (2 * D + 1))).astype(np.float32)
        else:
            w, b = _conv(rs, D, D)
            sd[f"{p}.attn.merge.weight"], sd[f"{p}.attn.merge.bias"] = w, b
            for j in range(3):
                w, b = _conv(rs, D, D)
                sd[f"{p}.attn.proj.{j}.weight"], sd[f"{p}.attn.proj.{j}.bias"] = w, b
            w, b = _conv(rs, 2 * D, 2 * D)
            sd[f"{p}.mlp.0.weight"], sd[f"{p}.mlp.0.bias"] = w, b
            w, b = _conv(rs, D, 2 * D)
            b[:] = 0
            sd[f"{p}.mlp.3.weight"], sd[f"{p}.mlp.3.bias"] = w, b
    w, b = _conv(rs, D, D)
    sd["final_proj.weight"], sd["final_proj.bias"] = w, b
    sd["bin_score"] = np.array(1.0, dtype=np.float32)
    return sd


def make_passthrough_state_dict(seed=0):
    """Fixture-B weights: the last MLP conv of every AttentionPropagation layer
    is zeroed (deltas vanish) and final_proj is the identity, so planted descriptor matches
    survive the network and the dual-softmax produces confidences near 1."""
    sd = make_state_dict(seed)
    for i in ATTN_LAYERS:
        sd[f"gnn.layers.{i}.mlp.3.weight"][:] = 0
        sd[f"gnn.layers.{i}.mlp.3.bias"][:] = 0
    sd["final_proj.weight"] = np.eye(D, dtype=np.float32)[:, :, None].copy()
    sd["final_proj.bias"][:] = 0
    return sd


TRAINED_BASE_SEED = 21


def make_trained_state_dict(path=None):
    """TRAINED weights (round 5; tests/golden/make_trained_golden.py): the random-init state dict of seed 21 plus a
    rank-8 update of every weight matrix on the forward path and a full update of every bias, obtained by training the
    REFERENCE module with the reference's focal loss on planted synthetic frames (every AttentionPropagation delta and
    final_proj active -- nothing zeroed, unlike the pass-through fixture).  Only the factors are committed (~1 MB):
    `W = W_init + sum_k U[:, k] V[:, k]^T`, accumulated rank by rank in float64 with elementwise numpy ops (no BLAS: the
    summation order is fixed, so the container that ran the reference and the GPU box rebuild the same fp32 bits)."""
    import os
    if path is None:
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "trained_lowrank.npz")
    fac = np.load(path)
    sd = make_state_dict(TRAINED_BASE_SEED)
    for key in fac.files:
        kind, name = key.split("::", 1)
        if kind == "B":
            sd[name] = (sd[name].astype(np.float64) + fac[key].astype(np.float64)).astype(np.float32)
        elif kind == "U":
            u, v = fac[key].astype(np.float64), fac["V::" + name].astype(np.float64)
            w = sd[name]
            acc = w.reshape(w.shape[0], -1).astype(np.float64)
            for k in range(u.shape[1]):
                acc = acc + u[:, k:k + 1] * v[None, :, k]
            sd[name] = acc.astype(np.float32).reshape(w.shape)
    return sd


def _unit(x, axis):
    return (x / np.linalg.norm(x, axis=axis, keepdims=True)).astype(np.float32)


def make_inputs(b, n1, n2, num_leaf=8, seed=1, planted=False, noise=(0.2, 0.3), with_targets=False):
    """Synthetic forward() inputs (keys of GATs_SuperGlue.py:181-189).

    planted=False: independent unit-norm random descriptors (benchmark distribution).
    planted=True (fixture B): leaves are noisy copies (noise norm ~0.2) of their 3D descriptor and
    the first n1//2 query descriptors are noisy copies (noise norm ~0.3) of distinct random 3D
    descriptors, so well-separated ground-truth matches exist.  `noise` = (leaf, query) noise norms.
    with_targets=True additionally returns the planted ground truth `[b, k]` (query i of sample bi matches
    3D point targets[bi, i]); the dict itself never changes.
    """
    rs = np.random.RandomState(seed)
    targets = []
    d3 = _unit(rs.standard_normal((b, D, n2)), 1)
    if planted:
        leaves = np.repeat(d3, num_leaf, axis=2) + noise[0] * rs.standard_normal((b, D, n2 * num_leaf)) / np.sqrt(D)
        d2db = _unit(leaves, 1)
        dq = _unit(rs.standard_normal((b, D, n1)), 1)
        k = min(n1 // 2, n2)
        for bi in range(b):
            tgt = rs.permutation(n2)[:k]
            dq[bi, :, :k] = _unit(d3[bi][:, tgt] + noise[1] * rs.standard_normal((D, k)) / np.sqrt(D), 0)
            targets.append(tgt)
    else:
        d2db = _unit(rs.standard_normal((b, D, n2 * num_leaf)), 1)
        dq = _unit(rs.standard_normal((b, D, n1)), 1)
    data = {
        "keypoints2d": (rs.rand(b, n1, 2) * 512).astype(np.float32),
        "keypoints3d": (rs.rand(b, n2, 3) - 0.5).astype(np.float32),
        "descriptors2d_query": dq,
        "descriptors3d_db": d3,
        "descriptors2d_db": d2db,
    }
    if with_targets:
        return data, (np.stack(targets) if targets else np.zeros((b, 0), np.int64))
    return data


# ---- SuperPoint extractor (src/models/extractors/SuperPoint/superpoint.py:115-133) ---------------
# (name, out channels, in channels, kernel size) in forward order; the state_dict holds
# "<name>.weight" [out, in, k, k] and "<name>.bias" [out].
SPP_LAYERS = (
    ("conv1a", 64, 1, 3), ("conv1b", 64, 64, 3), ("conv2a", 64, 64, 3), ("conv2b", 64, 64, 3),
    ("conv3a", 128, 64, 3), ("conv3b", 128, 128, 3), ("conv4a", 128, 128, 3), ("conv4b", 128, 128, 3),
    ("convPa", 256, 128, 3), ("convPb", 65, 256, 1), ("convDa", 256, 128, 3), ("convDb", 256, 256, 1),
)


def make_spp_state_dict(seed=0, descriptor_dim=256, logit_gain=6.0):
    """Random weights of the SuperPoint architecture.  He-normal (variance-preserving through the
    ReLU stack) instead of PyTorch's default init, and a larger gain on the detector's last
    conv, so that the 65-way cell softmax has a wide spread: keypoint scores then straddle the
    0.005 threshold and NMS decisions are separated by far more than fp32 rounding."""
    rs = np.random.RandomState(seed)
    sd = {}
    for name, oc, ic, k in SPP_LAYERS:
        if name == "convDb":
            oc = descriptor_dim
        std = np.sqrt(2.0 / (ic * k * k))
        if name == "convPb":
            std *= logit_gain
        sd[f"{name}.weight"] = (rs.standard_normal((oc, ic, k, k)) * std).astype(np.float32)
        sd[f"{name}.bias"] = (rs.standard_normal((oc,)) * 0.05).astype(np.float32)
    return sd


def make_image(b=1, h=512, w=512, seed=0):
    """Synthetic grayscale image batch [b, 1, h, w] in [0, 1]: overlapping rectangles and discs on a
    smooth gradient plus a little noise (corners and edges for the detector to respond to)."""
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    out = np.empty((b, 1, h, w), np.float32)
    for i in range(b):
        img = 0.3 + 0.2 * np.sin(xx / w * rs.uniform(2, 6)) * np.cos(yy / h * rs.uniform(2, 6))
        for _ in range(max(4, h * w // 4096)):
            cy, cx = rs.uniform(0, h), rs.uniform(0, w)
            ry, rx = rs.uniform(3, h / 6), rs.uniform(3, w / 6)
            val = rs.uniform(-0.4, 0.4)
            if rs.rand() < 0.5:
                m = (np.abs(yy - cy) < ry) & (np.abs(xx - cx) < rx)
            else:
                m = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1
            img = img + val * m
        img = img + rs.standard_normal((h, w)) * 0.02
        out[i, 0] = np.clip(img, 0, 1)
    return out


# ---- PnP (src/utils/eval_utils.py:18-42) ------------------------------------------------------------------
def make_pnp_problem(n=300, outlier_frac=0.3, noise_px=0.5, seed=0):
    """A synthetic 2D-3D correspondence set like the matcher's output: object points in a ~20 cm box (metres, as in
    the OnePose annotations), a random pose 0.4-0.8 m in front of a 512x512 crop camera, Gaussian pixel noise and a
    fraction of gross outliers.  -> dict(K [3,3], pts_3d [n,3], pts_2d [n,2], pose_gt [3,4], inlier_mask [n])."""
    rs = np.random.RandomState(seed)
    k = np.array([[600.0 + rs.uniform(-50, 50), 0, 256.0 + rs.uniform(-20, 20)],
                  [0, 600.0 + rs.uniform(-50, 50), 256.0 + rs.uniform(-20, 20)], [0, 0, 1.0]])
    pts = rs.uniform(-0.1, 0.1, size=(n, 3))
    a = rs.standard_normal((3, 3))
    q, r = np.linalg.qr(a)
    q = q * np.sign(np.diag(r))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    t = np.array([rs.uniform(-0.05, 0.05), rs.uniform(-0.05, 0.05), rs.uniform(0.4, 0.8)])
    pc = pts @ q.T + t
    uv = np.stack([k[0, 2] + k[0, 0] * pc[:, 0] / pc[:, 2], k[1, 2] + k[1, 1] * pc[:, 1] / pc[:, 2]], axis=1)
    uv = uv + rs.standard_normal(uv.shape) * noise_px
    out = rs.rand(n) < outlier_frac
    uv[out] = rs.uniform(0, 512, size=(int(out.sum()), 2))
    return {"K": k, "pts_3d": pts.astype(np.float32), "pts_2d": uv.astype(np.float32),
            "pose_gt": np.concatenate([q, t[:, None]], axis=1), "inlier_mask": ~out}


def make_pose_pairs(seed=0, n=40):
    """(pose_pred, pose_gt) pairs whose rotation / translation errors straddle the 1, 3 and 5 cm-degree thresholds of the
    reference evaluator; every 7th prediction is given as a 4x4 matrix (cmd_evaluator.py:42-45)."""
    rs = np.random.RandomState(seed)
    out = []
    for i in range(n):
        q, _ = np.linalg.qr(rs.standard_normal((3, 3)))
        if np.linalg.det(q) < 0:
            q[:, 0] = -q[:, 0]
        t = rs.uniform(-0.3, 0.3, 3) + np.array([0, 0, 0.6])
        ang = rs.choice([0.2, 0.8, 2.0, 4.0, 8.0]) * np.pi / 180
        ax = rs.standard_normal(3)
        ax /= np.linalg.norm(ax)
        kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        dr = np.eye(3) + np.sin(ang) * kx + (1 - np.cos(ang)) * kx @ kx
        dt = rs.choice([0.002, 0.008, 0.02, 0.04, 0.08]) * ax
        gt = np.concatenate([q, t[:, None]], 1)
        pred = np.concatenate([dr @ q, (t + dt)[:, None]], 1)
        if i % 7 == 0:
            pred = np.concatenate([pred, [[0, 0, 0, 1.0]]], 0)
        out.append((pred, gt))
    return out


# ---- object annotation files (src/sfm/postprocess/feature_process.py:191-194,357-363) --------------
def make_annotation(n=50, dim=256, seed=0, max_views=15):
    """Synthetic content of anno_3d_average.npz / anno_3d_collect.npz / idxs.npy: every 3D point has 1..max_views-1
    collected per-view descriptors (some fewer than num_leaf = 8, some more), concatenated point after point."""
    rs = np.random.RandomState(seed)
    idxs = rs.randint(1, max_views, size=n)
    k = int(idxs.sum())
    collect = rs.standard_normal((dim, k)).astype(np.float32)
    owner = np.repeat(np.arange(n), idxs)
    avg = np.stack([collect[:, owner == i].mean(axis=1) for i in range(n)], axis=1).astype(np.float32)
    return {"keypoints3d": (rs.rand(n, 3) - 0.5).astype(np.float32), "idxs": idxs, "owner": owner,
            "collect_descriptors": collect, "collect_scores": rs.rand(k, 1).astype(np.float32),
            "avg_descriptors": avg, "avg_scores": rs.rand(n, 1).astype(np.float32)}


def write_annotation(dirname, anno):
    """The three files inference.py:113-115 loads, in the reference's layout."""
    import os
    np.savez(os.path.join(dirname, "anno_3d_average.npz"), keypoints3d=anno["keypoints3d"], descriptors3d=anno["avg_descriptors"],
             scores3d=anno["avg_scores"])
    np.savez(os.path.join(dirname, "anno_3d_collect.npz"), keypoints3d=anno["keypoints3d"],
             descriptors3d=anno["collect_descriptors"], scores3d=anno["collect_scores"])
    np.save(os.path.join(dirname, "idxs.npy"), anno["idxs"])
