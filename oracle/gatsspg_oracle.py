"""CPU oracle for the GATsSPG 2D-3D matcher forward pass.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.  The product path
(``onepose_amd.GATsSuperGlue``) never routes through this file and fails loudly
when its HIP extension is missing.

It is a numpy (float32) restatement of the reference algorithm, written from
the maths of the reference -- one function per reference symbol, each citing
the ``file:line`` (relative to the reference checkout) it follows:

    src/models/GATsSPG_architectures/GATs_SuperGlue.py
    src/models/GATsSPG_architectures/GATs.py

Parity pinning: the reference ships no golden vectors or tests for this path
(SURVEY.md §4, §8c).  The oracle is therefore pinned against outputs of the
reference module itself, executed in the build container by
``tests/golden/make_golden.py`` and committed as ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every tensor in those files.

The arithmetic the reference delegates to PyTorch (conv1d, matmul, einsum,
softmax, elu, instance_norm, normalize, max) is restated here with its
published semantics (PyTorch pinned at 1.8.0 by the reference's
environment.yaml:9-11); the golden files were generated with torch 2.10 whose
semantics for these ops are unchanged.

The evaluation is *literal*: the GATs layer multiplies every leaf descriptor by
W exactly as GATs.py:40 does, the MLP concatenates [x, message], etc.  No
algebraic shortcuts of the HIP path are used here, so the two are independent.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32

DEFAULT_HPARAMS = {
    # configs/experiment/train_GATsSPG.yaml:44-60 (the shipped hyper-parameters)
    "descriptor_dim": 256,
    "keypoints_encoder": [32, 64, 128],
    "match_type": "softmax",
    "scale_factor": 0.07,
    "match_threshold": 0.2,
    "include_self": True,
    "additional": False,
    "with_linear_transform": False,
}

GNN_LAYER_NAMES = ["GATs", "self", "cross"] * 4  # GATs_SuperGlue.py:162
NUM_HEADS = 4  # GATs_SuperGlue.py:43 (AttentionPropagation(feature_dim, 4))


# --------------------------------------------------------------------------------------
# elementary ops (PyTorch semantics)
# --------------------------------------------------------------------------------------
def _f32(x):
    return np.ascontiguousarray(x, dtype=F32)


def elu(x):
    """F.elu, alpha=1: x if x > 0 else exp(x) - 1."""
    x = _f32(x)
    return np.where(x > 0, x, np.expm1(np.minimum(x, F32(0)))).astype(F32)


def conv1x1(w, b, x):
    """nn.Conv1d(kernel_size=1): y[b,o,n] = sum_i w[o,i,0] x[b,i,n] + b[o]."""
    w2 = _f32(w).reshape(w.shape[0], w.shape[1])
    y = np.matmul(w2[None], _f32(x))
    if b is not None:
        y = y + _f32(b)[None, :, None]
    return y.astype(F32)


def instance_norm1d(x, eps=1e-5):
    """nn.InstanceNorm1d(affine=False, track_running_stats=False): per (sample, channel)
    statistics over the point axis, biased variance (GATs_SuperGlue.py:126)."""
    x = _f32(x)
    mean = x.mean(axis=2, keepdims=True, dtype=np.float64)
    var = ((x.astype(np.float64) - mean) ** 2).mean(axis=2, keepdims=True)
    return ((x - mean) / np.sqrt(var + eps)).astype(F32)


def softmax(x, axis):
    x = _f32(x)
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return (e / e.sum(axis=axis, keepdims=True, dtype=F32)).astype(F32)


def l2_normalize(x, axis=1, eps=1e-12):
    """F.normalize(p=2): x / max(||x||_2, eps)."""
    x = _f32(x)
    n = np.sqrt((x * x).sum(axis=axis, keepdims=True, dtype=F32))
    return (x / np.maximum(n, F32(eps))).astype(F32)


# --------------------------------------------------------------------------------------
# MLP / KeypointEncoder        GATs_SuperGlue.py:116-140
# --------------------------------------------------------------------------------------
def mlp(sd, prefix, n_layers, x):
    """``MLP(channels)``: Conv1d -> InstanceNorm1d -> ReLU for all but the last Conv1d
    (GATs_SuperGlue.py:116-128).  Sequential indices of the convs are 0,3,6,..."""
    for i in range(n_layers):
        idx = 3 * i
        x = conv1x1(sd[f"{prefix}.{idx}.weight"], sd[f"{prefix}.{idx}.bias"], x)
        if i < n_layers - 1:
            x = np.maximum(instance_norm1d(x), F32(0))
    return x


def keypoint_encoder(sd, prefix, kpts, scores):
    """KeypointEncoder.forward (GATs_SuperGlue.py:138-140): MLP on cat([kpts^T, scores]).
    kpts [b,N,2|3], scores [b,N] -> [b,256,N].  Never called by the reference forward."""
    inp = np.concatenate([np.transpose(_f32(kpts), (0, 2, 1)), _f32(scores)[:, None, :]], axis=1)
    return mlp(sd, f"{prefix}.encoder", 4, inp)


# --------------------------------------------------------------------------------------
# GraphAttentionLayer        GATs.py:35-88
# --------------------------------------------------------------------------------------
def graph_attention_layer(W, a, h_2d, h_3d, include_self=True, additional=False,
                          with_linear_transform=False, alpha=0.2):
    """h_2d [b, N*L, d] leaf descriptors, h_3d [b, N, d] -> [b, N, d]  (GATs.py:35-72)."""
    W, a, h_2d, h_3d = _f32(W), _f32(a), _f32(h_2d), _f32(h_3d)
    b, n1, dim = h_3d.shape
    n2 = h_2d.shape[1]
    num_leaf = int(n2 / n1)  # GATs.py:38
    out_f = W.shape[1]

    wh_2d = np.matmul(h_2d, W)  # GATs.py:40 (the literal 2*N*L*d*d GEMM)
    wh_3d = np.matmul(h_3d, W)  # GATs.py:41

    # _prepare_attentional_mechanism_input, GATs.py:74-88
    s_2d = np.matmul(wh_2d, a[:out_f, :]).reshape(b, n1, num_leaf, 1)
    s_3d = np.matmul(wh_3d, a[out_f:, :])  # [b, n1, 1]
    if include_self:
        s_2d = np.concatenate([s_3d[:, :, None, :], s_2d], axis=2)
    e = s_3d[:, :, None, :] + s_2d
    e = np.where(e > 0, e, F32(alpha) * e).astype(F32)  # LeakyReLU(0.2), GATs.py:30,88
    attention = softmax(e, axis=2)  # GATs.py:44

    h_2d_r = h_2d.reshape(b, n1, num_leaf, dim)
    wh_2d_r = wh_2d.reshape(b, n1, num_leaf, out_f)
    if include_self:  # GATs.py:48-62
        wh_cat = np.concatenate([wh_3d[:, :, None, :], wh_2d_r], axis=2)
        h_cat = np.concatenate([h_3d[:, :, None, :], h_2d_r], axis=2)
        src = wh_cat if with_linear_transform else h_cat
        h_prime = (attention * src).sum(axis=2, dtype=F32)
        if additional:
            h_prime = h_prime + h_3d
    else:  # GATs.py:63-67
        src = wh_2d_r if with_linear_transform else h_2d_r
        h_prime = (attention * src).sum(axis=2, dtype=F32) / F32(2.0)
        h_prime = h_prime + (wh_3d if with_linear_transform else h_3d)
    return elu(h_prime)  # concat=True -> F.elu, GATs.py:69-70


# --------------------------------------------------------------------------------------
# linear attention / MultiHeadedAttention / AttentionPropagation   GATs_SuperGlue.py:69-113
# --------------------------------------------------------------------------------------
def linear_attention(query, key, value):
    """GATs_SuperGlue.py:69-80.  query [b,d,h,n], key/value [b,d,h,m]."""
    eps = F32(1e-6)
    query = elu(query) + F32(1)
    key = elu(key) + F32(1)
    v_length = value.shape[3]
    value = (_f32(value) / F32(v_length)).astype(F32)
    KV = np.einsum("bdhm,bqhm->bqdh", key, value).astype(F32)
    Z = (F32(1) / (np.einsum("bdhm,bdh->bhm", query, key.sum(axis=3, dtype=F32)) + eps)).astype(F32)
    out = np.einsum("bdhm,bqdh,bhm->bqhm", query, KV, Z).astype(F32) * F32(v_length)
    return np.ascontiguousarray(out, dtype=F32)


def multi_headed_attention(sd, prefix, x, source):
    """MultiHeadedAttention.forward (GATs_SuperGlue.py:93-101): channel c = d_idx*4 + head."""
    b = x.shape[0]
    dim = x.shape[1] // NUM_HEADS
    q = conv1x1(sd[f"{prefix}.proj.0.weight"], sd[f"{prefix}.proj.0.bias"], x)
    k = conv1x1(sd[f"{prefix}.proj.1.weight"], sd[f"{prefix}.proj.1.bias"], source)
    v = conv1x1(sd[f"{prefix}.proj.2.weight"], sd[f"{prefix}.proj.2.bias"], source)
    q = q.reshape(b, dim, NUM_HEADS, -1)
    k = k.reshape(b, dim, NUM_HEADS, -1)
    v = v.reshape(b, dim, NUM_HEADS, -1)
    msg = linear_attention(q, k, v).reshape(b, dim * NUM_HEADS, -1)
    return conv1x1(sd[f"{prefix}.merge.weight"], sd[f"{prefix}.merge.bias"], msg)


def attention_propagation(sd, prefix, x, source):
    """AttentionPropagation.forward (GATs_SuperGlue.py:111-113)."""
    message = multi_headed_attention(sd, f"{prefix}.attn", x, source)
    return mlp(sd, f"{prefix}.mlp", 2, np.concatenate([x, message], axis=1))


# --------------------------------------------------------------------------------------
# AttentionalGNN        GATs_SuperGlue.py:48-66
# --------------------------------------------------------------------------------------
def attentional_gnn(sd, hp, desc2d_query, desc3d_db, desc2d_db, trace=None):
    for i, name in enumerate(GNN_LAYER_NAMES):
        p = f"gnn.layers.{i}"
        if name == "GATs":  # :50-54 (transposes to point-major, replaces desc3d_db)
            out = graph_attention_layer(
                sd[f"{p}.W"], sd[f"{p}.a"],
                np.transpose(desc2d_db, (0, 2, 1)), np.transpose(desc3d_db, (0, 2, 1)),
                include_self=hp["include_self"], additional=hp["additional"],
                with_linear_transform=hp["with_linear_transform"])
            desc3d_db = np.ascontiguousarray(np.transpose(out, (0, 2, 1)))
        elif name == "cross":  # :55-59 (both deltas from the pre-update values)
            d0 = attention_propagation(sd, p, desc2d_query, desc3d_db)
            d1 = attention_propagation(sd, p, desc3d_db, desc2d_query)
            desc2d_query, desc3d_db = desc2d_query + d0, desc3d_db + d1
        else:  # 'self', :60-64
            d0 = attention_propagation(sd, p, desc2d_query, desc2d_query)
            d1 = attention_propagation(sd, p, desc3d_db, desc3d_db)
            desc2d_query, desc3d_db = desc2d_query + d0, desc3d_db + d1
        if trace is not None:
            trace.append((i, name, desc2d_query.copy(), desc3d_db.copy()))
    return desc2d_query, desc3d_db


# --------------------------------------------------------------------------------------
# dual-softmax + mutual nearest neighbour       GATs_SuperGlue.py:217-237
# --------------------------------------------------------------------------------------
def dual_softmax(scores):
    """conf = softmax(scores, dim=1) * softmax(scores, dim=2)  (GATs_SuperGlue.py:218)."""
    return (softmax(scores, axis=1) * softmax(scores, axis=2)).astype(F32)


def mutual_nn_match(conf, match_threshold):
    """GATs_SuperGlue.py:220-237, for the whole batch.  torch.max tie-break on CPU is the
    first index, which is also numpy's argmax rule."""
    conf = _f32(conf)
    b, n1, n2 = conf.shape
    idx0 = conf.argmax(axis=2).astype(np.int64)  # [b,n1]
    idx1 = conf.argmax(axis=1).astype(np.int64)  # [b,n2]
    val0 = np.take_along_axis(conf, idx0[:, :, None], axis=2)[:, :, 0]
    ar0 = np.arange(n1, dtype=np.int64)[None]
    ar1 = np.arange(n2, dtype=np.int64)[None]
    mutual0 = ar0 == np.take_along_axis(idx1, idx0, axis=1)
    mutual1 = ar1 == np.take_along_axis(idx0, idx1, axis=1)
    ms0 = np.where(mutual0, val0, F32(0)).astype(F32)
    ms1 = np.where(mutual1, np.take_along_axis(ms0, idx1, axis=1), F32(0)).astype(F32)
    valid0 = mutual0 & (ms0 > F32(match_threshold))
    valid1 = mutual1 & np.take_along_axis(valid0, idx1, axis=1)
    m0 = np.where(valid0, idx0, np.int64(-1))
    m1 = np.where(valid1, idx1, np.int64(-1))
    return {"matches0": m0, "matches1": m1, "matching_scores0": ms0, "matching_scores1": ms1,
            "indices0_raw": idx0, "indices1_raw": idx1}


# --------------------------------------------------------------------------------------
# GATsSuperGlue.forward        GATs_SuperGlue.py:179-241
# --------------------------------------------------------------------------------------
def forward(sd, data, hparams=None, return_intermediates=False):
    """Returns (pred, conf_matrix) exactly like the reference: ``pred`` holds batch element 0
    only (GATs_SuperGlue.py:232-237), ``conf_matrix`` the whole batch.  With
    ``return_intermediates`` a third dict carries mdesc/scores/raw indices for all b."""
    hp = dict(DEFAULT_HPARAMS)
    if hparams:
        hp.update(hparams)
    if hp["match_type"] != "softmax":
        raise NotImplementedError  # :238-239
    kpts2d = _f32(data["keypoints2d"])
    kpts3d = _f32(data["keypoints3d"])
    desc2d_query = _f32(data["descriptors2d_query"])
    desc3d_db = _f32(data["descriptors3d_db"])
    desc2d_db = _f32(data["descriptors2d_db"])

    if kpts2d.shape[1] == 0 or kpts3d.shape[1] == 0:  # :195-203 (a bare dict, int32 matches)
        return {
            "matches0": np.full(kpts2d.shape[:-1], -1, dtype=np.int32)[0],
            "matches1": np.full(kpts3d.shape[:-1], -1, dtype=np.int32)[0],
            "matching_scores0": np.zeros(kpts2d.shape[:-1], dtype=F32)[0],
            "matching_scores1": np.zeros(kpts3d.shape[:-1], dtype=F32)[0],
            "skip_train": True,
        }

    trace = [] if return_intermediates else None
    desc2d_query, desc3d_db = attentional_gnn(sd, hp, desc2d_query, desc3d_db, desc2d_db, trace)
    mdesc2d = l2_normalize(conv1x1(sd["final_proj.weight"], sd["final_proj.bias"], desc2d_query))
    mdesc3d = l2_normalize(conv1x1(sd["final_proj.weight"], sd["final_proj.bias"], desc3d_db))
    scores = (np.einsum("bdn,bdm->bnm", mdesc2d, mdesc3d).astype(F32) / F32(hp["scale_factor"])).astype(F32)
    conf = dual_softmax(scores)
    m = mutual_nn_match(conf, hp["match_threshold"])
    pred = {k: m[k][0] for k in ("matches0", "matches1", "matching_scores0", "matching_scores1")}
    if return_intermediates:
        inter = {"mdesc2d": mdesc2d, "mdesc3d": mdesc3d, "scores": scores, "trace": trace,
                 "batched": m, "desc2d_query": desc2d_query, "desc3d_db": desc3d_db}
        return pred, conf, inter
    return pred, conf
