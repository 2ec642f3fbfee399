"""CPU restatement (numpy, fp32) of the reference SuperPoint extractor forward.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (``onepose_amd/``) may import this file:
it is the checker the HIP extractor is compared against in ``tests/`` (and the ``cpu_baseline``
leg of ``bench.py --extractor``), never the thing measured or shipped.

Parity pinning: ``tests/test_spp_oracle_golden.py`` checks every function below against outputs of
the unmodified reference module (``tests/golden/make_spp_golden.py`` ran
``/root/reference/src/models/extractors/SuperPoint/superpoint.py`` in the build container on the
seeded weights / images of ``onepose_amd.synthetic``).

Each function cites the reference lines it restates (file = superpoint.py).
"""
from __future__ import annotations

import numpy as np

F32 = np.float32

DEFAULT_CONFIG = {  # superpoint.py:104-110
    "descriptor_dim": 256,
    "nms_radius": 4,
    "keypoint_threshold": 0.005,
    "max_keypoints": -1,
    "remove_borders": 4,
}


def conv2d(x, w, b):
    """nn.Conv2d, stride 1, padding k//2 (:119-133).  x [C, H, W], w [O, C, k, k], b [O]."""
    o, c, k, _ = w.shape
    _, h, wd = x.shape
    if k == 1:
        out = w[:, :, 0, 0] @ x.reshape(c, h * wd)
        return (out + b[:, None]).reshape(o, h, wd).astype(F32)
    xp = np.zeros((c, h + 2, wd + 2), F32)
    xp[:, 1:-1, 1:-1] = x
    out = np.zeros((o, h * wd), F32)
    for dy in range(3):
        for dx in range(3):
            out += w[:, :, dy, dx] @ xp[:, dy:dy + h, dx:dx + wd].reshape(c, h * wd)
    return (out + b[:, None]).reshape(o, h, wd).astype(F32)


def relu(x):
    return np.maximum(x, F32(0))


def max_pool2(x):
    """nn.MaxPool2d(2, 2) (:113).  Odd trailing rows / columns are dropped (floor mode)."""
    c, h, w = x.shape
    h2, w2 = h // 2, w // 2
    v = x[:, :h2 * 2, :w2 * 2].reshape(c, h2, 2, w2, 2)
    return v.max(axis=(2, 4))


def encoder(sd, img):
    """Shared encoder (:142-153).  img [H, W] -> [128, H/8, W/8]."""
    x = img[None].astype(F32)
    x = relu(conv2d(x, sd["conv1a.weight"], sd["conv1a.bias"]))
    x = relu(conv2d(x, sd["conv1b.weight"], sd["conv1b.bias"]))
    x = max_pool2(x)
    x = relu(conv2d(x, sd["conv2a.weight"], sd["conv2a.bias"]))
    x = relu(conv2d(x, sd["conv2b.weight"], sd["conv2b.bias"]))
    x = max_pool2(x)
    x = relu(conv2d(x, sd["conv3a.weight"], sd["conv3a.bias"]))
    x = relu(conv2d(x, sd["conv3b.weight"], sd["conv3b.bias"]))
    x = max_pool2(x)
    x = relu(conv2d(x, sd["conv4a.weight"], sd["conv4a.bias"]))
    x = relu(conv2d(x, sd["conv4b.weight"], sd["conv4b.bias"]))
    return x


def score_map(sd, feat):
    """Detector head up to the pixel shuffle (:156-162): softmax over 65 cell channels, dustbin
    dropped, [64, h, w] -> [h*8, w*8]."""
    cpa = relu(conv2d(feat, sd["convPa.weight"], sd["convPa.bias"]))
    logits = conv2d(cpa, sd["convPb.weight"], sd["convPb.bias"])          # [65, h, w]
    e = np.exp(logits - logits.max(axis=0, keepdims=True))
    p = (e / e.sum(axis=0, keepdims=True))[:-1].astype(F32)                # [64, h, w]
    _, h, w = p.shape
    p = p.transpose(1, 2, 0).reshape(h, w, 8, 8)
    return np.ascontiguousarray(p.transpose(0, 2, 1, 3).reshape(h * 8, w * 8))


def dense_descriptors(sd, feat):
    """Descriptor head (:183-185): [D, h, w], L2-normalised over channels (eps 1e-12)."""
    cda = relu(conv2d(feat, sd["convDa.weight"], sd["convDa.bias"]))
    d = conv2d(cda, sd["convDb.weight"], sd["convDb.bias"])
    n = np.sqrt((d.astype(F32) ** 2).sum(axis=0, keepdims=True))
    return (d / np.maximum(n, F32(1e-12))).astype(F32)


def _max_pool_same(x, r):
    """max_pool2d(kernel 2r+1, stride 1, padding r): out-of-image taps are -inf (:51-53)."""
    if r == 0:
        return x.copy()
    h, w = x.shape
    pad = np.full((h + 2 * r, w + 2 * r), -np.inf, x.dtype)
    pad[r:r + h, r:r + w] = x
    rows = pad[:, 0:w].copy()
    for d in range(1, 2 * r + 1):
        np.maximum(rows, pad[:, d:d + w], out=rows)
    out = rows[0:h].copy()
    for d in range(1, 2 * r + 1):
        np.maximum(out, rows[d:d + h], out=out)
    return out


def simple_nms(scores, nms_radius):
    """:47-62, literally (two suppression rounds, exact float equality)."""
    assert nms_radius >= 0
    zeros = np.zeros_like(scores)
    max_mask = scores == _max_pool_same(scores, nms_radius)
    for _ in range(2):
        supp_mask = _max_pool_same(max_mask.astype(F32), nms_radius) > 0
        supp_scores = np.where(supp_mask, zeros, scores)
        new_max_mask = supp_scores == _max_pool_same(supp_scores, nms_radius)
        max_mask = max_mask | (new_max_mask & (~supp_mask))
    return np.where(max_mask, scores, zeros)


def select_keypoints(nms_scores, keypoint_threshold, remove_borders, max_keypoints):
    """:165-180.  Returns (rows/cols int64 [n, 2] in (y, x) order, scores [n]).

    nonzero order is row-major; the border filter keeps that order; when more than
    max_keypoints remain, torch.topk orders by descending score.  torch leaves the order of
    exactly equal scores unspecified -- this oracle (and the HIP path) breaks ties by the lower
    row-major index."""
    h, w = nms_scores.shape
    ys, xs = np.nonzero(nms_scores > F32(keypoint_threshold))
    sc = nms_scores[ys, xs]
    b = remove_borders
    keep = (ys >= b) & (ys < h - b) & (xs >= b) & (xs < w - b)           # :65-70
    ys, xs, sc = ys[keep], xs[keep], sc[keep]
    if max_keypoints >= 0 and max_keypoints < len(sc):                   # :73-78
        order = np.argsort(-sc, kind="stable")[:max_keypoints]
        ys, xs, sc = ys[order], xs[order], sc[order]
    return np.stack([ys, xs], axis=1).astype(np.int64), sc.astype(F32)


def grid_sample_bilinear(desc, gx, gy, align_corners):
    """F.grid_sample(mode='bilinear', padding_mode='zeros') at normalised coords (gx, gy) in
    [-1, 1]; desc [C, h, w] -> [C, n]."""
    c, h, w = desc.shape
    gx = gx.astype(F32)
    gy = gy.astype(F32)
    if align_corners:
        ix = (gx + F32(1)) / F32(2) * F32(w - 1)
        iy = (gy + F32(1)) / F32(2) * F32(h - 1)
    else:
        ix = ((gx + F32(1)) * F32(w) - F32(1)) / F32(2)
        iy = ((gy + F32(1)) * F32(h) - F32(1)) / F32(2)
    x0 = np.floor(ix)
    y0 = np.floor(iy)
    out = np.zeros((c, len(gx)), F32)
    for dy in (0, 1):
        for dx in (0, 1):
            xi = x0 + dx
            yi = y0 + dy
            wgt = (F32(1) - np.abs(ix - xi)) * (F32(1) - np.abs(iy - yi))
            ok = (xi >= 0) & (xi < w) & (yi >= 0) & (yi < h)
            xi_c = np.clip(xi, 0, w - 1).astype(np.int64)
            yi_c = np.clip(yi, 0, h - 1).astype(np.int64)
            out += desc[:, yi_c, xi_c] * (wgt * ok).astype(F32)[None]
    return out


def sample_descriptors(keypoints_xy, dense, s=8, align_corners=True):
    """:81-94.  keypoints_xy float [n, 2] (x, y); dense [C, h, w] (already normalised).

    The reference passes align_corners=True only when ``int(torch.__version__[2]) > 2`` (:87), i.e.
    on the torch 1.3-1.9 builds OnePose pins; on torch >= 1.10 / most 2.x the same line silently
    falls back to align_corners=False.  Both behaviours are restated; the caller chooses."""
    c, h, w = dense.shape
    kp = keypoints_xy.astype(F32) - F32(s / 2) + F32(0.5)
    kp = kp / np.array([w * s - s / 2 - 0.5, h * s - s / 2 - 0.5], F32)[None]
    kp = kp * F32(2) - F32(1)
    d = grid_sample_bilinear(dense, kp[:, 0], kp[:, 1], align_corners)
    n = np.sqrt((d ** 2).sum(axis=0, keepdims=True))
    return (d / np.maximum(n, F32(1e-12))).astype(F32)


def forward(sd, image, config=None, align_corners=True, return_intermediates=False):
    """SuperPoint.forward (:140-197) for a batch [b, 1, H, W]; returns the reference's dict of
    per-image lists (keypoints float [n, 2] (x, y), scores [n], descriptors [D, n])."""
    cfg = {**DEFAULT_CONFIG, **(config or {})}
    out = {"keypoints": [], "scores": [], "descriptors": []}
    inter = []
    for img in np.asarray(image, F32)[:, 0]:
        feat = encoder(sd, img)
        sm = score_map(sd, feat)
        nms = simple_nms(sm, cfg["nms_radius"])
        yx, sc = select_keypoints(nms, cfg["keypoint_threshold"], cfg["remove_borders"], cfg["max_keypoints"])
        kp = yx[:, ::-1].astype(F32)                                        # :180 (h, w) -> (x, y)
        dense = dense_descriptors(sd, feat)
        desc = sample_descriptors(kp, dense, 8, align_corners)
        out["keypoints"].append(kp)
        out["scores"].append(sc)
        out["descriptors"].append(desc)
        inter.append({"feat": feat, "score_map": sm, "nms": nms, "dense": dense})
    return (out, inter) if return_intermediates else out
