"""Stock-PyTorch restatement of the reference SuperPoint forward (same algorithm, ATen / MIOpen ops).

TEST / BASELINE INFRASTRUCTURE ONLY (see oracle/superpoint_oracle.py): used by the CPU tests as a second,
independent check against the reference goldens, and by ``bench.py --extractor --torch-eager`` as the informative
"reference algorithm through stock PyTorch-ROCm on this GPU" baseline.  Never imported by the product path.

Follows src/models/extractors/SuperPoint/superpoint.py: encoder :142-153, detector head :156-162,
simple_nms :47-62, keypoint selection :165-180, descriptor head and sampling :81-94,183-190.
"""
import torch
import torch.nn.functional as F

DEFAULT_CONFIG = {"descriptor_dim": 256, "nms_radius": 4, "keypoint_threshold": 0.005, "max_keypoints": -1,
                  "remove_borders": 4}


def _conv(sd, name, x, relu=True):
    w = sd[name + ".weight"]
    y = F.conv2d(x, w, sd[name + ".bias"], padding=w.shape[-1] // 2)
    return F.relu(y) if relu else y


def _window_max(t, r):
    """Maximum over the (2r+1) x (2r+1) window around every pixel, as two 1-D passes (rows, then columns) -- the separable form
    the HIP nms_kernel evaluates in LDS.  Out-of-image neighbours never win (max_pool2d pads with -inf)."""
    k = 2 * r + 1
    return F.max_pool2d(F.max_pool2d(t, (1, k), 1, (0, r)), (k, 1), 1, (r, 0))


def _nms(scores, r):
    """Keypoint non-maximum suppression with the semantics of the reference's simple_nms (superpoint.py:47-62): window maxima
    are kept; twice more, pixels farther than r from every kept pixel compete among themselves and their window maxima join."""
    if r == 0:
        return scores
    keep = scores == _window_max(scores, r)
    for _ in range(2):
        near_kept = _window_max(keep.to(scores.dtype), r) > 0
        rest = scores.masked_fill(near_kept, 0.0)
        keep = keep | ((rest == _window_max(rest, r)) & ~near_kept)
    return scores.masked_fill(~keep, 0.0)


def forward(sd, image, config=None, align_corners=True):
    cfg = {**DEFAULT_CONFIG, **(config or {})}
    x = _conv(sd, "conv1a", image)
    x = F.max_pool2d(_conv(sd, "conv1b", x), 2, 2)
    x = _conv(sd, "conv2a", x)
    x = F.max_pool2d(_conv(sd, "conv2b", x), 2, 2)
    x = _conv(sd, "conv3a", x)
    x = F.max_pool2d(_conv(sd, "conv3b", x), 2, 2)
    x = _conv(sd, "conv4a", x)
    x = _conv(sd, "conv4b", x)
    s = F.softmax(_conv(sd, "convPb", _conv(sd, "convPa", x), relu=False), 1)[:, :-1]
    b, _, h, w = s.shape
    s = s.permute(0, 2, 3, 1).reshape(b, h, w, 8, 8).permute(0, 1, 3, 2, 4).reshape(b, h * 8, w * 8)
    s = _nms(s, cfg["nms_radius"])
    dense = F.normalize(_conv(sd, "convDb", _conv(sd, "convDa", x), relu=False), p=2, dim=1)
    out = {"keypoints": [], "scores": [], "descriptors": []}
    bd, k = cfg["remove_borders"], cfg["max_keypoints"]
    for i in range(b):
        yx = torch.nonzero(s[i] > cfg["keypoint_threshold"])
        sc = s[i][yx[:, 0], yx[:, 1]]
        keep = (yx[:, 0] >= bd) & (yx[:, 0] < h * 8 - bd) & (yx[:, 1] >= bd) & (yx[:, 1] < w * 8 - bd)
        yx, sc = yx[keep], sc[keep]
        if 0 <= k < len(sc):
            sc, idx = torch.topk(sc, k, dim=0)
            yx = yx[idx]
        kp = torch.flip(yx, [1]).float()
        g = (kp - 4 + 0.5) / torch.tensor([w * 8 - 4 - 0.5, h * 8 - 4 - 0.5], device=kp.device, dtype=kp.dtype)[None]
        g = g * 2 - 1
        d = F.grid_sample(dense[i:i + 1], g.view(1, 1, -1, 2), mode="bilinear", align_corners=align_corners)
        d = F.normalize(d.reshape(1, dense.shape[1], -1), p=2, dim=1)[0]
        out["keypoints"].append(kp)
        out["scores"].append(sc)
        out["descriptors"].append(d)
    return out
