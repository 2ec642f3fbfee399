"""Torch restatement of the reference GATsSPG forward (literal evaluation order, stock torch ops).

TEST / BASELINE INFRASTRUCTURE ONLY -- same rules as gatsspg_oracle.py: never imported by the product path.
Purpose: (1) a second, independent oracle (pinned against the reference goldens in
tests/test_oracle_golden.py); (2) run on the MI355X through PyTorch-ROCm it is what "the reference on
this GPU" costs -- every op is a stock ATen / rocBLAS / MIOpen kernel, one launch per op, exactly like the
reference module (~650-700 device-op launches per forward).  bench.py --torch-eager
times it.  Cites: src/models/GATsSPG_architectures/GATs_SuperGlue.py, GATs.py.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

LAYER_NAMES = ["GATs", "self", "cross"] * 4  # GATs_SuperGlue.py:162
HEADS = 4


def _conv(sd, prefix, x):  # nn.Conv1d(kernel_size=1)
    return F.conv1d(x, sd[prefix + ".weight"], sd[prefix + ".bias"])


def _mlp(sd, prefix, n, x):  # GATs_SuperGlue.py:116-128
    for i in range(n):
        x = _conv(sd, f"{prefix}.{3 * i}", x)
        if i < n - 1:
            x = F.relu(F.instance_norm(x))
    return x


def _linear_attention(q, k, v):  # GATs_SuperGlue.py:69-80
    q, k = F.elu(q) + 1, F.elu(k) + 1
    n = v.size(3)
    v = v / n
    kv = torch.einsum("bdhm,bqhm->bqdh", k, v)
    z = 1 / (torch.einsum("bdhm,bdh->bhm", q, k.sum(3)) + 1e-6)
    return (torch.einsum("bdhm,bqdh,bhm->bqhm", q, kv, z) * n).contiguous()


def _attn_prop(sd, p, x, src):  # GATs_SuperGlue.py:93-113
    b = x.size(0)
    q, k, v = (_conv(sd, f"{p}.attn.proj.{j}", t).view(b, 64, HEADS, -1) for j, t in ((0, x), (1, src), (2, src)))
    msg = _conv(sd, f"{p}.attn.merge", _linear_attention(q, k, v).view(b, 256, -1))
    return _mlp(sd, f"{p}.mlp", 2, torch.cat([x, msg], dim=1))


def _gats(W, a, h2, h3, include_self, additional, wlt):  # GATs.py:35-88
    b, n1, d = h3.shape
    L = h2.shape[1] // n1
    wh2, wh3 = h2 @ W, h3 @ W
    s2 = (wh2 @ a[:d]).view(b, n1, L, 1)
    s3 = wh3 @ a[d:]
    if include_self:
        s2 = torch.cat([s3.unsqueeze(2), s2], dim=2)
    att = F.softmax(F.leaky_relu(s3.unsqueeze(2) + s2, 0.2), dim=2)
    h2r, wh2r = h2.view(b, n1, L, d), wh2.view(b, n1, L, d)
    if include_self:
        src = torch.cat([(wh3 if wlt else h3).unsqueeze(2), wh2r if wlt else h2r], dim=2)
        hp = torch.einsum("bncd,bncq->bnq", att, src)
        if additional:
            hp = hp + h3
    else:
        hp = torch.einsum("bncd,bncq->bnq", att, wh2r if wlt else h2r) / 2.0 + (wh3 if wlt else h3)
    return F.elu(hp)


@torch.no_grad()
def forward(sd, data, hp):
    """sd: dict name -> tensor (reference state_dict names); returns (pred, conf) like the reference."""
    x, y, lf = (data[k].float() for k in ("descriptors2d_query", "descriptors3d_db", "descriptors2d_db"))
    for i, name in enumerate(LAYER_NAMES):  # GATs_SuperGlue.py:48-66
        p = f"gnn.layers.{i}"
        if name == "GATs":
            y = _gats(sd[p + ".W"], sd[p + ".a"], lf.transpose(1, 2), y.transpose(1, 2), hp["include_self"],
                      hp["additional"], hp["with_linear_transform"]).transpose(1, 2)
        elif name == "self":
            x, y = x + _attn_prop(sd, p, x, x), y + _attn_prop(sd, p, y, y)
        else:
            x, y = x + _attn_prop(sd, p, x, y), y + _attn_prop(sd, p, y, x)
    mx = F.normalize(_conv(sd, "final_proj", x), p=2, dim=1)  # :209-213
    my = F.normalize(_conv(sd, "final_proj", y), p=2, dim=1)
    scores = torch.einsum("bdn,bdm->bnm", mx, my) / hp["scale_factor"]  # :217
    conf = F.softmax(scores, 1) * F.softmax(scores, 2)  # :218
    max0, max1 = conf.max(2), conf.max(1)  # :220-237
    i0, i1 = max0.indices, max1.indices
    ar0 = torch.arange(i0.shape[1], device=conf.device)[None]
    ar1 = torch.arange(i1.shape[1], device=conf.device)[None]
    mutual0, mutual1 = ar0 == i1.gather(1, i0), ar1 == i0.gather(1, i1)
    zero = conf.new_tensor(0)
    ms0 = torch.where(mutual0, max0.values, zero)
    ms1 = torch.where(mutual1, ms0.gather(1, i1), zero)
    valid0 = mutual0 & (ms0 > hp["match_threshold"])
    valid1 = mutual1 & valid0.gather(1, i1)
    m0 = torch.where(valid0, i0, i0.new_tensor(-1))
    m1 = torch.where(valid1, i1, i1.new_tensor(-1))
    return {"matches0": m0[0], "matches1": m1[0], "matching_scores0": ms0[0], "matching_scores1": ms1[0]}, conf
