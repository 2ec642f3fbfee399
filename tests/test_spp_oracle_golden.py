"""Pin the numpy SuperPoint oracle against golden vectors produced by the reference module
(tests/golden/make_spp_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from onepose_amd import synthetic
from oracle import superpoint_oracle as so

with open(os.path.join(GOLDEN_DIR, "spp_golden_meta.json")) as f:
    META = {k: v for k, v in json.load(f).items() if not k.startswith("_")}

ATOL_SCORE = 1e-5   # fp32 re-association noise (numpy tap-by-tap conv vs ATen), amplified by the 65-way softmax
ATOL_DESC = 2e-6


def load(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, f"spp_{name}.npz")))


def run(name):
    m = META[name]
    sd = synthetic.make_spp_state_dict(m["wseed"])
    img = synthetic.make_image(**m["img"])
    return so.forward(sd, img, m["cfg"], align_corners=m["align"], return_intermediates=True)


def match_rows(kp, gk):
    """index array p with kp[p] == gk (both are sets of distinct pixel coordinates)."""
    key = lambda a: (a[:, 1].astype(np.int64) << 20) | a[:, 0].astype(np.int64)
    ka, kb = key(kp), key(gk)
    oa, ob = np.argsort(ka), np.argsort(kb)
    assert np.array_equal(ka[oa], kb[ob]), "keypoint sets differ"
    p = np.empty(len(gk), np.int64)
    p[ob] = oa
    return p


@pytest.mark.parametrize("name", ["tiny_default", "rect_pipeline", "rect_noalign", "r0_thr", "odd_size"])
def test_oracle_matches_reference(name):
    g = load(name)
    out, inter = run(name)
    for i in range(META[name]["img"]["b"]):
        np.testing.assert_array_equal(out["keypoints"][i], g[f"keypoints{i}"])     # same set, same (row-major) order
        np.testing.assert_allclose(out["scores"][i], g[f"scores{i}"], atol=ATOL_SCORE)
        np.testing.assert_allclose(out["descriptors"][i], g[f"descriptors{i}"], atol=ATOL_DESC)
        assert out["keypoints"][i].dtype == np.float32 and out["keypoints"][i].shape[1] == 2
    # dense stages against the reference's forward hooks
    raw = g["dense_raw"]
    nrm = raw / np.maximum(np.sqrt((raw ** 2).sum(axis=1, keepdims=True)), 1e-12)
    for i in range(META[name]["img"]["b"]):
        np.testing.assert_allclose(inter[i]["dense"], nrm[i], atol=ATOL_DESC)
        lg = g["logits"][i]
        e = np.exp(lg - lg.max(axis=0, keepdims=True))
        p = (e / e.sum(axis=0, keepdims=True))[:-1]
        h, w = p.shape[1:]
        ref_map = p.transpose(1, 2, 0).reshape(h, w, 8, 8).transpose(0, 2, 1, 3).reshape(h * 8, w * 8)
        np.testing.assert_allclose(inter[i]["score_map"], ref_map, atol=ATOL_SCORE)


@pytest.mark.parametrize("name", ["topk50", "crop512"])
def test_oracle_topk_cases(name):
    """max_keypoints engaged: the same keypoint SET as the reference, scores descending; the order of
    near-equal scores may differ (scores differ by fp32 noise between numpy and ATen)."""
    g = load(name)
    out, _ = run(name)
    kp, sc, de = out["keypoints"][0], out["scores"][0], out["descriptors"][0]
    gk, gs = g["keypoints0"], g["scores0"]
    assert len(kp) == META[name]["cfg"]["max_keypoints"] == len(gk)
    assert np.all(np.diff(sc) <= 0)
    p = match_rows(kp, gk)
    np.testing.assert_allclose(sc[p], gs, atol=ATOL_SCORE)
    if "descriptors0" in g:
        np.testing.assert_allclose(de[:, p], g["descriptors0"], atol=ATOL_DESC)
    else:
        np.testing.assert_allclose(de[:, p[::8]], g["descriptors0_every8"], atol=ATOL_DESC)


def test_nms_properties():
    """simple_nms keeps isolated maxima, suppresses neighbours, and treats plateaus as the reference does."""
    rs = np.random.RandomState(0)
    s = rs.rand(40, 50).astype(np.float32)
    out = so.simple_nms(s, 3)
    ys, xs = np.nonzero(out)
    assert len(ys) > 0
    assert out[np.unravel_index(np.argmax(s), s.shape)] == s.max()          # the global maximum survives
    pts = np.stack([ys, xs], 1)
    d = np.abs(pts[:, None] - pts[None]).max(-1) + np.eye(len(pts), dtype=np.int64) * 99
    assert d.min() > 3                                                       # distinct values: survivors > radius apart
    first = s == so._max_pool_same(s, 3)
    assert np.all(out[first] == s[first])                                    # every window maximum is kept
    assert np.array_equal(so.simple_nms(s, 0), s)
    flat = np.full((9, 9), 0.5, np.float32)
    assert np.array_equal(so.simple_nms(flat, 2), flat)      # s == maxpool(s) everywhere on a plateau


def test_select_ties_and_borders():
    s = np.zeros((16, 16), np.float32)
    s[5, 5] = s[5, 9] = s[9, 5] = 0.5
    s[2, 8] = 0.9          # inside a border of 4 -> removed
    s[10, 10] = 0.7
    yx, sc = so.select_keypoints(s, 0.005, 4, 2)
    assert yx.tolist() == [[10, 10], [5, 5]] and sc.tolist() == [np.float32(0.7), np.float32(0.5)]
    yx, _ = so.select_keypoints(s, 0.005, 4, -1)
    assert yx.tolist() == [[5, 5], [5, 9], [9, 5], [10, 10]]
    yx, _ = so.select_keypoints(s, 0.005, 0, -1)
    assert [2, 8] in yx.tolist()


@pytest.mark.parametrize("name", ["tiny_default", "rect_pipeline", "rect_noalign", "odd_size"])
def test_torch_restatement_matches_reference(name):
    """oracle/torch_superpoint_oracle.py (the stock-PyTorch baseline of bench.py --extractor --torch-eager)."""
    import torch
    from oracle import torch_superpoint_oracle as tso
    m = META[name]
    sd = {k: torch.from_numpy(v) for k, v in synthetic.make_spp_state_dict(m["wseed"]).items()}
    out = tso.forward(sd, torch.from_numpy(synthetic.make_image(**m["img"])), m["cfg"], align_corners=m["align"])
    g = load(name)
    for i in range(m["img"]["b"]):
        np.testing.assert_array_equal(out["keypoints"][i].numpy(), g[f"keypoints{i}"])
        np.testing.assert_allclose(out["scores"][i].numpy(), g[f"scores{i}"], atol=1e-6)
        np.testing.assert_allclose(out["descriptors"][i].numpy(), g[f"descriptors{i}"], atol=1e-6)
