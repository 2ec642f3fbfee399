"""Object annotation files -> database tensors (CPU).  The loader is pinned against outputs of the reference's own
data_utils.pad_features3d_random / build_features3d_leaves (tests/golden/make_db_golden.py)."""
import numpy as np
import pytest

from conftest import load_golden
from onepose_amd import synthetic
from onepose_amd.database_io import build_features3d_leaves, build_leaves, load_object_database, pad_features3d_random

NP_SEED = 123       # tests/golden/make_db_golden.py
CASES = {"exact": 0, "padded": 5, "truncated": -7}


@pytest.fixture(scope="module")
def anno():
    return synthetic.make_annotation(n=50, dim=16, seed=4)


@pytest.mark.parametrize("name", list(CASES))
def test_loader_functions_reproduce_the_reference_bit_for_bit(name, anno):
    """Same inputs, same numpy seed -> identical padded averages, identical LEAF TENSORS (the reference's per-point
    np.random.permutation stream), identical scores; n_target equal to / above / below the number of points."""
    g = load_golden("db_loader")
    n_target = anno["idxs"].shape[0] + CASES[name]
    d_avg, s_avg = pad_features3d_random(anno["avg_descriptors"], anno["avg_scores"], n_target)
    np.testing.assert_array_equal(d_avg, g[f"{name}_avg_desc"])
    np.testing.assert_array_equal(s_avg, g[f"{name}_avg_scores"])
    np.random.seed(NP_SEED)            # the reference draws from numpy's global generator: rng=None does too
    d, s = build_features3d_leaves(anno["collect_descriptors"], anno["collect_scores"], anno["idxs"], n_target, 8)
    np.testing.assert_array_equal(d, g[f"{name}_leaves"])
    np.testing.assert_array_equal(s, g[f"{name}_leaf_scores"])
    d2, s2 = build_features3d_leaves(anno["collect_descriptors"], anno["collect_scores"], anno["idxs"], n_target, 8, rng=NP_SEED)
    np.testing.assert_array_equal(d2, g[f"{name}_leaves"])       # private stream, same bits
    assert d.dtype == np.float32 and s.shape == (n_target * 8, 1)


def test_subset_branch_with_fewer_leaves_than_views(anno):
    g = load_golden("db_loader")
    d, s = build_features3d_leaves(anno["collect_descriptors"], anno["collect_scores"], anno["idxs"], anno["idxs"].shape[0], 3,
                                   rng=NP_SEED)
    np.testing.assert_array_equal(d, g["leaf3_leaves"])
    np.testing.assert_array_equal(s, g["leaf3_leaf_scores"])


def test_load_object_database_is_the_reference_pipeline(tmp_path, anno):
    """inference.py:113-130 end to end from the three files: reference leaf choice by default."""
    g = load_golden("db_loader")
    synthetic.write_annotation(str(tmp_path), anno)
    paths = (tmp_path / "anno_3d_average.npz", tmp_path / "anno_3d_collect.npz", tmp_path / "idxs.npy")
    np.random.seed(NP_SEED)
    db = load_object_database(*paths, num_leaf=8, device="cpu")
    np.testing.assert_array_equal(db["descriptors2d_db"][0].numpy(), g["exact_leaves"])
    np.testing.assert_array_equal(db["descriptors3d_db"][0].numpy(), g["exact_avg_desc"])
    db2 = load_object_database(*paths, num_leaf=8, seed=NP_SEED, device="cpu")
    np.testing.assert_array_equal(db2["descriptors2d_db"][0].numpy(), g["exact_leaves"])
    n = anno["idxs"].shape[0]
    assert db["keypoints3d"].shape == (1, n, 3) and db["descriptors3d_db"].shape == (1, 16, n)
    assert db["descriptors2d_db"].shape == (1, 16, n * 8)
    assert db["descriptors2d_db"].is_contiguous() and str(db["descriptors2d_db"].dtype) == "torch.float32"
    vec = load_object_database(*paths, num_leaf=8, seed=1, device="cpu", reference_rng=False)
    assert vec["descriptors2d_db"].shape == (1, 16, n * 8)


def test_vectorised_leaf_selection_semantics():
    """build_leaves (explicit Generator, different stream): every point gets num_leaf leaves, dustbin only when it has too
    few views, no collected descriptor used twice -- the per-point SET semantics of the reference."""
    a = synthetic.make_annotation(n=37, dim=32, seed=0)
    idxs, collect, owner = a["idxs"], a["collect_descriptors"], a["owner"]
    L = 8
    leaves = build_leaves(collect, idxs, L, np.random.default_rng(1))
    assert leaves.shape == (32, len(idxs) * L)
    for i, cnt in enumerate(idxs):
        cols = leaves[:, i * L:(i + 1) * L]
        mine = collect[:, owner == i]
        dust = np.all(cols == 1.0, axis=0)
        assert dust.sum() == max(0, L - cnt)
        real_cols = cols[:, ~dust]
        match = (real_cols[:, :, None] == mine[:, None, :]).all(axis=0)
        assert match.any(axis=1).all() and (match.sum(axis=0) <= 1).all()
    np.testing.assert_array_equal(leaves, build_leaves(collect, idxs, L, np.random.default_rng(1)))
    assert not np.array_equal(leaves, build_leaves(collect, idxs, L, np.random.default_rng(2)))
    # the reference-stream function obeys the same per-point set semantics
    ref, _ = build_features3d_leaves(collect, a["collect_scores"], idxs, len(idxs), L, rng=5)
    for i, cnt in enumerate(idxs):
        assert np.all(ref[:, i * L:(i + 1) * L] == 1.0, axis=0).sum() == max(0, L - cnt)
