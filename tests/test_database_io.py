"""Object annotation files -> database tensors (CPU)."""
import numpy as np

from onepose_amd.database_io import build_leaves, load_object_database


def _fake_anno(tmp_path, n=37, seed=0):
    rs = np.random.RandomState(seed)
    idxs = rs.randint(1, 15, size=n)           # some points have fewer than 8 views, some more
    k = int(idxs.sum())
    collect = rs.standard_normal((256, k)).astype(np.float32)
    owner = np.repeat(np.arange(n), idxs)
    avg = np.stack([collect[:, owner == i].mean(axis=1) for i in range(n)], axis=1)
    kp = rs.rand(n, 3).astype(np.float32)
    np.savez(tmp_path / "anno_3d_average.npz", keypoints3d=kp, descriptors3d=avg, scores3d=np.ones((n, 1), np.float32))
    np.savez(tmp_path / "anno_3d_collect.npz", keypoints3d=kp, descriptors3d=collect, scores3d=np.ones((k, 1), np.float32))
    np.save(tmp_path / "idxs.npy", idxs)
    return idxs, collect, owner


def test_leaf_selection_semantics(tmp_path):
    idxs, collect, owner = _fake_anno(tmp_path)
    L = 8
    leaves = build_leaves(collect, idxs, L, np.random.default_rng(1))
    assert leaves.shape == (256, len(idxs) * L)
    for i, cnt in enumerate(idxs):
        cols = leaves[:, i * L:(i + 1) * L]
        mine = collect[:, owner == i]
        dust = np.all(cols == 1.0, axis=0)
        assert dust.sum() == max(0, L - cnt)                      # dustbin only when a point has too few views
        real_cols = cols[:, ~dust]
        # every real leaf is one of this point's collected descriptors, none is used twice
        match = (real_cols[:, :, None] == mine[:, None, :]).all(axis=0)
        assert match.any(axis=1).all() and (match.sum(axis=0) <= 1).all()
    again = build_leaves(collect, idxs, L, np.random.default_rng(1))
    np.testing.assert_array_equal(leaves, again)                  # reproducible for a given seed
    assert not np.array_equal(leaves, build_leaves(collect, idxs, L, np.random.default_rng(2)))


def test_load_object_database_shapes(tmp_path):
    idxs, _, _ = _fake_anno(tmp_path, n=21, seed=3)
    db = load_object_database(tmp_path / "anno_3d_average.npz", tmp_path / "anno_3d_collect.npz", tmp_path / "idxs.npy",
                              num_leaf=8, seed=0, device="cpu")
    assert db["keypoints3d"].shape == (1, 21, 3)
    assert db["descriptors3d_db"].shape == (1, 256, 21)
    assert db["descriptors2d_db"].shape == (1, 256, 21 * 8)
    assert db["descriptors2d_db"].is_contiguous() and str(db["descriptors2d_db"].dtype) == "torch.float32"
