"""Golden for the database loader, produced by RUNNING THE REFERENCE's data_utils (src/utils/data_utils.py:143-205).

Run in the build container only (needs /root/reference):   PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_db_golden.py

data_utils imports cv2 and loguru at module top but the two functions used here touch neither: both are stubbed in
sys.modules (neither package is installed in this image).  The synthetic annotation (onepose_amd.synthetic.make_annotation)
is regenerated from its seed wherever the golden is consumed; only the reference's OUTPUTS are stored.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
REF = os.environ.get("ONEPOSE_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("GOLDEN_OUT", HERE)
sys.path.insert(0, REF)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
for name in ("cv2", "loguru"):
    if name not in sys.modules:
        m = types.ModuleType(name)
        m.logger = None
        sys.modules[name] = m

import numpy as np  # noqa: E402

from src.utils import data_utils  # noqa: E402  (reference)
from onepose_amd import synthetic  # noqa: E402

CASES = {"exact": 0, "padded": 5, "truncated": -7}   # n_target_shape - number of 3D points
NP_SEED = 123


def main():
    anno = synthetic.make_annotation(n=50, dim=16, seed=4)
    out = {}
    for name, delta in CASES.items():
        n_target = anno["idxs"].shape[0] + delta
        np.random.seed(NP_SEED)
        d_avg, s_avg = data_utils.pad_features3d_random(anno["avg_descriptors"], anno["avg_scores"], n_target)
        d_clt, s_clt = data_utils.build_features3d_leaves(anno["collect_descriptors"], anno["collect_scores"], anno["idxs"], n_target, 8)
        out[f"{name}_avg_desc"], out[f"{name}_avg_scores"] = d_avg.numpy(), s_avg.numpy()
        out[f"{name}_leaves"], out[f"{name}_leaf_scores"] = d_clt.numpy(), s_clt.numpy()
        print(name, tuple(d_avg.shape), tuple(d_clt.shape))
    np.random.seed(NP_SEED)                       # num_leaf = 3: most points have MORE views than leaves (subset branch)
    d3, s3 = data_utils.build_features3d_leaves(anno["collect_descriptors"], anno["collect_scores"], anno["idxs"], anno["idxs"].shape[0], 3)
    out["leaf3_leaves"], out["leaf3_leaf_scores"] = d3.numpy(), s3.numpy()
    np.savez_compressed(os.path.join(OUT, "db_loader.npz"), **out)


if __name__ == "__main__":
    main()
