"""Summary goldens at the BENCHMARKED shapes, produced by RUNNING THE REFERENCE MODULE.

Run in the build container only (needs /root/reference):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_bench_golden.py

The reference ``GATsSuperGlue`` (src/models/GATsSPG_architectures/GATs_SuperGlue.py:143-241) is imported
unmodified and executed on CPU in fp32 on the seeded synthetic weights / inputs of ``onepose_amd.synthetic``:

  head_rand     BASELINE configs[1]  N_2D=1000 N_3D=7000 b=1, random weights, threshold 0   (the bench.py workload)
  head_planted  same shape, pass-through weights + planted matches, threshold 0.2
  head_b8       BASELINE configs[2]'s per-GPU share: b=8 frames of 1000/7000, random weights, threshold 0
  stress_rand   BASELINE configs[4]  N_3D=20000 dense cloud, b=1, random weights, threshold 0
  stress_planted  same shape, planted matches
  stress_b4     BASELINE configs[4]'s per-GPU share: b=4 frames of 1000/20000, random weights, threshold 0
  real_rand     OnePose's own operating point (configs[0]; test_GATsSPG.yaml:21 caps N_3D at 2500, train_GATsSPG.yaml:78-79
                uses 1000 x 2000): N_2D=500 N_3D=2000 b=1, random weights, threshold 0
  real_b8       the same shape, 8 frames per step

`python tests/golden/make_bench_golden.py stress_b4 ...` regenerates only the named cases (the meta file keeps the others).

Only reference OUTPUT summaries are stored (inputs and weights are regenerated from seeds): raw row / column arg-max
indices of every sample, ``pred`` of sample 0, a strided sub-sample of ``conf``, its row/column maxima and sums, and the
relative gap between the best and second-best entry of every row / column (the parity tests allow an arg-max to
differ from the reference only where that gap is below fp32 resolution, and report how many such places exist).
"""
import json
import os
import sys

sys.dont_write_bytecode = True
REF = os.environ.get("ONEPOSE_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("GOLDEN_OUT", HERE)
sys.path.insert(0, REF)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from src.models.GATsSPG_architectures.GATs_SuperGlue import GATsSuperGlue  # noqa: E402  (reference)
from onepose_amd import synthetic  # noqa: E402

BASE_HP = {
    "descriptor_dim": 256, "keypoints_encoder": [32, 64, 128], "match_type": "softmax",
    "scale_factor": 0.07, "match_threshold": 0.2, "include_self": True, "additional": False,
    "with_linear_transform": False,
}

# 'sub': (row stride, column stride) of the stored conf sub-sample
CASES = {
    "head_rand": dict(weights=("random", 0), inputs=dict(b=1, n1=1000, n2=7000, num_leaf=8, seed=1),
                      hp={"match_threshold": 0.0}, sub=(7, 13)),
    "head_planted": dict(weights=("passthrough", 0), inputs=dict(b=1, n1=1000, n2=7000, num_leaf=8, seed=2, planted=True),
                         hp={}, sub=(7, 13)),
    "head_b8": dict(weights=("random", 0), inputs=dict(b=8, n1=1000, n2=7000, num_leaf=8, seed=3),
                    hp={"match_threshold": 0.0}, sub=(37, 41)),
    "stress_rand": dict(weights=("random", 0), inputs=dict(b=1, n1=1000, n2=20000, num_leaf=8, seed=5),
                        hp={"match_threshold": 0.0}, sub=(11, 17)),
    "stress_planted": dict(weights=("passthrough", 0), inputs=dict(b=1, n1=1000, n2=20000, num_leaf=8, seed=6, planted=True),
                           hp={}, sub=(11, 17)),
    "stress_b4": dict(weights=("random", 0), inputs=dict(b=4, n1=1000, n2=20000, num_leaf=8, seed=7),
                      hp={"match_threshold": 0.0}, sub=(37, 53)),
    "real_rand": dict(weights=("random", 0), inputs=dict(b=1, n1=500, n2=2000, num_leaf=8, seed=8),
                      hp={"match_threshold": 0.0}, sub=(5, 7)),
    "real_b8": dict(weights=("random", 0), inputs=dict(b=8, n1=500, n2=2000, num_leaf=8, seed=9),
                    hp={"match_threshold": 0.0}, sub=(11, 13)),
}


def top2_rel_gap(c, axis):
    """(best - second best) / best along `axis` (0 where best == 0)."""
    part = np.partition(c, -2, axis=axis)
    best = np.take(part, -1, axis=axis)
    second = np.take(part, -2, axis=axis)
    with np.errstate(divide="ignore", invalid="ignore"):
        g = np.where(best > 0, (best - second) / best, 0.0)
    return g.astype(np.float32)


def run_case(name, spec):
    hp = dict(BASE_HP)
    hp.update(spec["hp"])
    kind, seed = spec["weights"]
    sd = synthetic.make_state_dict(seed) if kind == "random" else synthetic.make_passthrough_state_dict(seed)
    model = GATsSuperGlue(hp).eval()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    data = {k: torch.from_numpy(v) for k, v in synthetic.make_inputs(**spec["inputs"]).items()}
    with torch.no_grad():
        pred, conf = model(data)
    c = conf.numpy()
    rs, cs = spec["sub"]
    out = {
        "conf_shape": np.array(c.shape, dtype=np.int64),
        "matches0": pred["matches0"].numpy(), "matches1": pred["matches1"].numpy(),
        "matching_scores0": pred["matching_scores0"].numpy(), "matching_scores1": pred["matching_scores1"].numpy(),
        "indices0_raw": c.argmax(axis=2).astype(np.int32), "indices1_raw": c.argmax(axis=1).astype(np.int32),
        "conf_sub": c[:, ::rs, ::cs].copy(),
        "conf_rowmax": c.max(axis=2), "conf_colmax": c.max(axis=1),
        "conf_rowsum": c.sum(axis=2, dtype=np.float64).astype(np.float32),
        "conf_colsum": c.sum(axis=1, dtype=np.float64).astype(np.float32),
        "row_top2_rel_gap": top2_rel_gap(c, 2), "col_top2_rel_gap": top2_rel_gap(c, 1),
    }
    np.savez_compressed(os.path.join(OUT, f"bench_{name}.npz"), **out)
    n_valid = int((out["matches0"] >= 0).sum())
    print(f"{name}: conf {c.shape} max {c.max():.4g}  valid matches0 {n_valid}  "
          f"min top-2 gap rows {out['row_top2_rel_gap'].min():.2e} cols {out['col_top2_rel_gap'].min():.2e}", flush=True)
    return {"hparams": hp, "weights": list(spec["weights"]), "inputs": spec["inputs"], "sub": list(spec["sub"]),
            "valid_matches0": n_valid}


def main():
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    meta = {"torch": torch.__version__, "numpy": np.__version__, "cases": {}}
    only = sys.argv[1:]
    meta_path = os.path.join(OUT, "bench_golden_meta.json")
    if only and os.path.exists(meta_path):
        with open(meta_path) as f:
            meta["cases"] = json.load(f)["cases"]
    for name, spec in CASES.items():
        if not only or name in only:
            meta["cases"][name] = run_case(name, spec)
    with open(os.path.join(OUT, "bench_golden_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
