"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE MODULE.

Run in the build container only (it needs /root/reference, which does not travel to the GPU
box):   PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference ``GATsSuperGlue`` (src/models/GATsSPG_architectures/GATs_SuperGlue.py:143-241)
is imported unmodified, loaded with the seeded synthetic state dict of
``onepose_amd.synthetic`` and executed on CPU in fp32.  Inputs and weights are NOT stored:
they are regenerated from (seed, shape) by ``onepose_amd.synthetic`` wherever the goldens are
consumed; only reference OUTPUTS are committed (a few hundred KB in total).
"""
import json
import os
import sys

sys.dont_write_bytecode = True
REF = os.environ.get("ONEPOSE_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
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

# name -> spec.  'weights': ('random'|'passthrough', seed).  'inputs': kwargs of make_inputs.
CASES = {
    # fixture A: random weights, threshold 0 so the mutual-NN path is non-trivial
    "rand_small": dict(weights=("random", 0), inputs=dict(b=2, n1=48, n2=80, num_leaf=8, seed=1),
                       hp={"match_threshold": 0.0}, store="full"),
    # fixture B: pass-through weights + planted matches -> confidences near 1
    "planted_small": dict(weights=("passthrough", 0), inputs=dict(b=1, n1=64, n2=96, num_leaf=8, seed=2, planted=True),
                          hp={}, store="full"),
    # ragged sizes (nothing a multiple of any tile), num_leaf != 8
    "ragged_leaf3": dict(weights=("random", 3), inputs=dict(b=1, n1=37, n2=53, num_leaf=3, seed=4),
                         hp={"match_threshold": 0.0}, store="full"),
    # smallest size the reference accepts (InstanceNorm1d raises ValueError for a single point)
    "two_points": dict(weights=("random", 5), inputs=dict(b=1, n1=2, n2=2, num_leaf=8, seed=6),
                       hp={"match_threshold": 0.0}, store="full"),
    # the other GraphAttentionLayer flag combinations (GATs.py:56-67)
    "flags_noself": dict(weights=("random", 7), inputs=dict(b=1, n1=40, n2=72, num_leaf=8, seed=8),
                         hp={"match_threshold": 0.0, "include_self": False}, store="full"),
    "flags_wlt": dict(weights=("random", 9), inputs=dict(b=1, n1=40, n2=72, num_leaf=8, seed=10),
                      hp={"match_threshold": 0.0, "with_linear_transform": True}, store="full"),
    "flags_wlt_add": dict(weights=("random", 11), inputs=dict(b=1, n1=40, n2=72, num_leaf=8, seed=12),
                          hp={"match_threshold": 0.0, "with_linear_transform": True, "additional": True}, store="full"),
    "flags_noself_wlt": dict(weights=("random", 13), inputs=dict(b=1, n1=40, n2=72, num_leaf=8, seed=14),
                             hp={"match_threshold": 0.0, "include_self": False, "with_linear_transform": True}, store="full"),
    "flags_add": dict(weights=("random", 15), inputs=dict(b=1, n1=40, n2=72, num_leaf=8, seed=16),
                      hp={"match_threshold": 0.0, "additional": True}, store="full"),
    # config #1 shape (CPU plumbing): 500 / 2000
    "rand_mid": dict(weights=("random", 0), inputs=dict(b=1, n1=500, n2=2000, num_leaf=8, seed=1),
                     hp={"match_threshold": 0.0}, store="summary"),
    "planted_mid": dict(weights=("passthrough", 0), inputs=dict(b=1, n1=500, n2=2000, num_leaf=8, seed=2, planted=True),
                        hp={}, store="summary"),
}


def state_dict_for(kind, seed):
    sd = synthetic.make_state_dict(seed) if kind == "random" else synthetic.make_passthrough_state_dict(seed)
    return sd


def run_case(name, spec):
    hp = dict(BASE_HP)
    hp.update(spec["hp"])
    sd = state_dict_for(*spec["weights"])
    model = GATsSuperGlue(hp).eval()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    data_np = synthetic.make_inputs(**spec["inputs"])
    data = {k: torch.from_numpy(v) for k, v in data_np.items()}

    trace = []

    def hook(idx):
        def fn(mod, inp, out):
            trace.append((idx, out.detach().clone()))
        return fn

    handles = [layer.register_forward_hook(hook(i)) for i, layer in enumerate(model.gnn.layers)]
    captured = {}
    fp = model.final_proj.register_forward_hook(lambda m, i, o: captured.setdefault("fp", []).append(o.detach().clone()))
    with torch.no_grad():
        pred, conf = model(data)
    for h in handles:
        h.remove()
    fp.remove()

    out = {
        "conf_shape": np.array(conf.shape, dtype=np.int64),
        "matches0": pred["matches0"].numpy(), "matches1": pred["matches1"].numpy(),
        "matching_scores0": pred["matching_scores0"].numpy(), "matching_scores1": pred["matching_scores1"].numpy(),
    }
    confn = conf.numpy()
    mdesc2d = torch.nn.functional.normalize(captured["fp"][0], p=2, dim=1).numpy()
    mdesc3d = torch.nn.functional.normalize(captured["fp"][1], p=2, dim=1).numpy()
    out["indices0_raw"] = confn.argmax(axis=2).astype(np.int64)
    out["indices1_raw"] = confn.argmax(axis=1).astype(np.int64)
    if spec["store"] == "full":
        out["conf"] = confn
        out["mdesc2d_sub"] = mdesc2d[:, :, ::3].copy()
        out["mdesc3d_sub"] = mdesc3d[:, :, ::5].copy()
        # per-layer outputs: GATs layers give the new desc3d [b,N,256] (point-major);
        # attention layers are called twice (2D side then 3D side) and give the deltas.
        for j, (idx, t) in enumerate(trace):
            # keep the first 6 points only (GATs output is [b,N,256], the deltas [b,256,N])
            tn = t.numpy()
            out[f"trace_{j:02d}_layer{idx}"] = (tn[:, :6, :] if idx % 3 == 0 else tn[:, :, :6]).copy()
    else:
        out["conf_rowsum"] = confn.sum(axis=2, dtype=np.float64).astype(np.float32)
        out["conf_colsum"] = confn.sum(axis=1, dtype=np.float64).astype(np.float32)
        out["conf_sub"] = confn[:, ::7, ::13].copy()
        out["conf_rowmax"] = confn.max(axis=2)
        out["conf_colmax"] = confn.max(axis=1)
        out["mdesc2d_sub"] = mdesc2d[:, :, ::9].copy()
        out["mdesc3d_sub"] = mdesc3d[:, :, ::37].copy()
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
    n_valid = int((out["matches0"] >= 0).sum())
    print(f"{name}: conf {tuple(conf.shape)} max {confn.max():.4g}  valid matches0 {n_valid}")
    return {"hparams": hp, "weights": list(spec["weights"]), "inputs": spec["inputs"], "store": spec["store"],
            "valid_matches0": n_valid}


def kenc_case():
    """KeypointEncoder (GATs_SuperGlue.py:131-140) -- built by the reference, never called in
    forward; golden for the standalone kernel."""
    hp = dict(BASE_HP)
    sd = synthetic.make_state_dict(0)
    model = GATsSuperGlue(hp).eval()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    rs = np.random.RandomState(21)
    k2 = (rs.rand(2, 50, 2) * 512).astype(np.float32)
    s2 = rs.rand(2, 50).astype(np.float32)
    k3 = (rs.rand(2, 70, 3) - 0.5).astype(np.float32)
    s3 = rs.rand(2, 70).astype(np.float32)
    with torch.no_grad():
        o2 = model.kenc_2d(torch.from_numpy(k2), torch.from_numpy(s2)).numpy()
        o3 = model.kenc_3d(torch.from_numpy(k3), torch.from_numpy(s3)).numpy()
    np.savez_compressed(os.path.join(HERE, "kenc.npz"), kpts2d=k2, scores2d=s2, kpts3d=k3, scores3d=s3, out2d=o2, out3d=o3)
    print("kenc: ", o2.shape, o3.shape)
    return {"weights": ["random", 0], "seed": 21}


def single_point_case():
    """The reference raises ValueError for a single keypoint on either side (InstanceNorm1d
    refuses one spatial element, GATs_SuperGlue.py:126); record the exception type."""
    model = GATsSuperGlue(dict(BASE_HP)).eval()
    data = {k: torch.from_numpy(v) for k, v in synthetic.make_inputs(1, 1, 2, 8, seed=6).items()}
    try:
        with torch.no_grad():
            model(data)
    except Exception as e:  # noqa: BLE001
        print("single point:", type(e).__name__)
        return type(e).__name__
    return None


def empty_case():
    hp = dict(BASE_HP)
    model = GATsSuperGlue(hp).eval()
    data = {k: torch.from_numpy(v) for k, v in synthetic.make_inputs(1, 0, 5, 8, seed=3).items()}
    with torch.no_grad():
        out = model(data)
    assert isinstance(out, dict)
    meta = {k: ([list(v.shape), str(v.dtype)] if torch.is_tensor(v) else v) for k, v in out.items()}
    print("empty:", meta)
    return meta


def main():
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    meta = {"torch": torch.__version__, "numpy": np.__version__, "cases": {}}
    for name, spec in CASES.items():
        meta["cases"][name] = run_case(name, spec)
    meta["kenc"] = kenc_case()
    meta["empty"] = empty_case()
    meta["single_point_raises"] = single_point_case()
    with open(os.path.join(HERE, "golden_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
