"""Generate the SuperPoint golden vectors (tests/golden/spp_*.npz) by RUNNING THE REFERENCE MODULE.

Build container only (needs /root/reference):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_spp_golden.py

The reference ``SuperPoint`` (src/models/extractors/SuperPoint/superpoint.py:96-197) is imported
unmodified, loaded with the seeded synthetic weights of ``onepose_amd.synthetic`` and run on CPU in
fp32.  Weights and images are regenerated from seeds where the goldens are consumed; only reference
OUTPUTS are stored.  Forward hooks (not edits) on ``convPb`` / ``convDb`` capture the detector
logits and raw dense descriptors so the dense stages can be pinned separately from the discrete
keypoint selection.

``sample_descriptors`` (:87) picks ``align_corners`` from ``int(torch.__version__[2]) > 2``.  On the
torch 1.x builds OnePose targets that is True; on this container's torch 2.10 it is False.  Cases
tagged ``align=True`` are generated with ``torch.__version__`` temporarily presented as "1.8.0" to
the reference (its only use of the string), so both behaviours are pinned.
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

from src.models.extractors.SuperPoint.superpoint import SuperPoint  # noqa: E402  (reference)
from onepose_amd import synthetic  # noqa: E402

PIPELINE_CFG = {"descriptor_dim": 256, "nms_radius": 3, "max_keypoints": 4096}   # src/sfm/extract_features.py:21-26

CASES = {
    # default config (:104-110): radius 4, no top-k
    "tiny_default": dict(wseed=0, img=dict(b=1, h=64, w=64, seed=1), cfg={}, align=True, store="full"),
    # non-square, batch of 2, the pipeline's config
    "rect_pipeline": dict(wseed=2, img=dict(b=2, h=120, w=160, seed=3), cfg=PIPELINE_CFG, align=True, store="full"),
    # top-k engaged
    "topk50": dict(wseed=4, img=dict(b=1, h=96, w=96, seed=5), cfg={"nms_radius": 2, "max_keypoints": 50},
                   align=True, store="full"),
    # align_corners=False flavour of the version hack
    "rect_noalign": dict(wseed=2, img=dict(b=1, h=120, w=160, seed=3), cfg=PIPELINE_CFG, align=False, store="full"),
    # no NMS, higher threshold, no border removal
    "r0_thr": dict(wseed=6, img=dict(b=1, h=64, w=80, seed=7),
                   cfg={"nms_radius": 0, "keypoint_threshold": 0.05, "remove_borders": 0}, align=True, store="full"),
    # sizes that are not multiples of 8: floor-mode poolings, keypoints only in the (h*8) x (w*8) top-left part (:160-162)
    "odd_size": dict(wseed=8, img=dict(b=2, h=75, w=101, seed=9), cfg=PIPELINE_CFG, align=True, store="full"),
    # the pipeline's shape (512x512 crop, :extract_features.py:15-19)
    "crop512": dict(wseed=0, img=dict(b=1, h=512, w=512, seed=11), cfg=PIPELINE_CFG, align=True, store="sub"),
}


def run_case(spec):
    sd = synthetic.make_spp_state_dict(spec["wseed"])
    model = SuperPoint(dict(spec["cfg"])).eval()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    img = torch.from_numpy(synthetic.make_image(**spec["img"]))
    cap = {}
    hooks = [model.convPb.register_forward_hook(lambda m, i, o: cap.__setitem__("logits", o.detach().numpy().copy())),
             model.convDb.register_forward_hook(lambda m, i, o: cap.__setitem__("dense_raw", o.detach().numpy().copy()))]
    real = torch.__version__
    try:
        if spec["align"]:
            torch.__version__ = "1.8.0"
        with torch.no_grad():
            out = model(img)
    finally:
        torch.__version__ = real
        for h in hooks:
            h.remove()
    return out, cap


def main():
    meta = {}
    for name, spec in CASES.items():
        out, cap = run_case(spec)
        arrays = {}
        b = spec["img"]["b"]
        counts = []
        for i in range(b):
            kp = out["keypoints"][i].numpy()
            sc = out["scores"][i].numpy()
            de = out["descriptors"][i].numpy()
            counts.append(int(kp.shape[0]))
            arrays[f"keypoints{i}"] = kp
            arrays[f"scores{i}"] = sc
            if spec["store"] == "full":
                arrays[f"descriptors{i}"] = de
            else:
                arrays[f"descriptors{i}_every8"] = de[:, ::8].copy()
        if spec["store"] == "full":
            arrays["logits"] = cap["logits"]
            arrays["dense_raw"] = cap["dense_raw"]
        else:
            arrays["logits_every4"] = cap["logits"][:, :, ::4, ::4].copy()
            arrays["dense_raw_every4"] = cap["dense_raw"][:, ::4, ::4, ::4].copy()
        np.savez_compressed(os.path.join(HERE, f"spp_{name}.npz"), **arrays)
        meta[name] = dict(wseed=spec["wseed"], img=spec["img"], cfg=spec["cfg"], align=spec["align"], store=spec["store"],
                          counts=counts)
        print(name, counts, {k: v.shape for k, v in arrays.items() if k.startswith(("logits", "dense"))})
    meta["_torch_version"] = torch.__version__
    with open(os.path.join(HERE, "spp_golden_meta.json"), "w") as f:
        json.dump(meta, f, indent=1)


if __name__ == "__main__":
    main()
