"""The reference's evaluation loop (inference.py:16-182, configs/experiment/test_GATsSPG.yaml) on the HIP path.

    python -m onepose_amd.inference_runner [--data-dir data] [--objects 0408-colorbox-box:colorbox-4 ...]
                                           [--precision fp32|bf16x3|bf16x6|fp16x3|fp16x4] [--max-frames N]

Per sequence: load the object's annotation (anno_3d_average.npz / anno_3d_collect.npz / idxs.npy under
``<sfm_model_dir>/outputs_superpoint_superglue/anno``), keep the 3D database resident on the GPU, then for every cropped
query image ``color/*.png``: SuperPoint -> GATsSPG -> RANSAC-EPnP -> cm-degree bookkeeping against ``poses_ba/*.txt`` with
the crop intrinsics ``intrin_ba/*.txt`` (GT_box mode, path_utils.py:22-53).  Everything between the image upload and the
pose leaves HBM only for the final 3x4 matrix (onepose_amd.FrameMatcher).

The checkpoints and the dataset are not part of this repository: when ``<data-dir>/models/checkpoints/onepose/GATsSPG.ckpt``,
``<data-dir>/models/extractors/SuperPoint/superpoint_v1.pth`` or the sequences are missing the runner says what it looked
for and exits with code 0 (BASELINE configs[3] becomes runnable the moment the files appear).  Images are read with PIL and
converted to luma with OpenCV's fixed-point BGR2GRAY formula (the reference uses cv2.imread(..., IMREAD_GRAYSCALE),
normalized_dataset.py:24-34), so colour crops give the same uint8 image.

Determinism: the reference seeds every generator with 12345 when inference.py is imported (``seed_everything(12345)``,
inference.py:13) and draws the leaves of all sequences from that one numpy stream (data_utils.py:163-205).  ``main`` does the
same -- ``--seed 12345`` (the default) reproduces the reference's leaf selection; the stream continues across sequences.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import os.path as osp
import sys
import time

import numpy as np
import torch

SPP_CONF = {"descriptor_dim": 256, "nms_radius": 3, "max_keypoints": 4096, "keypoints_threshold": 0.6}   # src/sfm/extract_features.py:19-26
DEFAULT_OBJECTS = ("0408-colorbox-box:colorbox-4", "0409-aptamil-box:aptamil-3", "0419-cookies2-others:cookies2-4")


def default_paths(data_dir):
    """configs/experiment/test_GATsSPG.yaml:11-24."""
    return {"onepose_model_path": osp.join(data_dir, "models", "checkpoints", "onepose", "GATsSPG.ckpt"),
            "extractor_model_path": osp.join(data_dir, "models", "extractors", "SuperPoint", "superpoint_v1.pth"),
            "scan_data_dir": osp.join(data_dir, "onepose_datasets", "test_data"),
            "sfm_model_dir": osp.join(data_dir, "sfm_model")}


def sequence_paths(seq_dir, sfm_model_dir):
    """inference.py:17-46 (GT_box mode)."""
    anno_dir = osp.join(sfm_model_dir, "outputs_superpoint_superglue", "anno")
    return {"img_lists": sorted(glob.glob(osp.join(seq_dir, "color", "*.png"))),
            "avg_anno_3d_path": osp.join(anno_dir, "anno_3d_average.npz"),
            "clt_anno_3d_path": osp.join(anno_dir, "anno_3d_collect.npz"), "idxs_path": osp.join(anno_dir, "idxs.npy")}


def read_image(path):
    """NormalizedDataset.__getitem__ (normalized_dataset.py:21-41): grayscale uint8 / 255 as [1,1,H,W] float32."""
    from PIL import Image
    im = Image.open(path)
    if im.mode in ("L", "I;16", "1"):
        g = np.asarray(im.convert("L"), dtype=np.uint8)
    else:
        # OpenCV's 8-bit BGR2GRAY (what IMREAD_GRAYSCALE applies to a colour PNG): fixed point, 14 fractional bits.
        # PIL's convert("L") rounds differently (16-bit coefficients), which moves some pixels by one LSB.
        rgb = np.asarray(im.convert("RGB"), dtype=np.int64)
        g = ((rgb[..., 0] * 4899 + rgb[..., 1] * 9617 + rgb[..., 2] * 1868 + 8192) >> 14).astype(np.uint8)
    return torch.from_numpy(g.astype(np.float32) / 255.0)[None, None]


def load_models(paths, precision="fp32", device="cuda", extractor_precision="fp32"):
    """inference.py:49-77: the Lightning checkpoint's matcher + the SuperPoint weights (strict loads)."""
    from . import SuperPoint
    from .checkpoint import LitModelGATsSPG
    matcher = LitModelGATsSPG.load_from_checkpoint(paths["onepose_model_path"]).freeze().matcher
    matcher.precision = precision
    extractor = SuperPoint({k: v for k, v in SPP_CONF.items() if k != "keypoints_threshold"}, precision=extractor_precision).eval()
    sd = torch.load(paths["extractor_model_path"], map_location="cpu")
    if isinstance(sd, dict):     # model_io.load_network unwraps 'net' (src/utils/model_io.py); Lightning files use 'state_dict'
        sd = sd.get("net", sd.get("state_dict", sd))
    extractor.load_state_dict(sd, strict=True)
    return matcher.to(device).eval(), extractor.to(device).eval()


@torch.no_grad()
def inference_core(matcher, extractor, seq_dir, sfm_model_dir, num_leaf=8, max_frames=None, device="cuda", log=print):
    """inference.py:96-171 for one sequence -> the evaluator's summary dict (+ timing)."""
    from . import FrameMatcher
    from .database_io import load_object_database
    from .pnp import Evaluator
    p = sequence_paths(seq_dir, sfm_model_dir)
    missing = [k for k in ("avg_anno_3d_path", "clt_anno_3d_path", "idxs_path") if not osp.exists(p[k])]
    if missing or not p["img_lists"]:
        log(f"skip {seq_dir}: missing {[p[k] for k in missing] or 'color/*.png'}")
        return None
    db = load_object_database(p["avg_anno_3d_path"], p["clt_anno_3d_path"], p["idxs_path"], num_leaf=num_leaf, device=device)
    frames = FrameMatcher(extractor, matcher, db)
    evaluator = Evaluator()
    imgs = p["img_lists"][:max_frames] if max_frames else p["img_lists"]
    t0 = time.perf_counter()
    for img_path in imgs:
        K_crop = np.loadtxt(img_path.replace("/color/", "/intrin_ba/").replace(".png", ".txt"))      # path_utils.py:39-43
        pose_gt = np.loadtxt(img_path.replace("/color/", "/poses_ba/").replace(".png", ".txt"))      # path_utils.py:22-26
        pose_pred, _, _ = frames.solve_pose(read_image(img_path).to(device), K_crop, scale=1000)     # inference.py:140-155
        evaluator.evaluate(pose_pred, pose_gt)
    torch.cuda.synchronize()
    res = {k: float(v) for k, v in evaluator.summarize().items()}
    res.update(frames=len(imgs), seconds=round(time.perf_counter() - t0, 3), points_3d=int(db["keypoints3d"].shape[1]))
    return res


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--data-dir", default="data")
    ap.add_argument("--objects", nargs="*", default=list(DEFAULT_OBJECTS), help="<object dir>:<sequence> pairs (test_GATsSPG.yaml input.data_dirs)")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16x3", "bf16x6", "fp16x3", "fp16x4"])
    ap.add_argument("--extractor-precision", default="fp32", choices=["fp32", "fp16x4"], help="arithmetic of the SuperPoint GEMM convolutions")
    ap.add_argument("--num-leaf", type=int, default=8)
    ap.add_argument("--max-frames", type=int, default=None)
    ap.add_argument("--seed", type=int, default=12345,
                    help="numpy global seed set once before the first sequence (12345 = the reference's seed_everything, inference.py:13)")
    a = ap.parse_args(argv)
    np.random.seed(a.seed)      # one stream for all sequences, like the reference (load_object_database draws from it)
    paths = default_paths(a.data_dir)
    need = [paths["onepose_model_path"], paths["extractor_model_path"], paths["scan_data_dir"], paths["sfm_model_dir"]]
    missing = [q for q in need if not osp.exists(q)]
    if missing:
        print("onepose_amd.inference_runner: nothing to evaluate -- not found:\n  " + "\n  ".join(missing) +
              "\n(the OnePose checkpoints / test_data are not shipped with this repository; place them as in the reference's "
              "README and re-run: cm-degree accuracy per sequence is printed as JSON lines)")
        return 0
    if not torch.cuda.is_available():
        print("onepose_amd.inference_runner needs a ROCm GPU (the HIP path has no CPU fallback)", file=sys.stderr)
        return 2
    matcher, extractor = load_models(paths, a.precision, extractor_precision=a.extractor_precision)
    for item in a.objects:
        obj, seq = item.split(":")
        res = inference_core(matcher, extractor, osp.join(paths["scan_data_dir"], obj, seq), osp.join(paths["sfm_model_dir"], obj),
                             a.num_leaf, a.max_frames)
        if res is not None:
            print(json.dumps({"object": obj, "sequence": seq, "precision": a.precision, "extractor_precision": a.extractor_precision, **res}),
                  flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
