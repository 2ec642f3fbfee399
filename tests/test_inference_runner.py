"""The inference.py-shaped runner: skip message when the OnePose files are absent (CPU), and an end-to-end run over a
synthetic dataset tree laid out like the reference's (GPU; random weights, so the cm-degree numbers are meaningless --
the plumbing from color/*.png to the evaluator summary is what is tested)."""
import json
import os

import numpy as np
import pytest
import torch

from onepose_amd import inference_runner as ir
from onepose_amd import synthetic


def _make_tree(root, n_frames=3, n3d=300):
    """<root>/models/..., <root>/onepose_datasets/test_data/<obj>/<seq>/{color,intrin_ba,poses_ba}, <root>/sfm_model/<obj>/.../anno"""
    from PIL import Image
    paths = ir.default_paths(str(root))
    os.makedirs(os.path.dirname(paths["onepose_model_path"]))
    os.makedirs(os.path.dirname(paths["extractor_model_path"]))
    hp = {"descriptor_dim": 256, "keypoints_encoder": [32, 64, 128], "match_type": "softmax", "scale_factor": 0.07,
          "match_threshold": 0.0, "include_self": True, "additional": False, "with_linear_transform": False}
    torch.save({"state_dict": {"matcher." + k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(0).items()},
                "hyper_parameters": hp}, paths["onepose_model_path"])
    torch.save({k: torch.from_numpy(v) for k, v in synthetic.make_spp_state_dict(0).items()}, paths["extractor_model_path"])
    obj, seq = "0001-synthetic-box", "synthetic-1"
    seq_dir = os.path.join(paths["scan_data_dir"], obj, seq)
    for d in ("color", "intrin_ba", "poses_ba"):
        os.makedirs(os.path.join(seq_dir, d))
    prob = synthetic.make_pnp_problem(50, 0.0, 0.0, 1)
    for i in range(n_frames):
        img = (synthetic.make_image(1, 512, 512, 20 + i)[0, 0] * 255).astype(np.uint8)
        Image.fromarray(img).save(os.path.join(seq_dir, "color", f"{i}.png"))
        np.savetxt(os.path.join(seq_dir, "intrin_ba", f"{i}.txt"), prob["K"])
        np.savetxt(os.path.join(seq_dir, "poses_ba", f"{i}.txt"), np.concatenate([prob["pose_gt"], [[0, 0, 0, 1]]]))
    anno_dir = os.path.join(paths["sfm_model_dir"], obj, "outputs_superpoint_superglue", "anno")
    os.makedirs(anno_dir)
    synthetic.write_annotation(anno_dir, synthetic.make_annotation(n=n3d, dim=256, seed=2))
    return f"{obj}:{seq}"


def test_missing_files_give_a_skip_message_not_an_error(tmp_path, capsys):
    assert ir.main(["--data-dir", str(tmp_path)]) == 0
    out = capsys.readouterr().out
    assert "nothing to evaluate" in out and "GATsSPG.ckpt" in out and "superpoint_v1.pth" in out


def test_path_conventions_match_the_reference(tmp_path):
    p = ir.default_paths("data")
    assert p["onepose_model_path"] == "data/models/checkpoints/onepose/GATsSPG.ckpt"          # test_GATsSPG.yaml:12
    assert p["extractor_model_path"] == "data/models/extractors/SuperPoint/superpoint_v1.pth"  # :13
    s = ir.sequence_paths("/d/obj/seq", "/m/obj")
    assert s["avg_anno_3d_path"] == "/m/obj/outputs_superpoint_superglue/anno/anno_3d_average.npz"   # inference.py:17-20
    img = torch.rand(4, 6)
    from PIL import Image
    Image.fromarray((img.numpy() * 255).astype(np.uint8)).save(tmp_path / "x.png")
    t = ir.read_image(str(tmp_path / "x.png"))
    assert t.shape == (1, 1, 4, 6) and t.dtype == torch.float32 and float(t.max()) <= 1.0


@pytest.mark.gpu
def test_runner_end_to_end_on_a_synthetic_tree(tmp_path, capsys):
    item = _make_tree(tmp_path)
    assert ir.main(["--data-dir", str(tmp_path), "--objects", item, "--max-frames", "3"]) == 0
    lines = [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["frames"] == 3 and d["points_3d"] == 300 and {"cmd1", "cmd3", "cmd5"} <= set(d) and 0.0 <= d["cmd5"] <= 1.0


def test_colour_crops_use_opencv_luma_and_main_seeds_the_leaf_stream(tmp_path):
    """cv2.imread(IMREAD_GRAYSCALE) on a colour PNG = fixed-point BGR2GRAY ((R*4899 + G*9617 + B*1868 + 8192) >> 14); PIL's
    convert('L') rounds differently.  Known answers of OpenCV's formula, e.g. pure red 255 -> 76, green -> 150, blue -> 29."""
    from PIL import Image
    rgb = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255]], [[255, 255, 255], [12, 200, 77], [1, 2, 3]]], dtype=np.uint8)
    Image.fromarray(rgb, "RGB").save(tmp_path / "c.png")
    t = ir.read_image(str(tmp_path / "c.png"))
    want = np.array([[76, 150, 29], [255, (12 * 4899 + 200 * 9617 + 77 * 1868 + 8192) >> 14, 2]], dtype=np.float32) / 255.0
    np.testing.assert_array_equal(t[0, 0].numpy(), want)
    # main() seeds numpy's global stream (the reference's seed_everything(12345), inference.py:13) even when it has nothing to do
    np.random.seed(1)
    ir.main(["--data-dir", str(tmp_path)])
    a = np.random.permutation(10)
    np.random.seed(12345)
    np.testing.assert_array_equal(a, np.random.permutation(10))
