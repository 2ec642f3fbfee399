"""Lightning-layout checkpoint reading without pytorch_lightning / omegaconf (CPU)."""
import sys
import types

import numpy as np
import torch

from onepose_amd import synthetic
from onepose_amd.checkpoint import LitModelGATsSPG, read_checkpoint


def _fake_ckpt(tmp_path, with_foreign_classes):
    sd = {"matcher." + k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(3).items()}
    sd["extractor.conv1a.weight"] = torch.zeros(64, 1, 3, 3)  # SuperPoint tensors live in the same file
    hp = {"descriptor_dim": 256, "keypoints_encoder": [32, 64, 128], "match_type": "softmax", "scale_factor": 0.07,
          "match_threshold": 0.35, "include_self": True, "additional": False, "with_linear_transform": False,
          "optimizer": "adam", "lr": 1e-3}
    path = tmp_path / "GATsSPG.ckpt"
    if with_foreign_classes:
        # simulate hyper_parameters pickled as a class from a package that is not installed here
        mod = types.ModuleType("omegaconf_like_pkg")
        DictConfig = type("DictConfig", (dict,), {"__module__": "omegaconf_like_pkg", "__qualname__": "DictConfig"})
        mod.DictConfig = DictConfig
        sys.modules["omegaconf_like_pkg"] = mod
        try:
            torch.save({"state_dict": sd, "hyper_parameters": DictConfig(hp), "epoch": 9}, path)
        finally:
            del sys.modules["omegaconf_like_pkg"]
    else:
        torch.save({"state_dict": sd, "hyper_parameters": hp, "epoch": 9}, path)
    return path


def test_read_plain_lightning_checkpoint(tmp_path):
    sd, hp = read_checkpoint(_fake_ckpt(tmp_path, False))
    assert len(sd) == 123 and all(not k.startswith(("matcher.", "extractor.")) for k in sd)
    assert hp["match_threshold"] == 0.35 and "lr" not in hp
    model = LitModelGATsSPG.load_from_checkpoint(_fake_ckpt(tmp_path, False)).freeze()
    assert not any(p.requires_grad for p in model.parameters()) and not model.training
    ref = synthetic.make_state_dict(3)
    np.testing.assert_array_equal(model.matcher.state_dict()["gnn.layers.4.mlp.0.weight"].numpy(), ref["gnn.layers.4.mlp.0.weight"])
    assert model.matcher.hparams["match_threshold"] == 0.35


def test_read_checkpoint_with_unimportable_hparam_classes(tmp_path):
    sd, hp = read_checkpoint(_fake_ckpt(tmp_path, True))
    assert len(sd) == 123
    assert hp["match_threshold"] == 0.35 and hp["scale_factor"] == 0.07


def _omegaconf_like_ckpt(tmp_path, hp):
    """hyper_parameters pickled the way omegaconf does it: a DictConfig whose ``_content`` maps each key to a VALUE NODE
    object carrying the value under ``_val`` (both classes unimportable here)."""
    mod = types.ModuleType("omegaconf_like_pkg2")
    for cname in ("DictConfig", "AnyNode", "ListConfig"):
        setattr(mod, cname, type(cname, (object,), {"__module__": "omegaconf_like_pkg2", "__qualname__": cname}))
    sys.modules["omegaconf_like_pkg2"] = mod
    try:
        def node(v):
            if isinstance(v, list):
                n = mod.ListConfig()
                n.__dict__["_content"] = [node(e) for e in v]
                return n
            n = mod.AnyNode()
            n.__dict__["_val"] = v
            return n
        cfg = mod.DictConfig()
        cfg.__dict__["_content"] = {k: node(v) for k, v in hp.items()}
        sd = {"matcher." + k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(3).items()}
        path = tmp_path / "GATsSPG_omegaconf.ckpt"
        torch.save({"state_dict": sd, "hyper_parameters": cfg}, path)
    finally:
        del sys.modules["omegaconf_like_pkg2"]
    return path


def test_omegaconf_value_nodes_are_unwrapped_not_defaulted(tmp_path):
    """Non-default flags wrapped in omegaconf value nodes must reach the module (they used to fall back to the defaults)."""
    hp = {"descriptor_dim": 256, "keypoints_encoder": [32, 64, 128], "match_type": "softmax", "scale_factor": 0.05,
          "match_threshold": 0.3, "include_self": False, "additional": True, "with_linear_transform": True}
    sd, got = read_checkpoint(_omegaconf_like_ckpt(tmp_path, hp))
    assert len(sd) == 123
    assert got == hp
