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


def test_override_replaces_an_undecodable_hyper_parameter(tmp_path):
    """A stored value that cannot be decoded raises -- and the override the message suggests really takes effect."""
    import pytest
    sd = {"matcher." + k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(3).items()}
    hp = {"match_threshold": {"_weird": 1, "_node": 2}, "scale_factor": 0.05}
    path = tmp_path / "odd.ckpt"
    torch.save({"state_dict": sd, "hyper_parameters": hp}, path)
    with pytest.raises(ValueError, match="match_threshold"):
        read_checkpoint(path)
    _, got = read_checkpoint(path, overrides={"match_threshold": 0.4})
    assert got["match_threshold"] == 0.4 and got["scale_factor"] == 0.05
    model = LitModelGATsSPG.load_from_checkpoint(path, match_threshold=0.4)
    assert model.matcher.hparams["match_threshold"] == 0.4


def test_real_mapping_containers_are_decoded(tmp_path):
    """hyper_parameters as a Mapping that is not a dict subclass (what omegaconf.DictConfig is when omegaconf IS installed)."""
    from collections.abc import Mapping
    from onepose_amd.checkpoint import _plain

    class Cfg(Mapping):
        def __init__(self, d):
            self._d = d

        def __getitem__(self, k):
            return self._d[k]

        def __iter__(self):
            return iter(self._d)

        def __len__(self):
            return len(self._d)

    assert _plain(Cfg({"a": Cfg({"b": (1, 2)}), "s": "txt"})) == {"a": {"b": [1, 2]}, "s": "txt"}
