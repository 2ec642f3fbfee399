"""Tolerant reader for the reference's Lightning checkpoint (``GATsSPG.ckpt``) and a minimal stand-in
for ``LitModelGATsSPG`` so that ``inference.py:49-58``'s pattern
(``load_from_checkpoint(...).cuda().eval().freeze()`` then ``model(inp_data)``) works without
pytorch_lightning / hydra / omegaconf installed.

Checkpoint layout (src/models/GATsSPG_lightning_model.py:15-37): a pickled dict with
``state_dict`` (keys ``matcher.<123 tensors>`` and ``extractor.<SuperPoint>``) and ``hyper_parameters``
(the constructor kwargs of configs/experiment/train_GATsSPG.yaml:33-63, possibly wrapped in omegaconf /
Lightning container classes that are not importable here).
"""
from __future__ import annotations

import pickle
from collections.abc import Mapping, Sequence

import torch
import torch.nn as nn

from .gats_superglue import GATsSuperGlue

DEFAULT_HPARAMS = {  # configs/experiment/train_GATsSPG.yaml:44-60
    "descriptor_dim": 256, "keypoints_encoder": [32, 64, 128], "match_type": "softmax", "scale_factor": 0.07,
    "match_threshold": 0.2, "include_self": True, "additional": False, "with_linear_transform": False,
}


class _Stub(dict):
    """Stands in for any class the unpickler cannot import (omegaconf.DictConfig, AttributeDict, ...)."""

    def __init__(self, *a, **k):
        super().__init__()

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.update(state)
        else:
            self["_state"] = state

    def __reduce_ex__(self, protocol):  # pragma: no cover
        return (dict, (), dict(self))


class _TolerantUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        try:
            return super().find_class(module, name)
        except Exception:  # noqa: BLE001 -- missing package: substitute a dict-like stub
            return type(name, (_Stub,), {"__module__": module})


class _TolerantPickle:
    """``pickle_module`` for torch.load: stock pickle with the tolerant Unpickler."""
    __name__ = "onepose_amd_tolerant_pickle"
    Unpickler = _TolerantUnpickler
    load = staticmethod(pickle.load)
    loads = staticmethod(pickle.loads)
    dump = staticmethod(pickle.dump)
    dumps = staticmethod(pickle.dumps)
    HIGHEST_PROTOCOL = pickle.HIGHEST_PROTOCOL


def _plain(x):
    """omegaconf / stub containers -> plain python."""
    if isinstance(x, _Stub):
        if "_val" in x and "_content" not in x:      # omegaconf value node (AnyNode / BooleanNode / FloatNode ...)
            return _plain(x["_val"])
        inner = x.get("_content", x.get("_state", x))
        if inner is not x and isinstance(inner, dict):
            return _plain(inner if isinstance(inner, _Stub) else dict(inner))
        if isinstance(inner, (list, tuple)):         # omegaconf ListConfig
            return [_plain(v) for v in inner]
        return {k: _plain(v) for k, v in x.items()}
    if isinstance(x, Mapping):        # dict, and real omegaconf.DictConfig / Lightning AttributeDict when those are installed
        return {k: _plain(v) for k, v in x.items()}
    if isinstance(x, Sequence) and not isinstance(x, (str, bytes)):   # list / tuple / omegaconf.ListConfig
        return [_plain(v) for v in x]
    return x


def read_checkpoint(path, map_location="cpu", overrides=None):
    """-> (matcher_state_dict, hparams).  ``matcher.`` prefix stripped; hparams fall back to the shipped
    config for any key the (possibly stubbed) ``hyper_parameters`` entry does not yield.  Keys of ``overrides`` are
    taken from there and NOT decoded from the file (the way out when a stored value cannot be decoded)."""
    overrides = dict(overrides or {})
    try:
        ckpt = torch.load(path, map_location=map_location, weights_only=False)
    except Exception:  # noqa: BLE001 -- classes of missing packages inside the pickle
        ckpt = torch.load(path, map_location=map_location, weights_only=False, pickle_module=_TolerantPickle)
    sd = ckpt["state_dict"] if "state_dict" in ckpt else ckpt
    matcher = {k[len("matcher."):]: v for k, v in sd.items() if k.startswith("matcher.")}
    if not matcher:  # a bare matcher state dict
        matcher = {k: v for k, v in sd.items() if not k.startswith("extractor.")}
    hp = dict(DEFAULT_HPARAMS)
    raw = _plain(ckpt.get("hyper_parameters", {})) if isinstance(ckpt, dict) else {}
    if isinstance(raw, dict):
        for k in DEFAULT_HPARAMS:
            if k not in raw or k in overrides:
                continue
            if isinstance(raw[k], dict):   # a container that could not be decoded: never fall back silently
                raise ValueError(f"checkpoint hyper-parameter '{k}' could not be decoded ({type(raw[k]).__name__} with keys "
                                 f"{sorted(raw[k])[:6]}); pass it explicitly: load_from_checkpoint(path, {k}=...)")
            hp[k] = raw[k]
    hp.update({k: v for k, v in overrides.items() if k in DEFAULT_HPARAMS})
    return matcher, hp


class LitModelGATsSPG(nn.Module):
    """What inference.py needs from the Lightning wrapper: ``.matcher``, ``load_from_checkpoint``,
    ``forward(x) -> self.matcher(x)`` (GATsSPG_lightning_model.py:36-37), ``.freeze()``.  The SuperPoint
    extractor and the training/validation logic of the reference wrapper are out of scope."""

    def __init__(self, **hparams):
        super().__init__()
        hp = dict(DEFAULT_HPARAMS)
        hp.update({k: v for k, v in hparams.items() if k in DEFAULT_HPARAMS})
        self.hparams = hp
        self.matcher = GATsSuperGlue(hparams=hp)

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location="cpu", **overrides):
        sd, hp = read_checkpoint(checkpoint_path, map_location, overrides)
        model = cls(**hp)
        model.matcher.load_state_dict(sd, strict=True)
        return model

    def forward(self, x):
        return self.matcher(x)

    def freeze(self):
        for p in self.parameters():
            p.requires_grad = False
        return self.eval()
