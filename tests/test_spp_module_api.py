"""CPU-side checks of the SuperPoint drop-in: parameter layout of the reference module (superpoint.py:119-133),
constructor / config behaviour (:104-110,135-137), C-ABI library loads and exports what include/superpoint.h declares.
No compute calls (no GPU here)."""
import os
import re
import shutil

import numpy as np
import pytest
import torch

from conftest import ROOT
from onepose_amd import SuperPoint, synthetic

REF_SHAPES = {  # reference state_dict, forward order
    "conv1a": (64, 1, 3, 3), "conv1b": (64, 64, 3, 3), "conv2a": (64, 64, 3, 3), "conv2b": (64, 64, 3, 3),
    "conv3a": (128, 64, 3, 3), "conv3b": (128, 128, 3, 3), "conv4a": (128, 128, 3, 3), "conv4b": (128, 128, 3, 3),
    "convPa": (256, 128, 3, 3), "convPb": (65, 256, 1, 1), "convDa": (256, 128, 3, 3), "convDb": (256, 256, 1, 1),
}


def test_state_dict_layout_matches_reference():
    m = SuperPoint({})
    sd = m.state_dict()
    assert list(sd) == [f"{n}.{p}" for n in REF_SHAPES for p in ("weight", "bias")]
    for n, shp in REF_SHAPES.items():
        assert tuple(sd[f"{n}.weight"].shape) == shp and tuple(sd[f"{n}.bias"].shape) == (shp[0],)
    syn = synthetic.make_spp_state_dict(0)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in syn.items()}, strict=True)
    np.testing.assert_array_equal(m.convPb.weight.detach().numpy(), syn["convPb.weight"])


def test_config_defaults_and_validation():
    m = SuperPoint({"nms_radius": 3, "max_keypoints": 4096, "keypoints_threshold": 0.6})   # the pipeline's dict, typo included
    assert m.config["keypoint_threshold"] == 0.005 and m.config["remove_borders"] == 4 and m.config["nms_radius"] == 3
    for bad in (0, -2):
        with pytest.raises(ValueError, match="max_keypoints"):
            SuperPoint({"max_keypoints": bad})
    with pytest.raises(ValueError, match="descriptor_dim"):
        SuperPoint({"descriptor_dim": 128})


def test_cpu_tensors_are_refused():
    m = SuperPoint({})
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 1, 64, 64))
    with pytest.raises(ValueError, match="grayscale"):
        m(torch.zeros(1, 3, 64, 64))


hipcc = pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")


@hipcc
def test_every_declared_symbol_is_exported():
    from onepose_amd import _native_spp, build_ext
    build_ext.build(verbose=False)
    lib = _native_spp.load()
    text = open(os.path.join(ROOT, "include", "superpoint.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(spp_[a-z0-9_]+)\s*\(", text)))
    assert len(names) == 9
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/superpoint.h but not exported"
    assert set(names) == set(_native_spp.SYMBOLS)
    assert lib.spp_version() >= 1
    nw = 3 * 64 * 576 + 128 * 576 + 3 * 128 * 1152 + 512 * 1152 + 128 * 256 + 256 * 256           # GEMM-convolution weights
    assert lib.spp_packed_weights_bytes() == (4 * (64 * 9 + 64 + 64 + nw + 3 * 64 + 4 * 128 + 512 + 128 + 256)   # fp32 blob
                                              + 2 * 2 * nw)                                                       # + fp16 hi / lo planes
    assert 0 < lib.spp_workspace_bytes(1, 64, 64) < lib.spp_workspace_bytes(1, 512, 512) < 2**28
    assert lib.spp_workspace_bytes(1, 7, 64) == 0 and b">= 8" in lib.spp_last_error()
    assert lib.spp_workspace_bytes(1, 75, 101) > 0        # any size >= 8: floor-mode poolings like the reference


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from onepose_amd import _native_spp
    from onepose_amd._native import NativeError
    monkeypatch.setattr(_native_spp, "_lib", None)
    monkeypatch.setattr(_native_spp, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(NativeError, match="no CPU / PyTorch fallback"):
        _native_spp.load()
