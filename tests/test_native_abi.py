"""CPU-side checks of the C-ABI library: it builds for gfx950, loads, and exports every symbol that
include/gatsspg.h declares.  No compute calls (no GPU here)."""
import os
import re
import shutil

import pytest

from conftest import ROOT

pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"),
                                reason="hipcc not available")


@pytest.fixture(scope="module")
def lib():
    from onepose_amd import _native, build_ext
    build_ext.build(verbose=False)
    return _native.load()


def header_functions():
    text = open(os.path.join(ROOT, "include", "gatsspg.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gatsspg_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    from onepose_amd import _native
    names = header_functions()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/gatsspg.h but not exported"
    assert set(names) == set(_native.SYMBOLS), "ctypes binding table out of sync with the header"


def test_host_only_entry_points(lib):
    assert lib.gatsspg_version() >= 200
    big = 768 * 256 + 512 * 512 + 256 * 512           # the three big operators of an attention layer
    assert lib.gatsspg_packed_weights_bytes() == (4 * (8 * (big + 768 + 512 + 256) + 4 * (512 + 256 * 256) + 256 * 256 + 256)
                                                  + 2 * 3 * 8 * big)   # + their three bf16 planes (hi / lo / lo2)
    small = lib.gatsspg_workspace_bytes(1, 500, 2000, 8)
    head = lib.gatsspg_workspace_bytes(1, 1000, 7000, 8)
    assert 0 < small < head < 200 * 2**20
    assert lib.gatsspg_workspace_bytes(2, 1000, 7000, 8) > head
    # error paths: the reference returns early for empty sides and raises for single points;
    # the C ABI refuses both and says why
    assert lib.gatsspg_workspace_bytes(1, 1, 7000, 8) == 0
    assert b"n1 and n2" in lib.gatsspg_last_error()
    assert lib.gatsspg_workspace_bytes(0, 10, 10, 8) == 0
    assert lib.gatsspg_workspace_bytes(1, 10, 10, 65) == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from onepose_amd import _native
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_native.NativeError, match="no CPU / PyTorch fallback"):
        _native.load()


def test_product_libraries_never_read_the_environment(lib):
    """Precision and tile shapes are arguments / compile-time constants: the shipped libraries do not even import getenv
    (a stray GATSSPG_* variable in a user's shell cannot change numerics or code paths)."""
    import subprocess
    from onepose_amd import build_ext
    for path in (build_ext.LIB_PATH, build_ext.SPP_LIB_PATH, build_ext.PNP_LIB_PATH):
        syms = subprocess.run(["nm", "-D", "--undefined-only", path], capture_output=True, text=True, check=True).stdout
        assert "getenv" not in syms, f"{os.path.basename(path)} imports getenv"


def test_tuning_and_profiling_builds_compile():
    """The -DGATSSPG_TUNING / -DGATSSPG_PROFILING_BUILD variants (tools/ab_tuning.py, tools/trace_mlp0.py) must keep
    compiling: front-end check of every translation unit with both defines."""
    from onepose_amd import build_ext
    build_ext.build(profiling=True, syntax_only=True, verbose=False)
