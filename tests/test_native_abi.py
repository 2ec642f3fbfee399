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
    assert lib.gatsspg_version() >= 411
    big = 768 * 256 + 512 * 512 + 256 * 512           # the three big operators of an attention layer
    assert lib.gatsspg_packed_weights_bytes() == (4 * (8 * (big + 768 + 512 + 256 + 8) + 4 * (512 + 256 * 256) + 256 * 256 + 256)   # + 8: fp16 plane scales (4) and the per-head row-L1 norms of the message half (4) per layer
                                                 
                                                  + 2 * 5 * 8 * big)   # + their three bf16 planes (hi / lo / lo2) and two fp16 planes
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


ASAN_SCRIPT = r'''
import ctypes, sys
from ctypes import c_char_p, c_float, c_int, c_size_t, c_void_p
lib = ctypes.CDLL(sys.argv[1])
lib.gatsspg_last_error.restype = c_char_p
lib.gatsspg_workspace_bytes.restype = c_size_t
lib.gatsspg_db_cache_bytes.restype = c_size_t
lib.gatsspg_packed_weights_bytes.restype = c_size_t
lib.gatsspg_kenc_scratch_bytes.restype = c_size_t
assert lib.gatsspg_version() >= 411 and lib.gatsspg_packed_weights_bytes() > 0
for args in ((1, 1000, 7000, 8), (8, 1000, 7000, 8), (1, 2, 2, 1), (1, 1, 7000, 8), (0, 10, 10, 8), (1, 10, 10, 65), (4096, 100000, 100000, 8)):
    lib.gatsspg_workspace_bytes(*args)
    lib.gatsspg_last_error()
assert lib.gatsspg_db_cache_bytes(1, 7000) > 0 and lib.gatsspg_db_cache_bytes(0, 1) == 0
buf = (ctypes.c_char * 4096)()
p = ctypes.cast(buf, c_void_p)
fwd = [p, p, p, p, 1, 1000, 7000, 8, 1, c_float(0.07), c_float(0.2), p, p, p, p, p, p, c_size_t(4096), None]
assert lib.gatsspg_forward(*fwd) != 0 and b"workspace too small" in lib.gatsspg_last_error()      # 4 KB workspace for 1000/7000
bad = list(fwd); bad[8] = 0x40
assert lib.gatsspg_forward(*bad) != 0 and b"unknown bits" in lib.gatsspg_last_error()
bad = list(fwd); bad[8] = 0x300
assert lib.gatsspg_forward(*bad) != 0 and b"exclusive" in lib.gatsspg_last_error()
bad = list(fwd); bad[16] = None
assert lib.gatsspg_forward(*bad) != 0 and b"null" in lib.gatsspg_last_error()
bad = list(fwd); bad[5] = 1
assert lib.gatsspg_forward(*bad) != 0
assert lib.gatsspg_pack_weights(None, p, None) != 0
assert lib.gatsspg_attn_layer(p, 99, 0, 1, 100, 200, 8, 0, p, c_size_t(4096), None) != 0
assert lib.gatsspg_forward_profiled(*fwd, 99, 0, p, p) != 0 and b"kernel_id" in lib.gatsspg_last_error()
assert lib.gatsspg_prepare_database(p, p, p, 1, 7000, 8, 0, None, c_size_t(0), p, c_size_t(4096), None) != 0
assert lib.gatsspg_keypoint_encoder(None, p, p, 1, 10, p, p, c_size_t(0), None) != 0
print("ASAN-OK")
'''


def test_host_side_under_address_sanitizer(tmp_path):
    """SURVEY.md section 5 'sanitizers': the host side of the C ABI (argument checks, workspace carve-up, error strings -- every
    path that returns before a kernel is enqueued, so no GPU is needed) built with -fsanitize=address and driven through its
    refusals; any heap / stack / global overflow or use-after-scope aborts the child process."""
    import subprocess
    import sys
    from onepose_amd import build_ext
    rt = build_ext.asan_runtime()
    if rt is None:
        pytest.skip("clang AddressSanitizer runtime not found")
    so = build_ext.build_asan(str(tmp_path / "libgatsspg_asan.so"))
    script = tmp_path / "drive.py"
    script.write_text(ASAN_SCRIPT)
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:verify_asan_link_order=0")
    r = subprocess.run([sys.executable, str(script), so], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "ASAN-OK" in r.stdout and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
