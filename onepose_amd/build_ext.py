"""Build libgatsspg_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m onepose_amd.build_ext [--force] [--remarks] [--profiling]
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libgatsspg_hip.so")
SOURCES = ["gatsspg_gemm_kernels.hip", "gatsspg_stream_kernels.hip", "gatsspg_capi.hip"]
HEADERS = ["gatsspg_common.h", "gatsspg_launch.h", "gemm_f32_mfma.h", os.path.join("..", "..", "include", "gatsspg.h")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, remarks=False, verbose=True, profiling=False):
    """Compile every HIP source for gfx950 into onepose_amd/lib/libgatsspg_hip.so.
    profiling=True adds -DGATSSPG_PROFILING_BUILD (timing-only ablation variants + the mlp0 timeline hook used by
    tools/trace_mlp0.py); never ship that build."""
    if not force and not profiling and not is_stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", LIB_PATH]
    if remarks:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
    if profiling:
        cmd.append("-DGATSSPG_PROFILING_BUILD")
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv, remarks="--remarks" in sys.argv, profiling="--profiling" in sys.argv)
