"""Build libgatsspg_hip.so (matcher), libspp_hip.so (SuperPoint extractor) and libpnp_hip.so (RANSAC-EPnP) in-tree with hipcc for gfx950
(cross-compiles without a GPU).

    python -m onepose_amd.build_ext [--force] [--remarks] [--profiling] [--tuning]

The product libraries never read the environment.  ``--tuning`` builds SEPARATE libraries (``lib*_tuning.so``, never
loaded by the package) with -DGATSSPG_TUNING / -DSPP_TUNING, whose alternative tile shapes are selected by
GATSSPG_<KNOB> / SPP_<KNOB> environment variables (tools/ab_tuning.py); ``--profiling`` adds the timing-only ablation
variants and the mlp0 timeline hook (tools/trace_mlp0.py) to such a separate library as well.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libgatsspg_hip.so")
SOURCES = ["gatsspg_gemm_kernels.hip", "gatsspg_split_kernels.hip", "gatsspg_stream_kernels.hip", "gatsspg_capi.hip"]
HEADERS = ["gatsspg_common.h", "gatsspg_launch.h", "gemm_f32_mfma.h", "gemm_split_glds.h", os.path.join("..", "..", "include", "gatsspg.h")]
SPP_LIB_PATH = os.path.join(LIB_DIR, "libspp_hip.so")
SPP_SOURCES = ["spp_conv_kernels.hip", "spp_detect_kernels.hip", "spp_capi.hip"]
SPP_HEADERS = ["spp_common.h", "gemm_f32_mfma.h", "gatsspg_common.h", os.path.join("..", "..", "include", "superpoint.h")]
PNP_LIB_PATH = os.path.join(LIB_DIR, "libpnp_hip.so")
PNP_SOURCES = ["pnp_kernels.hip"]
PNP_HEADERS = [os.path.join("..", "..", "include", "pnp.h")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _stale(lib, deps):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(d) > t for d in (os.path.join(CSRC, s) for s in deps) if os.path.exists(d))


def is_stale():
    return (_stale(LIB_PATH, SOURCES + HEADERS) or _stale(SPP_LIB_PATH, SPP_SOURCES + SPP_HEADERS)
            or _stale(PNP_LIB_PATH, PNP_SOURCES + PNP_HEADERS))


def source_hash():
    """sha256 (first 16 hex digits) over the names and contents of every file under csrc/ and include/: identifies the build that a
    committed measurement (profiles/pmc_traffic.json) was taken on; bench.py says "stale" when it differs from the sources it runs."""
    import hashlib
    h = hashlib.sha256()
    for d in (CSRC, os.path.join(HERE, "..", "include")):
        for name in sorted(os.listdir(d)):
            if name.endswith((".hip", ".h")):
                h.update(name.encode())
                with open(os.path.join(d, name), "rb") as f:
                    h.update(f.read())
    return h.hexdigest()[:16]


def tuning_path(lib):
    return lib[:-3] + "_tuning.so"


def build(force=False, remarks=False, verbose=True, profiling=False, tuning=False, syntax_only=False):
    """Compile every HIP source for gfx950 into onepose_amd/lib/lib{gatsspg,spp,pnp}_hip.so.
    tuning / profiling builds go to lib*_tuning.so (environment knobs; profiling adds -DGATSSPG_PROFILING_BUILD: timing-only
    ablation variants + the mlp0 timeline hook) -- the package never loads those.  syntax_only: front-end check only."""
    os.makedirs(LIB_DIR, exist_ok=True)
    special = tuning or profiling
    for lib, srcs, deps in ((LIB_PATH, SOURCES, SOURCES + HEADERS), (SPP_LIB_PATH, SPP_SOURCES, SPP_SOURCES + SPP_HEADERS),
                            (PNP_LIB_PATH, PNP_SOURCES, PNP_SOURCES + PNP_HEADERS)):
        if special and lib == PNP_LIB_PATH:
            continue
        out = tuning_path(lib) if special else lib
        if not force and not syntax_only and not _stale(out, deps):
            continue
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
        cmd += ["-fsyntax-only"] if syntax_only else ["-shared", "-o", out]
        if remarks:
            cmd.append("-Rpass-analysis=kernel-resource-usage")
        if special:
            cmd += ["-DGATSSPG_TUNING", "-DSPP_TUNING"]
        if profiling:
            cmd.append("-DGATSSPG_PROFILING_BUILD")
        cmd += [os.path.join(CSRC, s) for s in srcs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)
    return LIB_PATH


def asan_runtime():
    """Path of clang's shared AddressSanitizer runtime (LD_PRELOAD it into the python that loads an ASan build), or None."""
    import glob
    cands = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    return cands[-1] if cands else None


def build_asan(out_path):
    """Debug build of the matcher library with AddressSanitizer on the HOST side (argument checking, workspace carve-up, launch
    plumbing; device code is not instrumented): SURVEY.md section 5 'sanitizers'.  Never loaded by the package -- tests only."""
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-fsanitize=address", "-fno-gpu-sanitize",
           "-shared-libsan", "-o", out_path] + [os.path.join(CSRC, s) for s in SOURCES]
    subprocess.run(cmd, check=True, cwd=CSRC, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out_path


if __name__ == "__main__":
    build(force="--force" in sys.argv, remarks="--remarks" in sys.argv, profiling="--profiling" in sys.argv,
          tuning="--tuning" in sys.argv)
