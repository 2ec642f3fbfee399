#!/bin/bash
cd "$(dirname "$0")/.."
for sl in 0 6 10 16; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DGATSSPG_WS_SLEEP=$sl -o onepose_amd/lib/libgatsspg_hip.so onepose_amd/csrc/gatsspg_gemm_kernels.hip onepose_amd/csrc/gatsspg_stream_kernels.hip onepose_amd/csrc/gatsspg_capi.hip 2>/dev/null
  echo "== ws sleep=$sl"
  GATSSPG_MLP0_TILE=3 python bench.py --steps 100 --warmup 10 --streams 1 --no-cpu-baseline --kernel mlp0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'],'fps', d['roofline']['kernel'], d['roofline']['kernel_ms'],'ms')"
done
