#!/bin/bash
# A/B a compile-time flag of the HIP library on the GPU box:  tools/ab_flag.sh -DGATSSPG_NT_B
cd "$(dirname "$0")/.."
for flag in "" "$1"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $flag -o onepose_amd/lib/libgatsspg_hip.so onepose_amd/csrc/gatsspg_gemm_kernels.hip onepose_amd/csrc/gatsspg_stream_kernels.hip onepose_amd/csrc/gatsspg_capi.hip 2>/dev/null
  echo "== flag=[$flag]"
  for k in mlp0 qkv_kv mlp3; do python bench.py --steps 100 --warmup 10 --streams 1 --no-cpu-baseline --kernel $k 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'],'fps', d['roofline']['kernel'], d['roofline']['kernel_ms'],'ms')"; done
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('3 in flight:', d['value'],'fps')"
done
