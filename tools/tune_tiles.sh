#!/bin/bash
# A/B runs of tuning knobs on the GPU box (one process per variant; same inputs).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/tune_tiles.log
: > $out
run() { echo "== K=$K $*" >> $out; env "$@" python bench.py --steps 100 --warmup 10 --no-cpu-baseline --kernel "$K" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'],'fps', d['ms_per_step'],'ms', d['roofline']['kernel'], d['roofline']['kernel_ms'],'ms', d['roofline']['achieved'],'TF')" >> $out; }
K=mlp3 run GATSSPG_MLP3_TILE=0
K=mlp3 run GATSSPG_MLP3_TILE=13
K=final_proj_norm run GATSSPG_MLP3_TILE=0
cat $out
