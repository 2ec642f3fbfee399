#!/bin/bash
# A/B of tuning-build knobs at custom shapes: tools/shape_ab.sh <tag> "<shape> <shape> ..." "<KNOBS>" "<KNOBS>" ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$1; mkdir -p $O
SHAPES=$2; shift 2
cd $R
for sh in $SHAPES; do
for knobs in "$@"; do
  envs=$(echo "$knobs" | tr ',' ' ' | sed 's/\([A-Z0-9_]*=\)/GATSSPG_\1/g')
  tag=${sh}_$(echo "${knobs:-default}" | tr ',=' '__')
  env $envs python bench.py --tuning-lib --shape $sh --steps 60 --warmup 10 --reps 3 --no-cpu-baseline > $O/sab_$tag.json 2>$O/sab_$tag.err
  python -c "
import json,sys
d=json.load(open('$O/sab_$tag.json')); print('$sh [$knobs] mlp0_ms', d['roofline']['kernel_ms'], 'latency_ms', d['config']['single_frame_latency_ms'], 'inflight', d['value'])" || tail -3 $O/sab_$tag.err
done; done
