#!/bin/bash
# A/B of tuning-build knobs on one config: tools/ab_knobs.sh <tag> <config> <kernel> "<KNOB=V,KNOB=V>" ...   ("" = defaults)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$1; mkdir -p $O
C=$2; K=$3; shift 3
cd $R
for knobs in "$@"; do
  envs=$(echo "$knobs" | tr ',' ' ' | sed 's/\([A-Z0-9_]*=\)/GATSSPG_\1/g')
  tag=$(echo "${knobs:-default}" | tr ',=' '__')
  env $envs python bench.py --tuning-lib --config $C --kernel $K --steps 60 --warmup 10 --reps 3 --no-side-arithmetics --no-cpu-baseline > $O/ab_$tag.json 2>$O/ab_$tag.err
  python -c "
import json,sys
d=json.load(open('$O/ab_$tag.json')); print('[$knobs] kernel_ms', d['roofline']['kernel_ms'], 'pair', d['roofline']['empty_event_pair_ms'], 'latency_ms', d['config']['single_frame_latency_ms'], 'inflight', d['value'], 'flips', (d.get('parity_check') or {}).get('argmax_flips'))" || tail -3 $O/ab_$tag.err
done
