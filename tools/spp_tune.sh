#!/bin/bash
# Extractor tile sweep: tools/spp_tune.sh "0,0,0,1,1,1,1,1,1,1" "0,0,0,2,2,2,2,2,1,2" ...
# prints images/s (3 in flight), single-image latency and the event-timed kernel for each SPP_TILES value.
for t in "$@"; do
  SPP_TILES="$t" python bench.py --extractor --steps 60 --warmup 10 --no-cpu-baseline ${SPP_BENCH_ARGS} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$t', d['value'], 'img/s', d['config']['single_image_latency_ms'], 'ms', d['roofline']['kernel'], d['roofline']['kernel_ms'], 'ms', d['roofline']['achieved'], 'TF/s')"
done
