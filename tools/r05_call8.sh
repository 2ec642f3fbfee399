#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05i; mkdir -p $O
cd $R
for n in 0 3 0 3; do
  timeout 120 python tools/trace_mlp0.py --inflight $n 2>&1 | grep -E "MLP0_TILE|shader clock|mainloop|epilogue|block end|dur on CUs" >> $O/trace_mlp0_inflight.txt
  echo "--" >> $O/trace_mlp0_inflight.txt
done
cat $O/trace_mlp0_inflight.txt
