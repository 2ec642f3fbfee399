#!/bin/bash
# round 5, call 2: the transposed mlp.0 epilogue (U^T + in-lane statistics) and the transposed B operand of mlp.3, fp16 modes: parity, then A/B
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05b; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -p no:cacheprovider -x -k "attention_layer_stage or trained or scale_invariant or message_operator or database_cache or saturate or random_shapes or benchmarked_shapes or small" 2>&1 | tail -25 > $O/pytest_subset.log
tail -4 $O/pytest_subset.log
for k in mlp0 mlp3; do
  timeout 300 python tools/ab_live.py --config fp16x4 --kernel $k --rounds 6 --steps 30 "" SP_UT=0 > $O/ab_ut_fp16x4_$k.txt 2>&1
  tail -4 $O/ab_ut_fp16x4_$k.txt
done
timeout 300 python tools/ab_live.py --config fp16x4-b8 --kernel mlp0 --rounds 4 --steps 10 "" SP_UT=0 > $O/ab_ut_fp16x4_b8.txt 2>&1; tail -3 $O/ab_ut_fp16x4_b8.txt
timeout 300 python tools/ab_live.py --config fp16x4-real --kernel mlp0 --rounds 6 --steps 40 "" SP_UT=0 > $O/ab_ut_fp16x4_real.txt 2>&1; tail -3 $O/ab_ut_fp16x4_real.txt
timeout 300 python tools/ab_live.py --config fp16x3 --kernel mlp0 --rounds 4 --steps 30 "" SP_UT=0 > $O/ab_ut_fp16x3.txt 2>&1; tail -3 $O/ab_ut_fp16x3.txt
