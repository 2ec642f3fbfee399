#!/bin/bash
# round 5, call 3: A/B of this round's candidates inside one process each (tools/ab_live.py): transposed mlp.0 epilogue with the rank-1 MFMA fold,
# XCD pairing of the 64-column kernels, register-direct stores of the fp32 kernels
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05c; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -p no:cacheprovider -x -k "attention_layer_stage or trained or database_cache or benchmarked_shapes or small" 2>&1 | tail -6 > $O/pytest_subset.log
tail -3 $O/pytest_subset.log
timeout 300 python tools/ab_live.py --config fp16x4 --kernel mlp0 --rounds 6 --steps 30 "" SP_UT=0 SP_XCD_PAIR=0 SP_UT=0,SP_XCD_PAIR=0 > $O/ab_fp16x4_ut_xcd.txt 2>&1; tail -5 $O/ab_fp16x4_ut_xcd.txt
timeout 300 python tools/ab_live.py --config fp16x4 --kernel qkv_kv --rounds 4 --steps 30 "" SP_XCD_PAIR=0 > $O/ab_fp16x4_xcd_qkv.txt 2>&1; tail -3 $O/ab_fp16x4_xcd_qkv.txt
timeout 300 python tools/ab_live.py --config headline --kernel mlp3 --rounds 6 --steps 30 "" FP32_DIRECT=0 FP32_DIRECT=1 FP32_DIRECT=2 > $O/ab_fp32_direct_mlp3.txt 2>&1; tail -5 $O/ab_fp32_direct_mlp3.txt
timeout 300 python tools/ab_live.py --config headline --kernel qkv_kv --rounds 4 --steps 30 "" FP32_DIRECT=0 > $O/ab_fp32_direct_qkv.txt 2>&1; tail -3 $O/ab_fp32_direct_qkv.txt
timeout 300 python tools/ab_live.py --config fp16x4-b8 --kernel mlp0 --rounds 4 --steps 10 "" SP_UT=0 SP_XCD_PAIR=0 > $O/ab_fp16x4_b8.txt 2>&1; tail -4 $O/ab_fp16x4_b8.txt
timeout 300 python tools/ab_live.py --config real --kernel mlp3 --rounds 6 --steps 40 "" FP32_DIRECT=0 > $O/ab_fp32_direct_real.txt 2>&1; tail -3 $O/ab_fp32_direct_real.txt
