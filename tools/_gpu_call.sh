R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2o; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_spp_hip_parity.py -m gpu -q --timeout=180 -p no:cacheprovider -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python tools/ab_tuning.py --kernel mlp0 "" 2>&1 | tee $O/ab.log
python tools/ab_tuning.py --config bf16x3 --kernel mlp0 "" 2>&1 | tee -a $O/ab.log
python bench.py --extractor --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('extractor', d['value'], d['config']['single_image_latency_ms'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o r -- python $R/bench.py --steps 50 --warmup 5 --streams 1 --reps 1 --no-cpu-baseline > $O/prof.log 2>&1
python $R/tools/rocpd_stats.py $(find $O/prof -name "*.db" | head -1) 2>&1 | head -14
