set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2a
timeout 900 python -m pytest tests -m gpu -q --timeout=180 -p no:cacheprovider > $R/gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $R/gpurun_out/r2a/pytest.log
tail -25 $R/gpurun_out/r2a/pytest.log
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $R/gpurun_out/r2a/bench_headline.json 2> $R/gpurun_out/r2a/bench_headline.err; echo "bench rc=$?"
timeout 300 python bench.py --config bf16x3 --steps 100 --warmup 10 --no-cpu-baseline > $R/gpurun_out/r2a/bench_bf16x3.json 2> $R/gpurun_out/r2a/bench_bf16x3.err
timeout 300 python bench.py --config bf16x3-b8 --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/r2a/bench_bf16x3_b8.json 2> $R/gpurun_out/r2a/bench_bf16x3_b8.err
cat $R/gpurun_out/r2a/bench_*.json | cut -c1-1500
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2a/prof -o r -- python $R/bench.py --steps 50 --warmup 5 --streams 1 --reps 1 --no-cpu-baseline > $R/gpurun_out/r2a/prof.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2a/prof_bf -o r -- python $R/bench.py --config bf16x3 --steps 50 --warmup 5 --streams 1 --reps 1 --no-cpu-baseline > $R/gpurun_out/r2a/prof_bf.log 2>&1
ls -la $R/gpurun_out/r2a/prof* | head
python $R/tools/rocpd_stats.py $(find $R/gpurun_out/r2a/prof -name "*.db" | head -1) 2>&1 | head -30
python $R/tools/rocpd_stats.py $(find $R/gpurun_out/r2a/prof_bf -name "*.db" | head -1) 2>&1 | head -30
