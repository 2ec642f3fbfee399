#!/bin/bash
# Round profiles on the GPU box: GPU test suite, bench lines, rocprofv3 kernel stats (one frame at a time and 3 in flight), PMC passes
# (FETCH_SIZE / WRITE_SIZE / SQ counters, each in its own run).  Writes gpurun_out/$1/ ; copy what should be judged to profiles/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r04}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout=300 -p no:cacheprovider 2>&1 | tail -5 > $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" >> $O/pytest_gpu.log 2>&1
python bench.py > $O/bench_headline.json 2> $O/bench_headline.err
python bench.py --steps 20 --warmup 5 > $O/bench_headline_driver_protocol.json 2>/dev/null
python bench.py --amortised --no-cpu-baseline --no-side-arithmetics > $O/bench_amortised.json 2>/dev/null
for c in bf16x3 bf16x6 fp16x3 fp16x4 fp32-b8 bf16x6-b8 fp16x3-b8 fp16x4-b8 real real-b8 fp16x4-real fp16x4-real-b8 stress stress-b4 fp16x4-stress fp16x4-stress-b4; do
  steps=100; [[ $c == *b8* || $c == *stress* ]] && steps=20
  python bench.py --config $c --steps $steps --warmup 5 --no-cpu-baseline > $O/bench_$c.json 2>/dev/null
done
python bench.py --kernel gats --no-cpu-baseline --no-side-arithmetics > $O/bench_kernel_gats.json 2>/dev/null
python bench.py --kernel conf_finalize --no-cpu-baseline --no-side-arithmetics > $O/bench_kernel_conf_finalize.json 2>/dev/null
python bench.py --extractor --no-cpu-baseline > $O/bench_extractor.json 2>/dev/null
python bench.py --extractor --extractor-precision fp16x4 --no-cpu-baseline > $O/bench_extractor_fp16x4.json 2>/dev/null
python bench.py --pipeline --matcher-precision fp16x4 --extractor-precision fp16x4 > $O/bench_pipeline_fp16x4.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 50 --warmup 5 --reps 1 --no-cpu-baseline --no-side-arithmetics"
rocprofv3 --kernel-trace --stats -d $O/prof_s1 -o r -- $B --streams 1 > $O/prof_s1.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_s3 -o r -- $B --streams 3 > $O/prof_s3.log 2>&1
for c in fp16x4 fp16x4-b8 bf16x6 fp16x3 real fp16x4-stress-b4; do
  st=50; [[ $c == *stress* || $c == *b8* ]] && st=10
  rocprofv3 --kernel-trace --stats -d $O/prof_$c -o r -- python $R/bench.py --steps $st --warmup 5 --reps 1 --no-cpu-baseline --no-side-arithmetics --streams 1 --config $c > $O/prof_$c.log 2>&1
done
P="python $R/bench.py --steps 6 --warmup 2 --reps 1 --streams 1 --no-cpu-baseline --no-side-arithmetics"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch_fp16x4 -o r -- $P --config fp16x4 > $O/pmc_fetch_fp16x4.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write_fp16x4 -o r -- $P --config fp16x4 > $O/pmc_write_fp16x4.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $O/pmc_sq_fp16x4 -o r -- $P --config fp16x4 > $O/pmc_sq_fp16x4.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES -d $O/pmc_sq -o r -- $P > $O/pmc_sq.log 2>&1
for d in prof_s1 prof_s3 prof_fp16x4 prof_fp16x4-b8 prof_bf16x6 prof_fp16x3 prof_real prof_fp16x4-stress-b4; do python $R/tools/rocpd_stats.py $(find $O/$d -name "*.db" | head -1) > $O/kernel_stats_$d.txt 2>&1; done
for d in pmc_fetch_fp16x4 pmc_write_fp16x4 pmc_sq_fp16x4 pmc_sq; do python $R/tools/rocpd_pmc.py $(find $O/$d -name "*.db" | head -1) > $O/$d.txt 2>&1; done
find $O -name "*.db" -delete
ls $O | head -80
