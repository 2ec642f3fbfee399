#!/bin/bash
# Round profiles on the GPU box: bench lines, rocprofv3 kernel stats (one frame at a time and 3 in flight), PMC passes
# (FETCH_SIZE / WRITE_SIZE / SQ busy, each in its own run).  Writes gpurun_out/$1/ ; copy what should be judged to profiles/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r03}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
python bench.py > $O/bench_headline.json 2> $O/bench_headline.err
python bench.py --steps 20 --warmup 5 > $O/bench_headline_driver_protocol.json 2>/dev/null
python bench.py --amortised --no-cpu-baseline --no-side-arithmetics > $O/bench_amortised.json 2>/dev/null
for c in bf16x3 bf16x6 fp16x3 fp16x4 fp32-b8 bf16x3-b8 bf16x6-b8 fp16x3-b8 fp16x4-b8 real real-b8 stress stress-b4; do
  steps=100; [[ $c == *b8* || $c == stress* ]] && steps=20
  python bench.py --config $c --steps $steps --warmup 5 --no-cpu-baseline > $O/bench_$c.json 2>/dev/null
done
python bench.py --torch-eager > $O/bench_torch_eager.json 2>/dev/null
python bench.py --extractor --no-cpu-baseline > $O/bench_extractor.json 2>/dev/null
python bench.py --pipeline > $O/bench_pipeline.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 50 --warmup 5 --reps 1 --no-cpu-baseline --no-side-arithmetics"
rocprofv3 --kernel-trace --stats -d $O/prof_s1 -o r -- $B --streams 1 > $O/prof_s1.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_s3 -o r -- $B --streams 3 > $O/prof_s3.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_b6 -o r -- $B --streams 1 --config bf16x6 > $O/prof_b6.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_b3 -o r -- $B --streams 1 --config bf16x3 > $O/prof_b3.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_real -o r -- $B --streams 1 --config real > $O/prof_real.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_spp -o r -- python $R/bench.py --extractor --steps 50 --warmup 5 --reps 1 --no-cpu-baseline --streams 1 > $O/prof_spp.log 2>&1
P="python $R/bench.py --steps 6 --warmup 2 --reps 1 --streams 1 --no-cpu-baseline --no-side-arithmetics"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o r -- $P > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o r -- $P > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES -d $O/pmc_sq -o r -- $P > $O/pmc_sq.log 2>&1
for d in prof_s1 prof_s3 prof_b6 prof_b3 prof_real prof_spp; do python $R/tools/rocpd_stats.py $(find $O/$d -name "*.db" | head -1) > $O/kernel_stats_$d.txt 2>&1; done
for d in pmc_fetch pmc_write pmc_sq; do python $R/tools/rocpd_pmc.py $(find $O/$d -name "*.db" | head -1) > $O/$d.txt 2>&1; done
find $O -name "*.db" -size +20M -delete
ls -la $O | head -60
