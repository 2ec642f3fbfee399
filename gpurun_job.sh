cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_tile_variants.py -x -q -m gpu 2>&1 | tail -3
for v in 0 3 2 0 3 2; do GATSSPG_MLP0_TILE=$v python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mlp0 tile $v', d['value'], d['config']['single_frame_latency_ms'], d['roofline']['kernel_ms'], d['roofline']['achieved'])"; done
