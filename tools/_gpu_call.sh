for s in 2 3 4 6; do python bench.py --streams $s --steps 120 --warmup 12 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('streams', $s, d['value'], d['config']['timed_pass_seconds'])"; done
