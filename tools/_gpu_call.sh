python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['config']['timed_pass_seconds'], d['config']['single_stream_frames_per_sec'], d['roofline']['frac'], d['cpu_baseline'])"
