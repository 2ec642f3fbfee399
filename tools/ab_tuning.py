"""A/B runs of the tuning-build knobs on the GPU box: one bench.py process per setting, same inputs, same protocol.

    python -m onepose_amd.build_ext --tuning            # lib*_tuning.so (the package never loads these)
    python tools/ab_tuning.py [--config headline] [--kernel mlp3] [--extractor] "" MLP3_TILE=1 MLP0_BTILE=1,MLP3_TILE=1 ...

Each argument is a comma-separated list of KNOB=VALUE (GATSSPG_ / SPP_ prefix added here; "" = the defaults).  Prints, per
setting: frames/s with 3 frames in flight, one-frame-at-a-time frames/s, the event-timed kernel and the parity number of the
line; appends the JSON lines to gpurun_out/ab_tuning.jsonl.
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="headline")
ap.add_argument("--kernel", default="mlp0")
ap.add_argument("--steps", default="100")
ap.add_argument("--extractor", action="store_true")
ap.add_argument("settings", nargs="*", default=[""])
a = ap.parse_args()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
for spec in a.settings:
    env = dict(os.environ)
    for kv in filter(None, spec.split(",")):
        k, v = kv.split("=")
        env[("SPP_" if a.extractor else "GATSSPG_") + k] = v
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--tuning-lib", "--steps", a.steps, "--warmup", "10", "--reps", "3",
           "--no-cpu-baseline"]
    cmd += ["--extractor"] if a.extractor else ["--config", a.config, "--kernel", a.kernel]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=ROOT)
    line = next((l for l in r.stdout.splitlines() if l.startswith("{")), None)
    if line is None:
        print(f"[{spec}] FAILED rc={r.returncode}: {r.stderr[-400:]}", flush=True)
        continue
    d = json.loads(line)
    with open(os.path.join(ROOT, "gpurun_out", "ab_tuning.jsonl"), "a") as f:
        f.write(json.dumps({"setting": spec, **d}) + "\n")
    c, rf, pc = d["config"], d["roofline"], d.get("parity_check") or {}
    lat = c.get("single_stream_frames_per_sec", c.get("single_image_latency_ms"))
    print(f"[{spec or 'defaults'}] {d['value']} {d['unit']} in flight; one at a time {lat}; {rf['kernel']} {rf['kernel_ms']} ms "
          f"({rf['achieved']} {rf['unit']}); parity err {pc.get('max_abs_conf_err')} flips {pc.get('argmax_flips')}", flush=True)
