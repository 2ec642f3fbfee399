"""A/B of tuning-build knobs INSIDE ONE PROCESS (one box, one clock state, settings interleaved round by round): the boxes of this pool
differ by up to 40 % on identical binaries and drift with load, so separate bench.py processes cannot resolve a few per cent.

    python -m onepose_amd.build_ext --tuning
    python tools/ab_live.py [--config fp16x4] [--kernel mlp0] [--rounds 8] [--steps 30] "" SP_SCHED=2 SP_MLP0_WIDE_MIN=100000 ...

Each setting is a comma-separated list of KNOB=VALUE (GATSSPG_ prefix added here, "" = defaults); only knobs that the library reads per
launch can be flipped this way (SP_SCHED, SP_ABL, SP_MLP0_WIDE_MIN / _MAX, STAT_FUSED, SCORE_SPLIT, SPLIT_LOOP_BF16X3 / _BF16X6).
Per setting: median over the rounds of (a) the event-timed kernel (one launch per forward), (b) milliseconds per frame one frame at a time,
(c) frames/s with --slots (4) frames in flight.
"""
import argparse, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # one hardware queue per frame in flight (see bench.py); before torch loads the HIP runtime
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onepose_amd import _native, build_ext
_native.LIB_PATH = build_ext.tuning_path(build_ext.LIB_PATH)
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="fp16x4")
ap.add_argument("--kernel", default="mlp0")
ap.add_argument("--rounds", type=int, default=8)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--slots", type=int, default=4, help="frames in flight of the throughput leg (bench.py --streams)")
ap.add_argument("settings", nargs="*", default=[""])
a = ap.parse_args()
cfg = bench.CONFIGS[a.config]
NSLOT = a.slots
dev = torch.device("cuda:0")
w = bench.Weights(dev, cfg["precision"])
base = bench.Runner(dev, w, b=cfg["b"], n1=cfg["n1"], n2=cfg["n2"], golden_seed=bench.GOLDEN_SEEDS.get(cfg["golden"]))
slots = [bench.Runner(dev, w, base.shared_inputs, b=cfg["b"], n1=cfg["n1"], n2=cfg["n2"], own_stream=True) for _ in range(NSLOT)]
K = a.steps
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
r0 = slots[0]
with torch.cuda.stream(r0.stream):
    for e0, e1 in ev:
        e0.record(r0.stream); e1.record(r0.stream)


def apply(spec):
    for k in [k for k in os.environ if k.startswith("GATSSPG_")]:
        del os.environ[k]
    for kv in filter(None, spec.split(",")):
        k, v = kv.split("=")
        os.environ["GATSSPG_" + k] = v


res = {s: {"kern": [], "lat": [], "thr": [], "par": None} for s in a.settings}
for rnd in range(a.rounds + 1):
    for spec in a.settings:
        apply(spec)
        with torch.cuda.stream(r0.stream):
            for i in range(5):
                r0.step(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(K):
                r0.step_profiled(i, a.kernel, ev[i][0], ev[i][1])
            torch.cuda.synchronize()
            lat = (time.perf_counter() - t0) / K
        kern = float(np.mean([e0.elapsed_time(e1) for e0, e1 in ev]))
        for i in range(2 * NSLOT):
            slots[i % NSLOT].step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(2 * K):
            slots[i % NSLOT].step(i)
        torch.cuda.synchronize()
        thr = 2 * K * cfg["b"] / (time.perf_counter() - t0)
        if rnd:   # round 0 warms up
            res[spec]["kern"].append(kern); res[spec]["lat"].append(lat * 1e3 / cfg["b"]); res[spec]["thr"].append(thr)
        else:
            par = bench.golden_parity(r0, cfg)     # the setting's own parity number against the reference-run golden
            import hashlib   # digest of the golden frame's conf + matches under this setting: equal digests = bit-identical results
            dig = hashlib.sha256(r0.conf.cpu().numpy().tobytes() + r0.m0.cpu().numpy().tobytes()).hexdigest()[:12]
            res[spec]["par"] = par and (par["argmax_flips"], float(f"{par['max_abs_conf_err']:.3e}"), dig)
print(f"# {a.config}, kernel {a.kernel}, {a.rounds} interleaved rounds of {K} steps, one process")
print(f"# {'setting':40s} {a.kernel + '_ms':>10s} {'ms/frame':>10s} {'single fps':>11s} {'fps in flight':>16s}   (medians; min..max of ms/frame)   (arg-max flips, max |conf err|) vs the reference golden")
for spec in a.settings:
    r = res[spec]
    print(f"  {spec or 'defaults':40s} {np.median(r['kern']):10.5f} {np.median(r['lat']):10.4f} {1e3 / np.median(r['lat']):11.1f} {np.median(r['thr']):16.1f}   ({min(r['lat']):.4f}..{max(r['lat']):.4f})   {r['par']}")
