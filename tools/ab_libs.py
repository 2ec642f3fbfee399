"""A/B of two BUILDS of the matcher library inside one process (rounds interleaved on the same resident frames): ab_live.py flips knobs of one
tuning build; this loads two .so files side by side -- for experiments that change compile-time attributes (register caps, occupancy hints).

    python tools/ab_libs.py [--config headline] [--rounds 6] [--steps 40] onepose_amd/lib/libgatsspg_hip.so onepose_amd/lib/libgatsspg_hip_w5.so
"""
import argparse, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onepose_amd import _native
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="headline")
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--slots", type=int, default=4)
ap.add_argument("libs", nargs="+")
a = ap.parse_args()
cfg = bench.CONFIGS[a.config]
dev = torch.device("cuda:0")
sets = {}
shared = None
for path in a.libs:
    _native._lib = None                      # bind the next engine to this build
    _native.LIB_PATH = os.path.abspath(path)
    w = bench.Weights(dev, cfg["precision"])
    base = bench.Runner(dev, w, shared, b=cfg["b"], n1=cfg["n1"], n2=cfg["n2"], golden_seed=bench.GOLDEN_SEEDS.get(cfg["golden"]))
    shared = base.shared_inputs
    # both builds drive THE SAME streams: a second stream set created later shares hardware queues with the first and runs the same calls
    # 8-12 % slower (profiles/r06g_module_overhead_*_streams.txt) -- that would be charged to whichever library is listed second
    first = next(iter(sets.values()))[2] if sets else None
    sets[path] = (w, base, [bench.Runner(dev, w, shared, b=cfg["b"], n1=cfg["n1"], n2=cfg["n2"], own_stream=True,
                                         stream=first[i].stream if first else None) for i in range(a.slots)])
K = a.steps
res = {p: {"lat": [], "thr": [], "par": None} for p in a.libs}
for rnd in range(a.rounds + 1):
    for p in a.libs:
        w, base, slots = sets[p]
        r0 = slots[0]
        for i in range(5):
            r0.step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            r0.step(i)
        torch.cuda.synchronize()
        lat = (time.perf_counter() - t0) / K
        for i in range(2 * a.slots):
            slots[i % a.slots].step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(3 * K):
            slots[i % a.slots].step(i)
        torch.cuda.synchronize()
        thr = 3 * K * cfg["b"] / (time.perf_counter() - t0)
        if rnd:
            res[p]["lat"].append(lat * 1e3 / cfg["b"]); res[p]["thr"].append(thr)
        else:
            par = bench.golden_parity(r0, cfg)
            res[p]["par"] = par and (par["argmax_flips"], float(f"{par['max_abs_conf_err']:.3e}"))
print(f"# {a.config}: {a.rounds} interleaved rounds of {K} steps one frame at a time + {3 * K} steps with {a.slots} in flight, one process, two builds")
print(f"# {'library':48s} {'ms/frame':>10s} {'single fps':>11s} {'fps in flight':>14s}   (min..max in flight)   (arg-max flips, max |conf err|)")
for p in a.libs:
    r = res[p]
    print(f"  {os.path.basename(p):48s} {np.median(r['lat']):10.4f} {1e3 / np.median(r['lat']):11.1f} {np.median(r['thr']):14.1f}   ({min(r['thr']):.1f}..{max(r['thr']):.1f})   {r['par']}")
