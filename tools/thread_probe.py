"""Host side of frames in flight: is ONE launch thread enough?  For a bench config: (a) the CPU time one thread spends enqueuing a frame
(48 launches through the C ABI) while the GPU is kept busy, (b) frames/s with the K frames dealt round-robin by one thread (bench.py's
loop), (c) frames/s with one Python thread per in-flight slot (ctypes drops the GIL inside gatsspg_forward).
    python tools/thread_probe.py [--config real] [--slots 4] [--steps 400]"""
import argparse, os, sys, threading, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="real")
ap.add_argument("--slots", type=int, default=4)
ap.add_argument("--steps", type=int, default=400)
ap.add_argument("--rounds", type=int, default=5)
a = ap.parse_args()
cfg = bench.CONFIGS[a.config]
dev = torch.device("cuda:0")
w = bench.Weights(dev, cfg["precision"])
base = bench.Runner(dev, w, b=cfg["b"], n1=cfg["n1"], n2=cfg["n2"], golden_seed=bench.GOLDEN_SEEDS.get(cfg["golden"]))
slots = [bench.Runner(dev, w, base.shared_inputs, b=cfg["b"], n1=cfg["n1"], n2=cfg["n2"], own_stream=True) for _ in range(a.slots)]
S, K = a.slots, a.steps
for i in range(2 * S):
    slots[i % S].step(i)
torch.cuda.synchronize()


def one_thread():
    t0 = time.perf_counter()
    for i in range(K):
        slots[i % S].step(i)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    return K * cfg["b"] / (time.perf_counter() - t0), t_enq / K * 1e6


def per_slot_threads():
    def work(s):
        for i in range(K // S):
            slots[s].step(i)
    th = [threading.Thread(target=work, args=(s,)) for s in range(S)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    return (K // S) * S * cfg["b"] / (time.perf_counter() - t0)


r1, r2, enq = [], [], []
for _ in range(a.rounds):
    f, e = one_thread(); r1.append(f); enq.append(e)
    r2.append(per_slot_threads())
print(f"{a.config}: {S} frames in flight, {K} steps x {a.rounds} rounds (medians): one launch thread {np.median(r1):.1f} frames/s "
      f"(enqueue {np.median(enq):.1f} us of CPU per frame = {np.median(enq) / 48:.2f} us per launch; a frame lasts {1e6 / np.median(r1):.1f} us), "
      f"one thread per slot {np.median(r2):.1f} frames/s")
