"""Single-stream latency / throughput of the other BASELINE.json shapes (informative; bench.py stays on configs[1])."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
w = bench.Weights(dev)
out = []
for name, b, n1, n2 in (("configs[0] shape 500/2000 b=1", 1, 500, 2000), ("configs[1] 1000/7000 b=1", 1, 1000, 7000),
                        ("configs[2] per-GPU share 1000/7000 b=8", 8, 1000, 7000), ("configs[4] stress 1000/20000 b=1", 1, 1000, 20000),
                        ("configs[4] stress per-GPU share 1000/20000 b=4", 4, 1000, 20000)):
    r = bench.Runner(dev, w, b=b, n1=n1, n2=n2)
    for i in range(3): r.step(i)
    torch.cuda.synchronize()
    n = max(5, int(60 / (b * (n1 + n2) / 8000)))
    t0 = time.perf_counter()
    for i in range(n): r.step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    fa = bench.f_alg(n1, n2, 8) * b
    ba = (4 * 256 * (n1 + n2 + n2 * 8) + 4 * 5587200 / b + 4 * n1 * n2 + 12 * (n1 + n2)) * b
    rec = {"config": name, "ms_per_forward": round(dt * 1e3, 3), "frames_per_sec": round(b / dt, 1),
           "algorithmic_TFLOPs": round(fa / dt / 1e12, 1), "algorithmic_GBs": round(ba / dt / 1e9, 1)}
    out.append(rec); print(json.dumps(rec), flush=True)
    del r; torch.cuda.empty_cache()
