"""Throughput of one stream vs the number of frames per forward (batch b): amortisation of the fixed per-launch cost."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
w = bench.Weights(dev)
for b in (1, 2, 4, 8):
    r = bench.Runner(dev, w, b=b)
    for i in range(5): r.step(i)
    torch.cuda.synchronize()
    n = max(10, 120 // b)
    t0 = time.perf_counter()
    for i in range(n): r.step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"batch {b}: {dt*1e3:.3f} ms per forward, {b/dt:.1f} frames/s, {bench.f_alg(1000,7000,8)*b/dt/1e12:.1f} TF/s algorithmic")
