"""Host cost of one GATsSuperGlue.forward(data) call against the raw C-ABI call (bench.Runner.step): enqueue time per frame with nothing
synchronised in between, four frames in flight, plus a cProfile of the module path.   python tools/module_overhead.py [precision]"""
import cProfile, os, pstats, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from onepose_amd import StreamRing

prec = sys.argv[1] if len(sys.argv) > 1 else "fp16x4"
dev = torch.device("cuda:0")
w = bench.Weights(dev, prec)
base = bench.Runner(dev, w)
slots = [bench.Runner(dev, w, base.shared_inputs, own_stream=True) for _ in range(4)]
d3, d2db, queries = base.shared_inputs
frames = [{"keypoints2d": torch.zeros(1, bench.N1, 2, device=dev), "keypoints3d": torch.zeros(1, bench.N2, 3, device=dev),
           "descriptors2d_query": q, "descriptors3d_db": d3, "descriptors2d_db": d2db} for q in queries]
ring = StreamRing(dev, 4)
model = w.model


def run_raw(n):
    for i in range(n):
        slots[i % 4].step(i)


def run_mod(n):
    with torch.no_grad():
        for i in range(n):
            with ring.next():
                model(frames[i % 4])


ring_own = ring
ring_slots = StreamRing(dev, 4, streams=[sl.stream for sl in slots])


def run_mod_slots(n):
    global ring
    ring = ring_slots
    run_mod(n)
    ring = ring_own


def run_raw_ring(n):
    saved = [sl.stream for sl in slots]
    for sl, st in zip(slots, ring_own.streams):
        sl.stream = st
    run_raw(n)
    for sl, st in zip(slots, saved):
        sl.stream = st


for name, fn in (("raw C ABI", run_raw), ("module", run_mod), ("mod@slots", run_mod_slots), ("raw@ring", run_raw_ring), ("raw C ABI", run_raw), ("module", run_mod),
                 ("mod@slots", run_mod_slots), ("raw@ring", run_raw_ring)):
    fn(40)
    torch.cuda.synchronize()
    for n in (200, 1000):
        t0 = time.perf_counter()
        fn(n)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{name:10s} {prec} n={n:5d}: enqueue {1e6 * (t1 - t0) / n:7.1f} us/frame, with drain {1e6 * (t2 - t0) / n:7.1f} us/frame = {n / (t2 - t0):7.1f} frames/s")
pr = cProfile.Profile()
pr.enable()
run_mod(400)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
print("allocator:", {k: v for k, v in torch.cuda.memory_stats().items() if k in ("num_alloc_retries", "allocation.all.allocated", "segment.all.allocated", "num_device_alloc", "num_device_free")})
