"""Per-workgroup timeline of one mlp0_kernel launch (entry / loop end / exit, CU placement, shader clock).
Needs the profiling build:  python -m onepose_amd.build_ext --force --profiling   (-> lib*_tuning.so; the product library is untouched)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onepose_amd import _native, build_ext
_native.LIB_PATH = build_ext.tuning_path(build_ext.LIB_PATH)
import bench
dev = torch.device("cuda:0")
w = bench.Weights(dev); r = bench.Runner(dev, w)
# --inflight N (round 5): N other frames are kept running on streams of their own while the traced launch executes (the protocol of bench.py's
# `value`): what does the shader clock inside the kernel do when the socket sits at its power cap?
NBG = int(sys.argv[sys.argv.index("--inflight") + 1]) if "--inflight" in sys.argv else 0
bg = [bench.Runner(dev, w, r.shared_inputs, own_stream=True) for _ in range(NBG)]
lib = w.engine.lib
lib.gatsspg_debug_set_trace.argtypes = [ctypes.c_void_p]; lib.gatsspg_debug_set_trace.restype = None
G = 8 * 4 * 16
buf = torch.zeros(G * 8, dtype=torch.int64, device=dev)
for i in range(5): r.step(i)
torch.cuda.synchronize()
if NBG:
    for i in range(300):            # ~80 ms of background frames: the power manager has settled when the traced frame runs
        bg[i % NBG].step(i)
lib.gatsspg_debug_set_trace(buf.data_ptr())
r.step(0)
lib.gatsspg_debug_set_trace(None)   # (host-side: launches enqueued from here on are not traced)
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(G, 8)
t = t[t[:, 5] != 0]
print("MLP0_TILE", os.environ.get("GATSSPG_MLP0_TILE"), "| other frames in flight during the traced launch:", NBG)
# the buffer holds the LAST mlp0 launch of the frame (every launch overwrites it)
t0 = t[:, 2].min()
ent, loop, end = (t[:, 2] - t0) / 100.0, (t[:, 4] - t0) / 100.0, (t[:, 5] - t0) / 100.0
ghz = t[:, 3] / ((t[:, 5] - t[:, 2]) * 10.0)
print(f"shader clock (s_memtime ticks / wall ns): min {ghz.min():.3f} med {np.median(ghz):.3f} max {ghz.max():.3f} GHz")
print(f"blocks {len(t)}  entry: min {ent.min():.2f} med {np.median(ent):.2f} max {ent.max():.2f} us")
print(f"mainloop(+prologue) duration: min {(loop-ent).min():.2f} med {np.median(loop-ent):.2f} max {(loop-ent).max():.2f} us")
print(f"epilogue duration: min {(end-loop).min():.2f} med {np.median(end-loop):.2f} max {(end-loop).max():.2f} us")
print(f"block end: min {end.min():.2f} med {np.median(end):.2f} p90 {np.percentile(end,90):.2f} max {end.max():.2f} us")
xcc = t[:, 1] & 0xf
for x in range(8):
    m = xcc == x
    print(f" xcc{x}: n={m.sum()} entry {ent[m].min():.2f}-{ent[m].max():.2f}  end {end[m].min():.2f}-{end[m].max():.2f}  med dur {np.median((end-ent)[m]):.2f}")
cu = (t[:, 1] & 0xf) * 4096 + ((t[:, 0] >> 13) & 7) * 256 + ((t[:, 0] >> 8) & 0xf)
u, c = np.unique(cu, return_counts=True)
print("blocks per CU:", dict(zip(*np.unique(c, return_counts=True))))
one = np.isin(cu, u[c == 1])
print(f"dur on CUs with 1 block: {np.median((end-ent)[one]):.2f} us; with 2 blocks: {np.median((end-ent)[~one]):.2f} us")
# which operand panel do co-resident workgroups share?
import collections
by_cu = collections.defaultdict(list)
for k, row in zip(cu, t):
    by_cu[k].append((int(row[6]), int(row[7])))
pairs = [v for v in by_cu.values() if len(v) == 2]
same_rt = sum(1 for a, b in pairs if a[0] == b[0]); same_ct = sum(1 for a, b in pairs if a[1] == b[1])
print(f"CUs with 2 workgroups: {len(pairs)}; same row tile (weights shared in L1): {same_rt}; same column tile: {same_ct}")
