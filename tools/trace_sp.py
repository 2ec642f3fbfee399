"""Wave-level timeline of one mlp0_sp_kernel launch (split-16-bit main loop, gemm_split_glds.h): s_memtime stamps at the group
boundaries of one steady-state step, at the ends of prologue and loop and at kernel entry / exit, for every wave.
Needs the profiling build:  python -m onepose_amd.build_ext --force --profiling   (-> lib*_tuning.so; the product library is untouched).

    python tools/trace_sp.py [fp16x4]          (GATSSPG_SP_ABL=100 / GATSSPG_SP_MLP0_WIDE_MIN=1000 select the other tiles)
"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onepose_amd import _native, build_ext
_native.LIB_PATH = build_ext.tuning_path(build_ext.LIB_PATH)
import bench
prec = sys.argv[1] if len(sys.argv) > 1 else "fp16x4"
dev = torch.device("cuda:0")
w = bench.Weights(dev, prec); r = bench.Runner(dev, w)
lib = w.engine.lib
lib.gatsspg_debug_set_trace.argtypes = [ctypes.c_void_p]; lib.gatsspg_debug_set_trace.restype = None
NW = 8 * 4 * 16 * 8
buf = torch.zeros(NW * 24, dtype=torch.int64, device=dev)
for i in range(5): r.step(i)
torch.cuda.synchronize()
lib.gatsspg_debug_set_trace(buf.data_ptr())
r.step(0)
torch.cuda.synchronize()
lib.gatsspg_debug_set_trace(None)
t = buf.cpu().numpy().reshape(NW, 24)
t = t[t[:, 12] != 0]
print(f"{prec}: {len(t)} waves traced (last mlp0 launch of the frame); knobs:", {k: v for k, v in os.environ.items() if k.startswith("GATSSPG_")})
ent, pro, end_loop, end = t[:, 2], t[:, 3], t[:, 11], t[:, 12]
med = lambda a: float(np.median(a))
print(f"cycles (s_memtime ticks, shader clock): prologue {med(pro - ent):.0f}  loop {med(end_loop - pro):.0f} ({med(end_loop - pro) / 16:.0f} per step)  "
      f"epilogue {med(end - end_loop):.0f}  whole wave {med(end - ent):.0f}  (p10 / p90 whole: {np.percentile(end - ent, 10):.0f} / {np.percentile(end - ent, 90):.0f})")
names = ["products (I,P0) + split (I,P1) + hooks", "counted wait + barrier", "DMA requests + raw B reads + A reads", "products (I,P1) first half",
         "products (I,P1) second half + split (I+1,P0)", "A reads (I+1,P1)"]
if os.environ.get("GATSSPG_SP_SCHED") in ("3", "4"):   # the slot schedule: stamps at MFMA counts
    names = ["products (I,P0) 1..4 (+ 2 pairs of split (I,P1), hooks)", "products (I,P0) 5..8 (+ 2 pairs, head fold)", "counted wait + barrier",
             "products (I,P1) 1..2 (+ DMA pieces, B / A reads of slab I+1)", "products (I,P1) 3..4 (+ DMA piece, B reads)",
             "products (I,P1) 5..8 (+ 4 pairs of split (I+1,P0), DMA piece, A reads)"]
seg = t[:, 4:11].astype(np.int64)
for k, nme in enumerate(names):
    d = seg[:, k + 1] - seg[:, k]
    print(f"  step {5}: {nme:48s} median {med(d):6.0f}  p10 {np.percentile(d, 10):6.0f}  p90 {np.percentile(d, 90):6.0f}")
print(f"  step total (stamps 1..7): median {med(seg[:, 6] - seg[:, 0]):.0f}")
ghz = (end - ent) / ((t[:, 21] - t[:, 20]) * 10.0)
print(f"tick rate: s_memtime ticks per wall ns (100 MHz s_memrealtime): median {med(ghz):.3f}; whole wave {med(t[:, 21] - t[:, 20]) / 100.0:.2f} us")
ep = t[:, 16:20].astype(np.int64)
print(f"epilogue: loop end -> last barrier {med(ep[:, 0] - end_loop):.0f} | last fold + bias + tile to LDS {med(ep[:, 1] - ep[:, 0]):.0f} | barrier {med(ep[:, 2] - ep[:, 1]):.0f} | "
      f"tile stores issued {med(ep[:, 3] - ep[:, 2]):.0f} | statistics + exit {med(end - ep[:, 3]):.0f}")
span = (end.max() - ent.min())
print(f"launch span (first entry .. last exit): {span} ticks; entry spread {ent.max() - ent.min()}, exit spread {end.max() - end.min()}")
