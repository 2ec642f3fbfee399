#!/usr/bin/env python
"""Soak run of the drop-in module (serving evidence, not a benchmark): N query frames through GATsSuperGlue.forward(data) with four
frames in flight on onepose_amd.StreamRing, three databases of different size interleaved (inference.py:185-198 walks objects), every
arithmetic of the C ABI.  Checks, frame by frame on the GPU (one flag tensor, read once at the end -- no per-frame synchronisation):

  * every output (conf, matches0/1, matching_scores0/1) of frame i equals, bit for bit, the output of the SAME inputs computed once
    up front one frame at a time on the null stream  (no cross-stream scratch sharing, no stale workspace, no race that only shows
    after thousands of launches);
  * torch's allocator statistics after the warm-up round do not grow (no per-frame leak in the workspace / output caches).

    python tools/soak.py [--frames 20000] [--precision fp32]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402

from onepose_amd import GATsSuperGlue, StreamRing, synthetic  # noqa: E402

HP = {"descriptor_dim": 256, "keypoints_encoder": [32, 64, 128], "GNN_layers": ["GATs", "self", "cross"] * 4, "match_type": "softmax",
      "scale_factor": 0.07, "match_threshold": 0.2, "include_self": True, "additional": False, "with_linear_transform": False}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=20000)
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--queries", type=int, default=6, help="distinct query frames per database")
    ap.add_argument("--poison", action="store_true",
                    help="self-test of the check: one element of ONE expected conf is changed by one ulp -- the run must count exactly the frames that replay it and exit 1")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    sd = synthetic.make_state_dict(0)
    model = GATsSuperGlue(HP, precision=a.precision).eval()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model.to(dev)
    shapes = [(500, 2000), (1000, 7000), (777, 3333)]          # three objects: OnePose's own size, the headline size, a ragged one
    work = []                                                   # (data dict, expected outputs)
    for oi, (n1, n2) in enumerate(shapes):
        base = synthetic.make_inputs(1, n1, n2, 8, seed=300 + oi)
        rs = np.random.RandomState(400 + oi)
        for qi in range(a.queries):
            d = dict(base)
            if qi:
                q = rs.standard_normal((1, 256, n1)).astype(np.float32)
                d["descriptors2d_query"] = q / np.linalg.norm(q, axis=1, keepdims=True)
            work.append({k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in d.items()})
    with torch.no_grad():
        expect = []
        for d in work:
            pred, conf = model(d)
            expect.append((conf, pred["matches0"], pred["matches1"], pred["matching_scores0"], pred["matching_scores1"]))
        torch.cuda.synchronize()
        if a.poison:
            c = expect[0][0].view(-1)
            c[c.numel() // 2] = torch.nextafter(c[c.numel() // 2], torch.tensor(2.0, device=dev))
        ring = StreamRing.shared(dev)
        bad = {st.cuda_stream: torch.zeros(1, device=dev, dtype=torch.int64) for st in ring.streams}   # one counter per stream: no cross-stream RMW
        order = np.random.RandomState(1).randint(0, len(work), size=a.frames)

        def run(idx):
            for i in idx:
                with ring.next():
                    pred, conf = model(work[i])
                    e = expect[i]
                    got = (conf, pred["matches0"], pred["matches1"], pred["matching_scores0"], pred["matching_scores1"])
                    miss = sum((g != x).any().to(torch.int64) for g, x in zip(got, e))
                    bad[torch.cuda.current_stream(dev).cuda_stream].add_(miss)     # enqueued on the frame's own stream
            ring.synchronize()

        run(order[: 4 * len(work)])        # warm-up round: every (shape, stream) workspace exists afterwards
        torch.cuda.synchronize()
        mem0 = (torch.cuda.memory_allocated(dev), torch.cuda.memory_reserved(dev))
        t0 = time.perf_counter()
        run(order)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        mem1 = (torch.cuda.memory_allocated(dev), torch.cuda.memory_reserved(dev))
    out = {"tool": "tools/soak.py", "precision": a.precision, "frames": int(a.frames), "databases": shapes, "distinct_frames": len(work),
           "frames_in_flight": len(ring.streams), "seconds": round(dt, 2),
           "frames_per_sec_with_the_on_gpu_comparison_of_every_output": round(a.frames / dt, 1),
           "frames_with_any_output_differing_from_the_serial_result": int(sum(b.item() for b in bad.values())),
           "allocated_bytes_before_after": [mem0[0], mem1[0]], "reserved_bytes_before_after": [mem0[1], mem1[1]]}
    if a.poison:
        out["poisoned_self_test"] = {"frames_replaying_the_poisoned_expectation": int((order == 0).sum()) + int((order[: 4 * len(work)] == 0).sum())}
    print(json.dumps(out))
    ok = out["frames_with_any_output_differing_from_the_serial_result"] == 0 and mem1[0] <= mem0[0] and mem1[1] <= mem0[1]
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
