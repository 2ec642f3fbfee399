"""hipGraph replay vs direct launches of the same forward (C ABI on a torch stream): frames/s one frame at a time for a few
shapes.  Run on the GPU box:  python tools/graph_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onepose_amd import GATsSuperGlue, synthetic  # noqa: E402

HP = dict(descriptor_dim=256, keypoints_encoder=[32, 64, 128], GNN_layers=["GATs", "self", "cross"] * 4, match_type="softmax",
          scale_factor=0.07, match_threshold=0.2, include_self=True, with_linear_transform=False, additional=False)


def timeit(fn, n):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    model = GATsSuperGlue(HP).eval()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(0).items()}, strict=True)
    model.to(dev)
    for n1, n2 in ((64, 128), (200, 500), (300, 1000), (500, 2000), (1000, 7000)):
        data = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.make_inputs(1, n1, n2, 8, seed=1).items()}
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            for _ in range(5):
                out = model.forward_batched(data)
            side.synchronize()
            direct = timeit(lambda: model.forward_batched(data), 200)
            ref = [t.clone() for t in out]
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                gout = model.forward_batched(data)
            g.replay()
            side.synchronize()
            same = all(torch.equal(a, b) for a, b in zip(ref, gout))
            graph = timeit(g.replay, 200)
        print(f"n1={n1:5d} n2={n2:5d}  direct {direct:.4f} ms  graph {graph:.4f} ms  ratio {direct / graph:.3f}  identical={same}", flush=True)


if __name__ == "__main__":
    main()
