"""Informative baseline: the reference ALGORITHM through stock PyTorch-ROCm ops on this GPU (one ATen /
rocBLAS / MIOpen launch per op, like the reference module), headline shape, fp32, batch 1.

    python tools/torch_eager_baseline.py [--iters 50]
"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onepose_amd import synthetic
from oracle import torch_oracle, gatsspg_oracle

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--n1", type=int, default=1000)
ap.add_argument("--n2", type=int, default=7000)
args = ap.parse_args()
dev = torch.device("cuda:0")
hp = dict(gatsspg_oracle.DEFAULT_HPARAMS)
sd = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.make_state_dict(0).items()}
data = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.make_inputs(1, args.n1, args.n2, 8, seed=1).items()}
torch.backends.cuda.matmul.allow_tf32 = False
for _ in range(5):
    torch_oracle.forward(sd, data, hp)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.iters):
    torch_oracle.forward(sd, data, hp)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.iters
print(json.dumps({"baseline": "reference algorithm via stock PyTorch-ROCm eager ops on this GPU", "torch": torch.__version__,
                  "n_2d": args.n1, "n_3d": args.n2, "dtype": "f32", "ms_per_frame": round(dt * 1e3, 3),
                  "frames_per_sec": round(1 / dt, 2)}))
