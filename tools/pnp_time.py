"""Time the RANSAC-EPnP solver on a synthetic correspondence set (500 matches, 40 % outliers, 0.5 px noise)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onepose_amd import pnp, synthetic
p = synthetic.make_pnp_problem(500, 0.4, 0.5, 8)
p2, p3 = torch.from_numpy(p["pts_2d"]).cuda(), torch.from_numpy(p["pts_3d"]).cuda()
for it in (256, 1024, 10000):
    pnp.ransac_pnp_device(p["K"], p2, p3, 1000, iterations=it); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): out = pnp.ransac_pnp_device(p["K"], p2, p3, 1000, iterations=it)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(it, "iterations:", round(dt * 1e3, 3), "ms", out[2].cpu().tolist(), pnp.query_pose_error(out[0].cpu().numpy(), p["pose_gt"]))
