import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from onepose_amd import pnp, synthetic
from oracle import pnp_oracle as po

def cand_errs(pw, uv, k):
    fu, fv, uc, vc = k[0, 0], k[1, 1], k[0, 2], k[1, 2]
    cws = po._control_points(pw); al = po._barycentric(pw, cws); m = po._fill_m(al, uv, fu, fv, uc, vc)
    w, vecs = np.linalg.eigh(m.T @ m); v = [vecs[:, i] for i in range(4)]
    l = po._l6x10(v); rho = np.array([((cws[a] - cws[b]) ** 2).sum() for (a, b) in po.PAIRS])
    out = []
    for na in (1, 2, 3):
        b = po._gauss_newton(l, rho, po._betas_approx(na, l, rho))
        r, t, e = po._r_and_t(v, b, al, pw, uv, fu, fv, uc, vc)
        out.append((e, r))
    return w[:5], out

for noise in (0.0, 0.2, 0.5):
    p = synthetic.make_pnp_problem(200, 0.0, noise, 7)
    for n in (5, 6, 64, 65, 200):
        pw = p["pts_3d"][:n].astype(np.float64) * 1000; uv = p["pts_2d"][:n].astype(np.float64)
        r, t = po.epnp(pw, uv, p["K"])
        pose = pnp.epnp(p["K"], torch.from_numpy(p["pts_2d"][:n]).cuda(), torch.from_numpy(p["pts_3d"][:n]).cuda(), scale=1000).cpu().numpy()
        w, c = cand_errs(pw, uv, p["K"])
        print(f"noise {noise} n {n}: dR {np.abs(pose[:, :3] - r).max():.2e} dt {np.abs(pose[:, 3] - t / 1000).max():.2e}  oracle cand errs",
              [f"{e:.5f}" for e, _ in c], "gpu-vs-cand dR", [f"{np.abs(pose[:, :3] - rr).max():.1e}" for _, rr in c], "eig", [f"{x:.2e}" for x in w])
