"""Golden vectors for the pose solver, produced by RUNNING OpenCV -- the dependency behind the reference's ``ransac_PnP``
(src/utils/eval_utils.py:18-42: ``cv2.solvePnPRansac(..., reprojectionError=5, iterationsCount=10000, flags=cv2.SOLVEPNP_EPNP)``).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_pnp_golden.py          # needs `import cv2`

OpenCV is NOT installed in the build image of rounds 1-6 (no network): this script has never run there and ``tests/golden/pnp_cv2.npz``
does not exist; ``oracle/pnp_oracle.py`` therefore says PARITY UNPINNED and ``tests/test_pnp.py::test_pose_solver_against_opencv_golden``
skips.  The day cv2 is importable, running this script commits the vectors that pin the oracle and the HIP solver:

  for each seeded synthetic problem of ``onepose_amd.synthetic.make_pnp_problem`` (n correspondences, outlier fraction, pixel noise):
    rvec / tvec / inliers of cv2.solvePnPRansac with the reference's exact arguments (dist = zeros(8, 1), scale = 1000),
    the pose ``ransac_PnP`` builds from them (Rodrigues, tvec / scale), and cv2.solvePnP(SOLVEPNP_EPNP) on the first 12 exact points
    (the EPnP core without RANSAC: deterministic, compared tightly).

Only OUTPUTS are stored; the problems are rebuilt from their seeds.  If the reference checkout is present its own ``ransac_PnP`` is
imported and used (so the golden is literally the reference's output); otherwise the call is made with the same arguments.
"""
import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("ONEPOSE_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import numpy as np  # noqa: E402

CASES = {  # name -> make_pnp_problem(n, outlier fraction, pixel noise, seed)
    "clean40": (40, 0.0, 0.0, 0), "outl30": (300, 0.3, 0.0, 1), "outl60": (1000, 0.6, 0.0, 2), "noisy": (300, 0.3, 0.5, 1),
    "bench": (500, 0.4, 0.5, 8), "few": (12, 0.0, 0.2, 4),
}


def main():
    try:
        import cv2
    except ImportError:
        raise SystemExit("make_pnp_golden.py needs OpenCV (import cv2 failed): the pose solver's parity stays UNPINNED until it runs")
    from onepose_amd import synthetic
    ransac_pnp = None
    if os.path.isdir(REF):
        sys.path.insert(0, REF)
        try:
            from src.utils.eval_utils import ransac_PnP as ransac_pnp   # the reference function itself
        except Exception as e:  # noqa: BLE001
            print(f"reference eval_utils not importable ({e}); calling cv2 with its arguments")
    out, meta = {}, {"cv2": cv2.__version__, "numpy": np.__version__, "via_reference_function": ransac_pnp is not None, "cases": {}}
    for name, (n, outl, noise, seed) in CASES.items():
        p = synthetic.make_pnp_problem(n, outl, noise, seed)
        cv2.setRNGSeed(0)
        if ransac_pnp is not None:
            pose, _, inl = ransac_pnp(p["K"], p["pts_2d"], p["pts_3d"].copy(), scale=1000)
        else:
            dist = np.zeros((8, 1), np.float64)
            _, rvec, tvec, inl = cv2.solvePnPRansac(np.ascontiguousarray(p["pts_3d"].astype(np.float64)) * 1000,
                                                    np.ascontiguousarray(p["pts_2d"].astype(np.float64)), p["K"].astype(np.float64), dist,
                                                    reprojectionError=5, iterationsCount=10000, flags=cv2.SOLVEPNP_EPNP)
            pose = np.concatenate([cv2.Rodrigues(rvec)[0], tvec / 1000], axis=-1)
        inl = np.zeros((0, 1), np.int32) if inl is None or len(inl) == 0 else np.asarray(inl, np.int32)
        out[f"{name}_pose"], out[f"{name}_inliers"] = np.asarray(pose, np.float64), inl
        m = min(12, n)
        ok, rv, tv = cv2.solvePnP(p["pts_3d"][:m].astype(np.float64), p["pts_2d_exact"][:m].astype(np.float64) if "pts_2d_exact" in p
                                  else p["pts_2d"][:m].astype(np.float64), p["K"].astype(np.float64), np.zeros((8, 1)), flags=cv2.SOLVEPNP_EPNP)
        out[f"{name}_epnp12"] = np.concatenate([cv2.Rodrigues(rv)[0], tv], axis=-1)
        meta["cases"][name] = {"n": n, "outliers": outl, "noise": noise, "seed": seed, "inliers": int(len(inl))}
        print(name, "inliers", len(inl))
    np.savez_compressed(os.path.join(HERE, "pnp_cv2.npz"), **out)
    with open(os.path.join(HERE, "pnp_cv2_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
