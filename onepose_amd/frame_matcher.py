"""The per-frame part of the reference inference loop (inference.py:132-152) with every hand-off in HBM.

    extractor(image) -> pack_data -> matcher(inp_data) -> valid matches -> (mkpts2d, mkpts3d, mconf)

The reference moves the detector output to the host and back (``.cpu().numpy()`` at :141, ``torch.Tensor(..).cuda()`` in
``pack_data`` :80-94) and re-uploads the 3D database every frame; here the database dict (``database_io.load_object_database``)
stays resident, the descriptors go from the extractor's output buffer straight into the matcher, and only the final
variable-length correspondence lists are gathered; ``solve_pose`` feeds them, still on the GPU, to the RANSAC-EPnP solver
of ``onepose_amd.pnp`` (``ransac_PnP``, :155).
"""
from __future__ import annotations

import torch


class FrameMatcher:
    """extractor: onepose_amd.SuperPoint; matcher: onepose_amd.GATsSuperGlue; database: the dict of
    ``load_object_database`` (keypoints3d [1,N,3], descriptors3d_db [1,256,N], descriptors2d_db [1,256,N*L])."""

    def __init__(self, extractor, matcher, database, cache_database=True):
        self.extractor, self.matcher, self.db = extractor, matcher, database
        for k in ("keypoints3d", "descriptors3d_db", "descriptors2d_db"):
            if not database[k].is_cuda:
                raise RuntimeError(f"database['{k}'] must live on the GPU (there is no CPU fallback)")
        # the query-independent part of the first three GNN layers, computed once per object
        self.db_cache = matcher.prepare_database(database) if cache_database else None

    @torch.no_grad()
    def __call__(self, image):
        """image [1,1,H,W] on the GPU -> dict(mkpts2d [m,2], mkpts3d [m,3], mconf [m], keypoints2d, matches0) (inference.py:140-152)."""
        det = self.extractor(image)                                           # :140
        kpts2d, desc2d = det["keypoints"][0], det["descriptors"][0]
        inp = {"keypoints2d": kpts2d[None], "keypoints3d": self.db["keypoints3d"],            # pack_data :80-94
               "descriptors2d_query": desc2d[None].contiguous(), "descriptors3d_db": self.db["descriptors3d_db"],
               "descriptors2d_db": self.db["descriptors2d_db"]}
        pred, _ = self.matcher(inp, database=self.db_cache) if self.db_cache is not None else self.matcher(inp)   # :146
        matches, conf = pred["matches0"], pred["matching_scores0"]           # :147-151
        valid = matches > -1
        return {"mkpts2d": kpts2d[valid], "mkpts3d": self.db["keypoints3d"][0][matches[valid]], "mconf": conf[valid],
                "keypoints2d": kpts2d, "matches0": matches}

    @torch.no_grad()
    def solve_pose_device(self, image, K_crop, scale=1000, seed=0):
        """image -> (pose [3,4] float64, inlier mask per query keypoint, info [4], detections) all left on the GPU: the valid
        matches are selected on the device (``pnp_ransac_epnp_matches``), nothing is synchronised after the extractor."""
        from . import pnp
        det = self.extractor(image)
        kpts2d = det["keypoints"][0]
        inp = {"keypoints2d": kpts2d[None], "keypoints3d": self.db["keypoints3d"],
               "descriptors2d_query": det["descriptors"][0][None].contiguous(), "descriptors3d_db": self.db["descriptors3d_db"],
               "descriptors2d_db": self.db["descriptors2d_db"]}
        pred, _ = self.matcher(inp, database=self.db_cache) if self.db_cache is not None else self.matcher(inp)
        pose, mask, info = pnp.ransac_pnp_from_matches(K_crop, kpts2d, self.db["keypoints3d"][0], pred["matches0"], scale=scale, seed=seed)
        return pose, mask, info, det

    @torch.no_grad()
    def solve_pose(self, image, K_crop, scale=1000, seed=0):
        """image -> (pose_pred [3,4], pose_pred_homo [4,4], inliers [m,1]) like inference.py:140-155; numpy outputs, identity
        when fewer than 5 matches survive or the solve fails (eval_utils.py:40-42)."""
        from . import pnp
        out = self(image)
        return pnp.ransac_PnP(K_crop, out["mkpts2d"], out["mkpts3d"], scale=scale, seed=seed)
