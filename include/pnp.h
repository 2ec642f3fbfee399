/*
 * C ABI of the MI355X (gfx950) RANSAC-EPnP pose solver -- the step immediately after the GATsSPG
 * matcher in the OnePose inference loop.
 *
 * Replaces  src/utils/eval_utils.py:18-42  (ransac_PnP):
 *     cv2.solvePnPRansac(pts_3d * scale, pts_2d, K, dist = 0, reprojectionError = 5,
 *                        iterationsCount = 10000, flags = cv2.SOLVEPNP_EPNP)  +  cv2.Rodrigues
 * The algorithm itself lives in OpenCV (not vendored by the reference, not installed here): the
 * kernels restate its published form -- EPnP (Lepetit et al., IJCV 2009; OpenCV calib3d/epnp.cpp)
 * inside OpenCV's RANSAC scheme (minimal sets of 5, squared reprojection error <= threshold^2, most
 * inliers wins, final EPnP over the inliers) -- in fp64 like the reference's float64 call.  Two
 * documented differences: sample indices come from a counter-based hash (cv::RNG cannot be
 * reproduced without OpenCV) and every one of the `iterations` hypotheses is evaluated (OpenCV stops
 * at its adaptive confidence bound: a subset of this search).  See oracle/pnp_oracle.py.
 *
 * Conventions as in gatsspg.h: device pointers, caller-provided workspace, work enqueued on
 * `stream`, no allocation, no synchronisation, 0 = OK / non-zero = error + pnp_last_error().
 */
#ifndef ONEPOSE_AMD_PNP_H
#define ONEPOSE_AMD_PNP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* pnp_stream_t; /* hipStream_t */

int pnp_version(void);
const char* pnp_last_error(void);

size_t pnp_workspace_bytes(int n, int iterations);

/* pts_3d [n][3], pts_2d [n][2]: fp32 device arrays (the matcher's matched keypoints, inference.py:151-152);
 * K: 9 doubles on the HOST, row-major intrinsics; scale multiplies pts_3d before solving and divides t after
 * (eval_utils.py:27,35: OnePose passes 1000);
 * pose  [12] doubles (device): row-major 3x4 [R | t]; identity if the solve fails (:40-42);
 * inlier_mask [n] int32 (device): 1 for the inliers of the best hypothesis;
 * info  [4] int32 (device): {ok, number of inliers, index of the best hypothesis, its inlier count}. */
int pnp_ransac_epnp(const float* pts_3d, const float* pts_2d, const double* K_host, double scale, int n,
                    double reproj_error, int iterations, uint64_t seed, double* pose, int32_t* inlier_mask,
                    int32_t* info, void* workspace, size_t workspace_bytes, pnp_stream_t stream);

/* The same solve straight from the matcher's outputs, with the selection of inference.py:148-152 done on the device
 * (valid = matches0 > -1; mkpts2d = kpts2d[valid]; mkpts3d = kpts3d[matches0[valid]]) so that the number of
 * correspondences never has to reach the host:
 *   kpts2d [n1][2] fp32 (extractor keypoints), kpts3d [N3][3] fp32 (database points), matches0 [n1] int64 (-1 = unmatched);
 *   inlier_mask [n1] int32 indexed by QUERY KEYPOINT; info as above (ok = 0 and identity pose when fewer than 5 matches).
 * pnp_workspace_bytes(n1, iterations) sizes the workspace. */
int pnp_ransac_epnp_matches(const float* kpts2d, const float* kpts3d, const int64_t* matches0, int n1, const double* K_host,
                            double scale, double reproj_error, int iterations, uint64_t seed, double* pose,
                            int32_t* inlier_mask, int32_t* info, void* workspace, size_t workspace_bytes, pnp_stream_t stream);

/* EPnP alone over all n >= 4 correspondences (cv2.solvePnP(..., flags=SOLVEPNP_EPNP)); stage tests. */
int pnp_epnp(const float* pts_3d, const float* pts_2d, const double* K_host, double scale, int n, double* pose,
             void* workspace, size_t workspace_bytes, pnp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
