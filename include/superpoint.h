/*
 * C ABI of the MI355X (gfx950) SuperPoint extractor -- the step immediately before the GATsSPG
 * matcher in the OnePose inference loop (reference inference.py:140, `extractor_model(inp)`).
 *
 * Replaces, on the GPU, the forward of the reference module
 *   src/models/extractors/SuperPoint/superpoint.py:140-197  (SuperPoint.forward)
 * including its helpers simple_nms (:47-62), remove_borders (:65-70), top_k_keypoints (:73-78)
 * and sample_descriptors (:81-94).  The reference has no FFI for this path; INTEGRATION.md shows
 * the ctypes binding a maintainer would add.
 *
 * Conventions (same as gatsspg.h): every pointer is a DEVICE pointer to fp32 / int32 data unless
 * stated otherwise, 16-byte aligned, contiguous; all work is enqueued on `stream`; the library
 * never allocates and never synchronises; functions return 0 on success or a negative code, with
 * text available from spp_last_error().
 *
 * Shapes: image [b][1][H][W] grayscale in [0,1], H, W >= 8.  Hc = H/8, Wc = W/8 (integer division:
 * three floor-mode 2x2 poolings, :145-151); the score map and every keypoint live in the
 * Hs x Ws = (8*Hc) x (8*Wc) top-left part of the image (:160-162), exactly as in the reference.
 * descriptor_dim is 256 (the reference default, :105, and what the matcher consumes).
 */
#ifndef ONEPOSE_AMD_SUPERPOINT_H
#define ONEPOSE_AMD_SUPERPOINT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* spp_stream_t; /* hipStream_t */
typedef struct ihipEvent_t* spp_event_t;   /* hipEvent_t */

#define SPP_DESC_DIM 256
#define SPP_NUM_LAYERS 12
#define SPP_MAX_NMS_RADIUS 6

/* Raw parameters in the reference state_dict layout (superpoint.py:119-133), forward order:
 * conv1a conv1b conv2a conv2b conv3a conv3b conv4a conv4b convPa convPb convDa convDb;
 * weight[i] is [out][in][k][k], bias[i] is [out].  HOST struct holding DEVICE pointers. */
typedef struct spp_raw_weights {
    const float* weight[SPP_NUM_LAYERS];
    const float* bias[SPP_NUM_LAYERS];
} spp_raw_weights;

/* Arithmetic of the GEMM convolutions (conv1b .. convDb), selected PER CALL by the `flags` argument of spp_dense / spp_forward
 * (the library never reads the environment):
 *   0 (default): exact fp32 MFMA (v_mfma_f32_32x32x2_f32) -- the reference's arithmetic;
 *   SPP_FLAG_PREC_FP16X4: every fp32 operand as two fp16 terms (RNE, saturating at +-65504), all four products on
 *     v_mfma_f32_32x32x16_f16 with fp32 accumulation -- fp32-class results at a quarter of the matrix-pipe time (the matcher's
 *     GATSSPG_FLAG_PREC_FP16X4; include/gatsspg.h).  conv1a: fp32 VALU in the default mode and for odd H; under
 *     SPP_FLAG_PREC_FP16X4 with even H it is RECOMPUTED inside conv1b's fused kernel (conv1ab_pool_f16_kernel) on the f16 MFMA from
 *     fp16-split pixels, weights and bias -- the standalone conv1a kernel is then never launched (spp_forward_profiled with its
 *     kernel id returns an error).  The detector head's softmax, NMS / top-k and the descriptor sampling are fp32 in both modes.
 *     The extractor's fp16 terms are UNSCALED (unlike the matcher's since its ABI 400, include/gatsspg.h): x1 = RNE_fp16(x), x2 =
 *     RNE_fp16(x - x1), saturating at +-65504; an operand below 2^-3 in magnitude has a subnormal second term, i.e. an absolute error of
 *     up to 2^-25 per operand instead of the 2^-23 relative one (He-initialised / SuperPoint weights are mostly below 0.06: the products
 *     then carry ~2^-25 / |w| relative error, 1e-6 at |w| = 0.03).  Measured against the fp32 MFMA on the goldens: score map and
 *     descriptors within 1e-5 like the fp32 path (tests/test_spp_hip_parity.py); it is a reduced-precision mode with that measured
 *     tolerance, not a bit-for-bit fp32 replacement.  Unknown bits are refused. */
#define SPP_FLAG_PREC_FP16X4 0x800

int spp_version(void);
const char* spp_last_error(void);

/* One-time re-layout of the weights ([out][tap][in] K-order for the implicit-GEMM kernels, the two
 * 3x3 head convolutions stacked into one 512-row operator, rows padded to the tile height), followed by the fp16 hi / lo
 * planes of every GEMM convolution for SPP_FLAG_PREC_FP16X4 calls. */
size_t spp_packed_weights_bytes(void);
int spp_pack_weights(const spp_raw_weights* raw, float* packed, spp_stream_t stream);

size_t spp_workspace_bytes(int b, int H, int W);

/* Dense stages (:142-162,183-184): image -> score_map [b][Hs][Ws] (cell softmax, dustbin dropped, 8x8
 * shuffle; BEFORE nms) and dense_desc [b][256][Hc][Wc] (convDb output, NOT normalised). */
int spp_dense(const float* packed, const float* image, int b, int H, int W, float* score_map, float* dense_desc,
              void* workspace, size_t workspace_bytes, spp_stream_t stream, int flags);

/* Discrete stages (:163-195) from given dense tensors: NMS, threshold, border removal, top-k,
 * (h,w)->(x,y), descriptor normalisation + bilinear sampling + normalisation.
 *   max_keypoints  >= 0: keep the k highest scores, ordered by descending score (ties: lower row-major
 *                  pixel index first -- torch.topk leaves tie order unspecified); -1: keep all, row-major order.
 *   capacity       slots per image in the outputs; with max_keypoints >= 0 it must be >= max_keypoints.
 *   keypoints      [b][capacity][2] (x, y) as float;  scores [b][capacity];
 *   descriptors    [b][256][capacity];
 *   counts         int32 [b][2]: {keypoints written, candidates after threshold + border}.  With
 *                  max_keypoints == -1 and candidates > capacity only the first `capacity` (row-major) are
 *                  written: the caller must compare the two numbers.
 *   align_corners  the reference takes it from `int(torch.__version__[2]) > 2` (:87): 1 on the torch 1.x
 *                  builds OnePose pins, 0 is what the same line yields on torch >= 1.10 / 2.x. */
int spp_detect(const float* score_map, const float* dense_desc, int b, int H, int W, int nms_radius,
               float keypoint_threshold, int max_keypoints, int remove_borders, int align_corners, int capacity,
               float* keypoints, float* scores, float* descriptors, int32_t* counts, float* nms_out /* optional [b][Hs][Ws] */,
               void* workspace, size_t workspace_bytes, spp_stream_t stream);

/* SuperPoint.forward: spp_dense + spp_detect without the intermediate copies. */
int spp_forward(const float* packed, const float* image, int b, int H, int W, int nms_radius, float keypoint_threshold,
                int max_keypoints, int remove_borders, int align_corners, int capacity, float* keypoints,
                float* scores, float* descriptors, int32_t* counts, void* workspace, size_t workspace_bytes,
                spp_stream_t stream, int flags);

/* spp_forward with a HIP-event bracket around the `occurrence`-th launch of kernel `kernel_id`
 * (ids: onepose_amd/_native_spp.py::KERNEL_IDS) -- bench.py's per-kernel roofline timing. */
int spp_forward_profiled(const float* packed, const float* image, int b, int H, int W, int nms_radius,
                         float keypoint_threshold, int max_keypoints, int remove_borders, int align_corners,
                         int capacity, float* keypoints, float* scores, float* descriptors, int32_t* counts,
                         void* workspace, size_t workspace_bytes, spp_stream_t stream, int flags, int kernel_id, int occurrence,
                         spp_event_t ev_start, spp_event_t ev_stop);

#ifdef __cplusplus
}
#endif
#endif
