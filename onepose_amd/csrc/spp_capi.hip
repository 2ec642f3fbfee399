// C ABI of libspp_hip.so (declared in include/superpoint.h).  Thin: argument checks, workspace carve-up,
// kernel enqueue on the caller's stream.  No allocation, no synchronisation.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../../include/superpoint.h"
#include "spp_common.h"

using namespace spp;

namespace {
thread_local char g_err[512] = "";

int fail(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return -1;
}

int check_dims(int b, int H, int W) {
    if (b < 1) return fail("batch must be >= 1 (got %d)", b);
    if (H < 8 || W < 8)   // three floor-mode 2x2 poolings must leave at least one 8x8 cell
        return fail("H and W must be >= 8; got %dx%d", H, W);
    const long long cols = (long long)b * (((long long)(H + 2) * (W + 2) + 127) / 128 * 128);
    if (cols * 32 * 4 >= (1ll << 32)) return fail("problem too large: b*(H+2)*(W+2) = %lld columns", cols);
    return 0;
}

int check_ws(const void* ws, size_t ws_bytes, int b, int H, int W, Workspace& w) {
    if (int e = check_dims(b, H, W)) return e;
    if (!ws) return fail("workspace pointer is null");
    if (reinterpret_cast<uintptr_t>(ws) & 15) return fail("workspace must be 16-byte aligned");
    w = carve_workspace(const_cast<void*>(ws), b, H, W);
    if (ws_bytes < w.bytes) return fail("workspace too small: %zu < %zu bytes", ws_bytes, w.bytes);
    return 0;
}

int check_detect(int nms_radius, int max_keypoints, int remove_borders, int capacity, DetectParams& dp, float thr, int align) {
    if (nms_radius < 0 || nms_radius > MAX_R) return fail("nms_radius must be in [0, %d] (got %d)", MAX_R, nms_radius);
    if (max_keypoints == 0 || max_keypoints < -1)                       // superpoint.py:135-137
        return fail("\"max_keypoints\" must be positive or \"-1\"");
    if (remove_borders < 0) return fail("remove_borders must be >= 0");
    if (capacity < 1) return fail("capacity must be >= 1");
    if (max_keypoints > capacity) return fail("capacity (%d) is smaller than max_keypoints (%d)", capacity, max_keypoints);
    dp.nms_radius = nms_radius; dp.max_keypoints = max_keypoints; dp.remove_borders = remove_borders;
    dp.align_corners = align ? 1 : 0; dp.capacity = capacity; dp.threshold = thr;
    return 0;
}

int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("%s: %s", what, hipGetErrorString(e));
    return 0;
}

DescView padded_view(const Workspace& w) {
    DescView v;
    v.p = w.dd; v.cstride = 1; v.istride = (size_t)w.L4.ld * DD; v.rstride = w.L4.Wp * DD; v.xstride = DD; v.origin = (w.L4.Wp + 1) * DD;   // convDb writes [position][256]
    return v;
}

int check_flags(int flags, Workspace& w) {
    if (flags & ~SPP_FLAG_PREC_FP16X4) return fail("unknown bits in flags (0x%x)", flags);
    w.prec = (flags & SPP_FLAG_PREC_FP16X4) ? 4 : 0;
    return 0;
}

int forward_impl(const float* packed, const float* image, int b, int H, int W, int nms_radius, float thr, int max_keypoints,
                 int remove_borders, int align_corners, int capacity, float* keypoints, float* scores, float* descriptors,
                 int32_t* counts, void* ws, size_t ws_bytes, void* stream, int flags, ProfileHook* hk) {
    Workspace w;
    if (int e = check_ws(ws, ws_bytes, b, H, W, w)) return e;
    if (int e = check_flags(flags, w)) return e;
    DetectParams dp;
    if (int e = check_detect(nms_radius, max_keypoints, remove_borders, capacity, dp, thr, align_corners)) return e;
    if (!packed || !image || !keypoints || !scores || !descriptors || !counts) return fail("null argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    launch_dense(packed, image, w, s, hk);
    launch_score_map(w, w.score, s, hk);
    launch_detect(w.score, padded_view(w), w, dp, keypoints, scores, descriptors, counts, nullptr, s, hk);
    return check_launch("spp_forward");
}
}  // namespace

extern "C" {

int spp_version(void) { return 2; }   // 2: `flags` argument (convolution arithmetic), fp16 weight planes in the packed blob
const char* spp_last_error(void) { return g_err; }

size_t spp_packed_weights_bytes(void) { return PACKED_BYTES; }

int spp_pack_weights(const spp_raw_weights* raw, float* packed, spp_stream_t stream) {
    if (!raw || !packed) return fail("null argument");
    for (int i = 0; i < SPP_NUM_LAYERS; ++i)
        if (!raw->weight[i] || !raw->bias[i]) return fail("raw weights: layer %d has a null pointer", i);
    launch_pack_weights(raw, packed, reinterpret_cast<hipStream_t>(stream));
    return check_launch("spp_pack_weights");
}

size_t spp_workspace_bytes(int b, int H, int W) {
    if (check_dims(b, H, W)) return 0;
    return carve_workspace(nullptr, b, H, W).bytes;
}

int spp_dense(const float* packed, const float* image, int b, int H, int W, float* score_map, float* dense_desc, void* workspace,
              size_t workspace_bytes, spp_stream_t stream, int flags) {
    Workspace w;
    if (int e = check_ws(workspace, workspace_bytes, b, H, W, w)) return e;
    if (int e = check_flags(flags, w)) return e;
    if (!packed || !image || !score_map || !dense_desc) return fail("null argument");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    launch_dense(packed, image, w, s, nullptr);
    launch_score_map(w, score_map, s, nullptr);
    launch_export_dense(w, dense_desc, s);
    return check_launch("spp_dense");
}

int spp_detect(const float* score_map, const float* dense_desc, int b, int H, int W, int nms_radius, float keypoint_threshold,
               int max_keypoints, int remove_borders, int align_corners, int capacity, float* keypoints, float* scores,
               float* descriptors, int32_t* counts, float* nms_out, void* workspace, size_t workspace_bytes,
               spp_stream_t stream) {
    Workspace w;
    if (int e = check_ws(workspace, workspace_bytes, b, H, W, w)) return e;
    DetectParams dp;
    if (int e = check_detect(nms_radius, max_keypoints, remove_borders, capacity, dp, keypoint_threshold, align_corners)) return e;
    if (!score_map || !dense_desc || !keypoints || !scores || !descriptors || !counts) return fail("null argument");
    DescView v;
    const int Hc = H / 8, Wc = W / 8;   // floor: what three MaxPool2d(2, 2) leave
    v.p = dense_desc; v.cstride = (size_t)Hc * Wc; v.istride = (size_t)DD * Hc * Wc; v.rstride = Wc; v.xstride = 1; v.origin = 0;
    launch_detect(score_map, v, w, dp, keypoints, scores, descriptors, counts, nms_out, reinterpret_cast<hipStream_t>(stream),
                  nullptr);
    return check_launch("spp_detect");
}

int spp_forward(const float* packed, const float* image, int b, int H, int W, int nms_radius, float keypoint_threshold,
                int max_keypoints, int remove_borders, int align_corners, int capacity, float* keypoints, float* scores,
                float* descriptors, int32_t* counts, void* workspace, size_t workspace_bytes, spp_stream_t stream, int flags) {
    return forward_impl(packed, image, b, H, W, nms_radius, keypoint_threshold, max_keypoints, remove_borders, align_corners,
                        capacity, keypoints, scores, descriptors, counts, workspace, workspace_bytes, stream, flags, nullptr);
}

int spp_forward_profiled(const float* packed, const float* image, int b, int H, int W, int nms_radius, float keypoint_threshold,
                         int max_keypoints, int remove_borders, int align_corners, int capacity, float* keypoints,
                         float* scores, float* descriptors, int32_t* counts, void* workspace, size_t workspace_bytes,
                         spp_stream_t stream, int flags, int kernel_id, int occurrence, spp_event_t ev_start, spp_event_t ev_stop) {
    if (kernel_id < 0 || kernel_id >= KID_COUNT) return fail("kernel_id out of range");
    if (!ev_start || !ev_stop) return fail("null event");
    ProfileHook hk;
    memset(&hk, 0, sizeof(hk));
    hk.kernel_id = kernel_id; hk.occurrence = occurrence;
    hk.start = reinterpret_cast<hipEvent_t>(ev_start); hk.stop = reinterpret_cast<hipEvent_t>(ev_stop);
    const int rc = forward_impl(packed, image, b, H, W, nms_radius, keypoint_threshold, max_keypoints, remove_borders, align_corners,
                                capacity, keypoints, scores, descriptors, counts, workspace, workspace_bytes, stream, flags, &hk);
    if (rc != 0) return rc;
    // a kernel id that this configuration never launches (e.g. conv1a under SPP_FLAG_PREC_FP16X4 with even H: it is recomputed inside
    // conv1b's fused kernel) would leave the events unrecorded: say so instead of returning a bracket around nothing
    if (hk.seen[kernel_id] <= occurrence)
        return fail("kernel %d was launched %d times in this configuration, occurrence %d never ran", kernel_id, hk.seen[kernel_id], occurrence);
    return 0;
}

}  // extern "C"
