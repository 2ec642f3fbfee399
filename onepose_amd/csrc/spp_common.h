// SuperPoint extractor: activation layout in HBM, packed-weights blob, workspace carve-up.
// See DESIGN.md "SuperPoint extractor".
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace spp {

constexpr int DD = 256;          // descriptor_dim
constexpr int NMS_E = 64;        // NMS workgroup region (tile + 2 * 5 * radius halo)
constexpr int MAX_R = 6;

// Activation planes: channel-major [C][b * ld] fp32.  One image's plane is its (H+2) x (W+2) zero-padded
// picture flattened row-major (position q = (y+1)*Wp + (x+1) for pixel (y, x)), rounded up to a multiple of
// 128 columns.  A 3x3 tap (dy, dx) of a convolution is then the SAME matrix shifted by dy*Wp + dx columns,
// so the convolution is 9 accumulated GEMMs over shifted views with no im2col and no border tests; the
// pad positions of the output are written as zeros by the epilogue.
struct FeatLayout {
    int b, H, W, Wp, plane, ld, ldt;
};
__host__ __device__ inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
__host__ __device__ inline FeatLayout make_feat_layout(int b, int H, int W) {
    FeatLayout L;
    L.b = b; L.H = H; L.W = W; L.Wp = W + 2;
    L.plane = (H + 2) * (W + 2);
    L.ld = round_up(L.plane, 128);
    L.ldt = b * L.ld;
    return L;
}
// (is pixel, y, x) of padded position q of an image
__host__ __device__ inline bool feat_valid(const FeatLayout& L, int q, int& y, int& x) {
    y = q / L.Wp; x = q - y * L.Wp;
    return q < L.plane && y >= 1 && y <= L.H && x >= 1 && x <= L.W;
}
// floats of slack before the first / after the last channel row: shifted views reach Wp + 1 columns outside a
// flat tile, and the second row of a patch tile below an odd-height image reaches 2 * Wp + 1
__host__ __device__ inline size_t feat_guard(const FeatLayout& L) { return (size_t)round_up(2 * L.Wp + 8, 64); }

// ---- packed weights (floats) ---------------------------------------------------------------------------
// GEMM convolutions, forward order; weights [rows][taps * cin] with k = tap * cin + ci, tap = 3*(dy+1) + (dx+1)
struct ConvSpec { int cout, rows, cin, taps; };
constexpr int NGEMM = 10;
constexpr ConvSpec kConv[NGEMM] = {
    {64, 64, 64, 9},     // 0 conv1b
    {64, 64, 64, 9},     // 1 conv2a
    {64, 64, 64, 9},     // 2 conv2b
    {128, 128, 64, 9},   // 3 conv3a
    {128, 128, 128, 9},  // 4 conv3b
    {128, 128, 128, 9},  // 5 conv4a
    {128, 128, 128, 9},  // 6 conv4b
    {512, 512, 128, 9},  // 7 convPa (rows 0..255) stacked on convDa (rows 256..511)
    {65, 128, 256, 1},   // 8 convPb (rows padded with zeros)
    {256, 256, 256, 1},  // 9 convDb
};
constexpr size_t PW_C1A_W = 0;            // conv1a weights [64][9] (+ pad)
constexpr size_t PW_C1A_B = 64 * 9 + 64;  // conv1a bias [64]
constexpr size_t PW_GEMM0 = PW_C1A_B + 64;
constexpr size_t conv_w_off(int i) {
    size_t o = PW_GEMM0;
    for (int j = 0; j < i; ++j) o += (size_t)kConv[j].rows * kConv[j].cin * kConv[j].taps + kConv[j].rows;
    return o;
}
constexpr size_t conv_b_off(int i) { return conv_w_off(i) + (size_t)kConv[i].rows * kConv[i].cin * kConv[i].taps; }
constexpr size_t PW_TOTAL = conv_w_off(NGEMM);
// fp16 hi / lo planes of the GEMM convolutions' weights (SPP_FLAG_PREC_FP16X4): appended to the fp32 blob, in 16-bit elements from
// (unsigned short*)(packed + PW_TOTAL).  w = hi + lo with hi = RNE_fp16(w), lo = RNE_fp16(w - hi); per convolution the hi plane
// then the lo plane, each slab-major: element (m, k) of the [rows][K] operator at ((k / 32) * rows + m) * 32 + k % 32.
constexpr size_t conv_k(int i) { return (size_t)kConv[i].cin * kConv[i].taps; }
constexpr size_t conv_wp_off(int i) {
    size_t o = 0;
    for (int j = 0; j < i; ++j) o += 2 * (size_t)kConv[j].rows * conv_k(j);
    return o;
}
constexpr size_t PWP_TOTAL = conv_wp_off(NGEMM);                                  // 16-bit elements
constexpr size_t PACKED_BYTES = sizeof(float) * PW_TOTAL + sizeof(unsigned short) * PWP_TOTAL;

// ---- workspace ---------------------------------------------------------------------------------------------
struct Workspace {
    int prec;                        // 0: fp32 MFMA convolutions; 4: four-term split-fp16 (SPP_FLAG_PREC_FP16X4), from the call's flags
    FeatLayout L1, L2, L3, L4;       // full, 1/2, 1/4, 1/8 resolution
    float *a1, *b1, *a2, *b2, *a3, *b3, *c3, *a4, *b4, *hd, *lg, *dd;
    float *score, *nms, *invn;
    int *rowcnt, *rowoff, *ncand, *cand, *sel, *rank, *surv;
    float* cscore;
    unsigned* skey;
    size_t bytes;
};
inline size_t align_up(size_t x) { return (x + 255) & ~size_t(255); }

inline Workspace carve_workspace(void* base, int b, int H, int W) {
    Workspace w;
    w.prec = 0;
    w.L1 = make_feat_layout(b, H, W);
    w.L2 = make_feat_layout(b, H / 2, W / 2);
    w.L3 = make_feat_layout(b, H / 4, W / 4);
    w.L4 = make_feat_layout(b, H / 8, W / 8);
    char* p = static_cast<char*>(base);
    size_t off = 0;
    auto take = [&](size_t nbytes) { char* r = p ? p + off : nullptr; off += align_up(nbytes); return r; };
    auto feat = [&](int C, const FeatLayout& L) {
        take(sizeof(float) * feat_guard(L));    // front guard (the previous buffer's back guard is separate: simple, tiny)
        float* r = (float*)take(sizeof(float) * (size_t)C * L.ldt);
        take(sizeof(float) * feat_guard(L));
        return r;
    };
    w.a1 = feat(64, w.L1); w.b1 = feat(64, w.L1);
    w.a2 = feat(64, w.L2); w.b2 = feat(64, w.L2);
    w.a3 = feat(64, w.L3); w.b3 = feat(128, w.L3); w.c3 = feat(128, w.L3);
    w.a4 = feat(128, w.L4); w.b4 = feat(128, w.L4);
    w.hd = feat(512, w.L4); w.lg = feat(128, w.L4); w.dd = feat(256, w.L4);
    const size_t px = (size_t)b * H * W;
    w.score = (float*)take(sizeof(float) * px);
    w.nms = (float*)take(sizeof(float) * px);
    w.invn = (float*)take(sizeof(float) * (size_t)b * (H / 8) * (W / 8));
    w.rowcnt = (int*)take(sizeof(int) * (size_t)b * H);
    w.rowoff = (int*)take(sizeof(int) * (size_t)b * H);
    w.ncand = (int*)take(sizeof(int) * (size_t)b);
    w.cand = (int*)take(sizeof(int) * px);
    w.sel = (int*)take(sizeof(int) * px);
    w.rank = (int*)take(sizeof(int) * px);
    w.surv = (int*)take(sizeof(int) * px);
    w.cscore = (float*)take(sizeof(float) * px);
    w.skey = (unsigned*)take(sizeof(unsigned) * px);
    w.bytes = off;
    return w;
}

// Kernel ids of spp_forward_profiled
enum KernelId {
    KID_CONV1A = 0, KID_CONV1B = 1, KID_POOL = 2, KID_CONV2 = 3, KID_CONV3A = 4, KID_CONV3B = 5, KID_CONV4 = 6, KID_HEADS = 7,
    KID_CONVPB = 8, KID_CONVDB = 9, KID_SCORE = 10, KID_NMS = 11, KID_ROWCOUNT = 12, KID_SCAN = 13, KID_COMPACT = 14,
    KID_SELECT = 15, KID_CELLNORM = 16, KID_SAMPLE = 17, KID_RANK = 18, KID_SCATTER = 19, KID_COUNT = 20
};
struct ProfileHook {
    int kernel_id, occurrence;
    hipEvent_t start, stop;
    int seen[KID_COUNT];
};
inline bool hook_hit(ProfileHook* h, int kid) { return h && h->kernel_id == kid && h->seen[kid] == h->occurrence; }
inline void hook_before(ProfileHook* h, int kid, hipStream_t s) {
    if (hook_hit(h, kid)) (void)hipEventRecord(h->start, s);
}
inline void hook_after(ProfileHook* h, int kid, hipStream_t s) {
    if (hook_hit(h, kid)) (void)hipEventRecord(h->stop, s);
    if (h) h->seen[kid]++;
}
#define SPP_LAUNCH(hook, kid, stream, ...)   \
    do {                                     \
        spp::hook_before(hook, kid, stream); \
        hipLaunchKernelGGL(__VA_ARGS__);     \
        spp::hook_after(hook, kid, stream);  \
    } while (0)

struct DetectParams {
    int nms_radius, max_keypoints, remove_borders, align_corners, capacity;
    float threshold;
};

// spp_conv_kernels.hip
void launch_pack_weights(const void* raw_host, float* packed, hipStream_t s);
void launch_dense(const float* packed, const float* image, const Workspace& w, hipStream_t s, ProfileHook* hk);
void launch_export_dense(const Workspace& w, float* dense_desc, hipStream_t s);
// spp_detect_kernels.hip
void launch_score_map(const Workspace& w, float* score_map, hipStream_t s, ProfileHook* hk);
// dense descriptors are read through (ptr, channel stride, image stride, row stride, x stride, origin offset).  The extractor's own
// plane is POSITION-major ([padded position][256 channels], cstride 1, xstride 256: a keypoint's 256 channels of one bilinear tap
// are 1 KB contiguous); a caller's [b][256][Hc][Wc] tensor (spp_detect) is channel-major (xstride 1).
struct DescView { const float* p; size_t cstride, istride; int rstride, xstride, origin; };
void launch_detect(const float* score_map, DescView dv, const Workspace& w, const DetectParams& dp, float* keypoints,
                   float* scores, float* descriptors, int32_t* counts, float* nms_out, hipStream_t s, ProfileHook* hk);

}  // namespace spp
