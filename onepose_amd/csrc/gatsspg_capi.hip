// C ABI of libgatsspg_hip.so (declared in include/gatsspg.h).  Thin: argument checks, workspace
// carve-up, kernel enqueue on the caller's stream.  No allocation, no synchronisation.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../../include/gatsspg.h"
#include "gatsspg_launch.h"

using namespace gatsspg;

namespace {
thread_local char g_err[512] = "";

int fail(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}

int check_dims(int b, int n1, int n2, int num_leaf) {
    if (b < 1) return fail("batch must be >= 1 (got %d)", b);
    // the reference returns early for an empty side (GATs_SuperGlue.py:195) and InstanceNorm1d raises for
    // a single point (:126); both are handled by the host-side module, never enqueued.
    if (n1 < 2 || n2 < 2) return fail("n1 and n2 must be >= 2 (got n1=%d n2=%d)", n1, n2);
    if (num_leaf < 1 || num_leaf > 64) return fail("num_leaf must be in [1, 64] (got %d)", num_leaf);
    const long long ld = (long long)b * (round_up(n1, CP) + round_up(n2, CP));
    if (ld * 512 >= (1ll << 31)) return fail("problem too large: b*(n1p+n2p) = %lld columns", ld);
    if ((long long)n1 * n2 >= (1ll << 31)) return fail("n1*n2 too large");
    return 0;
}

int check_ws(const void* ws, size_t ws_bytes, int b, int n1, int n2, int num_leaf, Workspace& w, int flags = 0) {
    if (int e = check_dims(b, n1, n2, num_leaf)) return e;
    if (!ws) return fail("workspace pointer is null");
    if (reinterpret_cast<uintptr_t>(ws) & 15) return fail("workspace must be 16-byte aligned");
    constexpr int PREC_BITS = GATSSPG_FLAG_PREC_BF16X3 | GATSSPG_FLAG_PREC_BF16X6 | GATSSPG_FLAG_PREC_FP16X3 | GATSSPG_FLAG_PREC_FP16X4;
    if (flags & ~(GATSSPG_FLAG_INCLUDE_SELF | GATSSPG_FLAG_ADDITIONAL | GATSSPG_FLAG_WITH_LINEAR_TRANSFORM | PREC_BITS))
        return fail("unknown bits in flags (0x%x)", flags);
    const int pb = flags & PREC_BITS;
    if (pb & (pb - 1)) return fail("GATSSPG_FLAG_PREC_BF16X3, _BF16X6, _FP16X3 and _FP16X4 are exclusive");
    w = carve_workspace(const_cast<void*>(ws), b, n1, n2);
    w.prec = (flags & GATSSPG_FLAG_PREC_BF16X6) ? 2 : (flags & GATSSPG_FLAG_PREC_BF16X3) ? 1 : (flags & GATSSPG_FLAG_PREC_FP16X3) ? 3 : (flags & GATSSPG_FLAG_PREC_FP16X4) ? 4 : 0;
    if (ws_bytes < w.bytes) return fail("workspace too small: %zu < %zu bytes", ws_bytes, w.bytes);
    return 0;
}

// The dual softmax is evaluated without max-subtraction whenever that is safe: the scores are cosines / scale_factor, so
// exp() stays in range as long as 1 / scale_factor <= 80 (exp(80) = 5.5e34, row sums of 1e5 terms still fit fp32; the
// reference's 0.07 gives 14.3) and one pass over S yields both normalisers.  Smaller scale factors take the
// max-subtracting path (raw scores -> row/column maxima and shifted sums -> finalize), which has the full range of
// torch.softmax (GATs_SuperGlue.py:218).
int check_scale(float scale_factor) {
    if (!(scale_factor > 0.f)) return fail("scale_factor must be positive");
    return 0;
}
inline int softmax_shifted(float scale_factor) { return scale_factor < 0.0125f ? 1 : 0; }

int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("%s: %s", what, hipGetErrorString(e));
    return 0;
}

const float* attn_w(const float* packed, int layer) { return packed + PW_ATTN + (size_t)layer * AttnW::SIZE; }
const unsigned short* attn_wb(const float* packed, int layer) {
    return reinterpret_cast<const unsigned short*>(packed + PW_TOTAL) + (size_t)layer * AttnWB::SIZE;
}
const float* gats_w(const float* packed, int layer) { return packed + PW_GATS + (size_t)layer * GatsW::SIZE; }

// h3 / dq: fused state load (see launch_gats); only valid when gats_fuses_state_load() says so
void enqueue_gats(const float* packed, int layer, const float* desc2d_db, int num_leaf, int flags, const Workspace& w,
                  hipStream_t s, ProfileHook* hk = nullptr, const float* h3 = nullptr, const float* dq = nullptr,
                  const float* cached_logits = nullptr) {
    const float* g = gats_w(packed, layer);
    if (flags & GATSSPG_FLAG_WITH_LINEAR_TRANSFORM) {
        // pre-activation aggregate -> MSG (free between attention layers), then elu(W^T pre (+h))
        launch_gats(g + GatsW::U1, g + GatsW::U2, desc2d_db, num_leaf, flags, w.MSG, w, s, hk);
        const int add_h = (flags & GATSSPG_FLAG_INCLUDE_SELF) && (flags & GATSSPG_FLAG_ADDITIONAL);
        launch_gats_wlt(g + GatsW::W, w.MSG, w, add_h, s, hk);
    } else {
        launch_gats(g + GatsW::U1, g + GatsW::U2, desc2d_db, num_leaf, flags, w.Z, w, s, hk, h3, dq, cached_logits);
    }
}

void enqueue_attn(const float* packed, int layer, int kind, const Workspace& w, hipStream_t s, ProfileHook* hk = nullptr) {
    const float* a = attn_w(packed, layer);
    const unsigned short* ab = attn_wb(packed, layer);
    launch_qkv_kv(a + AttnW::WQKV, a + AttnW::BQKV, ab, w, s, hk);
    launch_kv_final(a + AttnW::W0, w, kind == GATSSPG_LAYER_CROSS, nullptr, s, hk);
    launch_mlp(a + AttnW::W0, a + AttnW::B0, a + AttnW::W3, a + AttnW::B3, ab, w, s, hk);
}

// ---- database cache: everything of the first three GNN layers that depends only on the
//      per-object 3D database.  Layout (floats): Y2 [b][256][n2] | QY [b][256][n2] | kvY [b][4][KVP] |
//      LL [3][b][tiles][32] (leaf logits of GATs layers 1..3; layer 0 is inside Y2)
struct DbCache {
    float *Y2, *QY, *kvY, *LL;
    size_t ll_layer;   // floats per layer of LL
    size_t bytes;
};
DbCache carve_cache(void* base, int b, int n2) {
    DbCache c;
    float* p = static_cast<float*>(base);
    const size_t plane = (size_t)b * D * n2;
    c.Y2 = p;
    c.QY = p ? p + plane : nullptr;
    c.kvY = p ? p + 2 * plane : nullptr;
    c.LL = p ? p + 2 * plane + (size_t)b * H * KVP : nullptr;
    c.ll_layer = gats_leaf_logit_floats(b, n2);
    c.bytes = sizeof(float) * (2 * plane + (size_t)b * H * KVP + 3 * c.ll_layer);
    return c;
}
Workspace windowed(const Workspace& w, int side) {
    Workspace v = w;
    v.L = side_window(w.L, side);
    return v;
}

int forward_impl(const float* packed, const float* desc2d_query, const float* desc3d_db, const float* desc2d_db, int b,
                 int n1, int n2, int num_leaf, int flags, float scale_factor, float match_threshold, float* conf,
                 int64_t* matches0, int64_t* matches1, float* mscores0, float* mscores1, void* ws, size_t ws_bytes,
                 void* stream, ProfileHook* hk) {
    Workspace w;
    if (int e = check_ws(ws, ws_bytes, b, n1, n2, num_leaf, w, flags)) return e;
    if (!packed || !desc2d_query || !desc3d_db || !desc2d_db) return fail("null input pointer");
    if (!conf || !matches0 || !matches1 || !mscores0 || !mscores1) return fail("null output pointer");
    if (int e = check_scale(scale_factor)) return e;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // the state load is fused into the first GATs launch where that kernel supports it (num_leaf == 8, no linear transform)
    const bool fused_load = gats_fuses_state_load(num_leaf, flags, w);
    if (!fused_load) launch_load_state(desc2d_query, desc3d_db, w, s, hk);
    for (int t = 0; t < 4; ++t) {  // ['GATs', 'self', 'cross'] * 4, GATs_SuperGlue.py:162
        if (t == 0 && fused_load) enqueue_gats(packed, t, desc2d_db, num_leaf, flags, w, s, hk, desc3d_db, desc2d_query);
        else enqueue_gats(packed, t, desc2d_db, num_leaf, flags, w, s, hk);
        enqueue_attn(packed, 2 * t, GATSSPG_LAYER_SELF, w, s, hk);
        enqueue_attn(packed, 2 * t + 1, GATSSPG_LAYER_CROSS, w, s, hk);
    }
    const int shifted = softmax_shifted(scale_factor);
    launch_final_proj_norm(packed + PW_FINAL_W, packed + PW_FINAL_B, w, s, hk);
    launch_score_exp(w, conf, scale_factor, shifted, s, hk);
    launch_dual_softmax_match(w, conf, scale_factor, shifted, match_threshold, matches0, matches1, mscores0, mscores1, s, hk);
    return check_launch("forward");
}
}  // namespace

extern "C" {

int gatsspg_version(void) { return 411; }   // 411: same layouts as 410; the bound data in the KV partials / database cache changed meaning (slots 0..3: per-wave largest key sum of the tile, summed by kv_final; 4..7: max |V|): a cache written by a 410 library must be re-prepared; 410: message-operator scale of the fp16 modes from a data bound (KV partials carry operand maxima: packed-weights, workspace and database-cache layouts changed); 400: split-16-bit GEMMs on the LDS-DMA loop, power-of-two operand scales of the fp16 modes (packed-weights and workspace layouts changed)
const char* gatsspg_last_error(void) { return g_err; }

size_t gatsspg_packed_weights_bytes(void) { return PACKED_BYTES; }

size_t gatsspg_workspace_bytes(int b, int n1, int n2, int num_leaf) {
    if (check_dims(b, n1, n2, num_leaf)) return 0;
    return carve_workspace(nullptr, b, n1, n2).bytes;
}

int gatsspg_pack_weights(const gatsspg_raw_weights* raw, float* packed, void* stream) {
    if (!raw || !packed) return fail("null argument");
    const void* const* p = reinterpret_cast<const void* const*>(raw);
    for (size_t i = 0; i < sizeof(gatsspg_raw_weights) / sizeof(void*); ++i)
        if (!p[i]) return fail("raw weight pointer #%zu is null", i);
    if (reinterpret_cast<uintptr_t>(packed) & 15) return fail("packed-weights buffer must be 16-byte aligned");
    launch_pack_weights(raw, packed, static_cast<hipStream_t>(stream));
    launch_split_weights(packed, reinterpret_cast<unsigned short*>(packed + PW_TOTAL), static_cast<hipStream_t>(stream));
    return check_launch("pack_weights");
}

int gatsspg_load_state(const float* dq, const float* d3, int b, int n1, int n2, int num_leaf, void* ws, size_t ws_bytes,
                       void* stream) {
    Workspace w;
    if (int e = check_ws(ws, ws_bytes, b, n1, n2, num_leaf, w)) return e;
    if (!dq || !d3) return fail("null descriptor pointer");
    launch_load_state(dq, d3, w, static_cast<hipStream_t>(stream));
    return check_launch("load_state");
}

int gatsspg_store_state(int which, float* out2d, float* out3d, int b, int n1, int n2, int num_leaf, void* ws,
                        size_t ws_bytes, void* stream) {
    Workspace w;
    if (int e = check_ws(ws, ws_bytes, b, n1, n2, num_leaf, w)) return e;
    if (!out2d || !out3d) return fail("null output pointer");
    if (which != 0 && which != 1) return fail("which must be 0 (state) or 1 (normalised final descriptors)");
    launch_store_state(which == 0 ? w.Z : w.MD, out2d, out3d, w, static_cast<hipStream_t>(stream));
    return check_launch("store_state");
}

int gatsspg_gats_layer(const float* packed, int layer, const float* desc2d_db, int b, int n1, int n2, int num_leaf,
                       int flags, void* ws, size_t ws_bytes, void* stream) {
    Workspace w;
    if (int e = check_ws(ws, ws_bytes, b, n1, n2, num_leaf, w, flags)) return e;
    if (!packed || !desc2d_db) return fail("null argument");
    if (layer < 0 || layer >= GATSSPG_NUM_GATS_LAYERS) return fail("GATs layer index %d out of range", layer);
    enqueue_gats(packed, layer, desc2d_db, num_leaf, flags, w, static_cast<hipStream_t>(stream));
    return check_launch("gats_layer");
}

int gatsspg_attn_layer(const float* packed, int layer, int kind, int b, int n1, int n2, int num_leaf, int flags, void* ws,
                       size_t ws_bytes, void* stream) {
    Workspace w;
    if (int e = check_ws(ws, ws_bytes, b, n1, n2, num_leaf, w, flags)) return e;
    if (!packed) return fail("null argument");
    if (layer < 0 || layer >= GATSSPG_NUM_ATTN_LAYERS) return fail("attention layer index %d out of range", layer);
    if (kind != GATSSPG_LAYER_SELF && kind != GATSSPG_LAYER_CROSS) return fail("kind must be SELF or CROSS");
    enqueue_attn(packed, layer, kind, w, static_cast<hipStream_t>(stream));
    return check_launch("attn_layer");
}

int gatsspg_final_proj_norm(const float* packed, int b, int n1, int n2, int num_leaf, void* ws, size_t ws_bytes,
                            void* stream) {
    Workspace w;
    if (int e = check_ws(ws, ws_bytes, b, n1, n2, num_leaf, w)) return e;
    if (!packed) return fail("null argument");
    launch_final_proj_norm(packed + PW_FINAL_W, packed + PW_FINAL_B, w, static_cast<hipStream_t>(stream));
    return check_launch("final_proj_norm");
}

int gatsspg_score_dual_softmax_match(int b, int n1, int n2, int num_leaf, float scale_factor, float match_threshold,
                                     float* conf, int64_t* matches0, int64_t* matches1, float* mscores0,
                                     float* mscores1, void* ws, size_t ws_bytes, void* stream) {
    Workspace w;
    if (int e = check_ws(ws, ws_bytes, b, n1, n2, num_leaf, w)) return e;
    if (!conf || !matches0 || !matches1 || !mscores0 || !mscores1) return fail("null output pointer");
    if (int e = check_scale(scale_factor)) return e;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int shifted = softmax_shifted(scale_factor);
    launch_score_exp(w, conf, scale_factor, shifted, s);
    launch_dual_softmax_match(w, conf, scale_factor, shifted, match_threshold, matches0, matches1, mscores0, mscores1, s);
    return check_launch("score_dual_softmax_match");
}

int gatsspg_forward(const float* packed, const float* desc2d_query, const float* desc3d_db, const float* desc2d_db, int b,
                    int n1, int n2, int num_leaf, int flags, float scale_factor, float match_threshold, float* conf,
                    int64_t* matches0, int64_t* matches1, float* mscores0, float* mscores1, void* ws, size_t ws_bytes,
                    void* stream) {
    return forward_impl(packed, desc2d_query, desc3d_db, desc2d_db, b, n1, n2, num_leaf, flags, scale_factor,
                        match_threshold, conf, matches0, matches1, mscores0, mscores1, ws, ws_bytes, stream, nullptr);
}

int gatsspg_forward_profiled(const float* packed, const float* desc2d_query, const float* desc3d_db,
                             const float* desc2d_db, int b, int n1, int n2, int num_leaf, int flags, float scale_factor,
                             float match_threshold, float* conf, int64_t* matches0, int64_t* matches1, float* mscores0,
                             float* mscores1, void* ws, size_t ws_bytes, void* stream, int kernel_id, int occurrence,
                             void* ev_start, void* ev_stop) {
    if (kernel_id < 0 || kernel_id >= KID_COUNT) return fail("kernel_id %d out of range", kernel_id);
    if (!ev_start || !ev_stop) return fail("null event");
    ProfileHook hk;
    memset(&hk, 0, sizeof(hk));
    hk.kernel_id = kernel_id;
    hk.occurrence = occurrence;
    hk.start = static_cast<hipEvent_t>(ev_start);
    hk.stop = static_cast<hipEvent_t>(ev_stop);
    if (int e = forward_impl(packed, desc2d_query, desc3d_db, desc2d_db, b, n1, n2, num_leaf, flags, scale_factor,
                             match_threshold, conf, matches0, matches1, mscores0, mscores1, ws, ws_bytes, stream, &hk))
        return e;
    if (hk.seen[kernel_id] <= occurrence) return fail("kernel %d was launched %d times, occurrence %d never ran", kernel_id, hk.seen[kernel_id], occurrence);
    return 0;
}

#ifdef GATSSPG_PROFILING_BUILD
/* profiling builds only (not part of the public header): per-workgroup timeline of mlp0_kernel */
void gatsspg_debug_set_trace(void* buf) { g_trace = static_cast<unsigned long long*>(buf); }
#endif

size_t gatsspg_db_cache_bytes(int b, int n2) {
    if (b < 1 || n2 < 2) return 0;
    return carve_cache(nullptr, b, n2).bytes;
}

int gatsspg_prepare_database(const float* packed, const float* desc3d_db, const float* desc2d_db, int b, int n2,
                             int num_leaf, int flags, void* cache, size_t cache_bytes, void* ws, size_t ws_bytes,
                             void* stream) {
    Workspace w;
    if (int e = check_ws(ws, ws_bytes, b, 2, n2, num_leaf, w, flags)) return e;   // 2 dummy (zero) query columns
    if (!packed || !desc3d_db || !desc2d_db || !cache) return fail("null argument");
    const DbCache c = carve_cache(cache, b, n2);
    if (cache_bytes < c.bytes) return fail("database cache too small: %zu < %zu bytes", cache_bytes, c.bytes);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Workspace wy = windowed(w, 1);
    launch_load_state(nullptr, desc3d_db, w, s);
    enqueue_gats(packed, 0, desc2d_db, num_leaf, flags, w, s);              // gnn.layers.0 (3D side only by nature)
    enqueue_attn(packed, 0, GATSSPG_LAYER_SELF, wy, s);                     // gnn.layers.1, 3D side
    const float* a1 = attn_w(packed, 1);
    launch_qkv_kv(a1 + AttnW::WQKV, a1 + AttnW::BQKV, attn_wb(packed, 1), wy, s);   // gnn.layers.2: 3D-side Q, KV, ksum
    launch_kv_final(a1 + AttnW::W0, wy, 1, nullptr, s);                             //   (the final sums land in w.kvfin)
    launch_store_state(w.Z, nullptr, c.Y2, w, s);
    launch_store_state(w.Q, nullptr, c.QY, w, s);
    if (hipMemcpy2DAsync(c.kvY, sizeof(float) * H * KVP, w.kvfin + (size_t)H * KVP, sizeof(float) * 2 * H * KVP,
                         sizeof(float) * H * KVP, b, hipMemcpyDeviceToDevice, s) != hipSuccess)
        return fail("prepare_database: copy of the KV sums failed");
    if (gats_caches_leaf_logits(num_leaf, flags))   // leaf . u1 of the three GATs layers still to come
        launch_gats_leaf_logits(gats_w(packed, 1) + GatsW::U1, GatsW::SIZE, 3, desc2d_db, c.LL, w, s);
    return check_launch("prepare_database");
}

int gatsspg_forward_cached(const float* packed, const float* desc2d_query, const float* desc2d_db, const void* cache,
                           size_t cache_bytes, int b, int n1, int n2, int num_leaf, int flags, float scale_factor,
                           float match_threshold, float* conf, int64_t* matches0, int64_t* matches1, float* mscores0,
                           float* mscores1, void* ws, size_t ws_bytes, void* stream) {
    Workspace w;
    if (int e = check_ws(ws, ws_bytes, b, n1, n2, num_leaf, w, flags)) return e;
    if (!packed || !desc2d_query || !desc2d_db || !cache) return fail("null input pointer");
    if (!conf || !matches0 || !matches1 || !mscores0 || !mscores1) return fail("null output pointer");
    if (int e = check_scale(scale_factor)) return e;
    const DbCache c = carve_cache(const_cast<void*>(cache), b, n2);
    if (cache_bytes < c.bytes) return fail("database cache too small: %zu < %zu bytes", cache_bytes, c.bytes);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Workspace wx = windowed(w, 0);
    launch_load_state(desc2d_query, c.Y2, w, s);                            // state = [X0 | cached Y2]
    enqueue_attn(packed, 0, GATSSPG_LAYER_SELF, wx, s);                     // gnn.layers.1, query side only
    const float* a1 = attn_w(packed, 1);
    launch_qkv_kv(a1 + AttnW::WQKV, a1 + AttnW::BQKV, attn_wb(packed, 1), wx, s);   // gnn.layers.2: query-side Q, KV, ksum
    launch_load_columns(nullptr, c.QY, w.Q, w, s);                          // 3D-side Q from the cache
    launch_kv_final(a1 + AttnW::W0, w, 1, c.kvY, s);                        // query-side sums from the partials, 3D-side sums from the cache
    launch_mlp(a1 + AttnW::W0, a1 + AttnW::B0, a1 + AttnW::W3, a1 + AttnW::B3, attn_wb(packed, 1), w, s);
    const bool cached_logits = gats_caches_leaf_logits(num_leaf, flags);
    for (int t = 1; t < 4; ++t) {
        enqueue_gats(packed, t, desc2d_db, num_leaf, flags, w, s, nullptr, nullptr, nullptr,
                     cached_logits ? c.LL + (size_t)(t - 1) * c.ll_layer : nullptr);
        enqueue_attn(packed, 2 * t, GATSSPG_LAYER_SELF, w, s);
        enqueue_attn(packed, 2 * t + 1, GATSSPG_LAYER_CROSS, w, s);
    }
    const int shifted = softmax_shifted(scale_factor);
    launch_final_proj_norm(packed + PW_FINAL_W, packed + PW_FINAL_B, w, s);
    launch_score_exp(w, conf, scale_factor, shifted, s);
    launch_dual_softmax_match(w, conf, scale_factor, shifted, match_threshold, matches0, matches1, mscores0, mscores1, s);
    return check_launch("forward_cached");
}

size_t gatsspg_kenc_scratch_bytes(int b, int n) { return (b < 1 || n < 1) ? 0 : kenc_scratch_bytes(b, n); }

int gatsspg_keypoint_encoder(const gatsspg_kenc_weights* kw, const float* kpts, const float* scores, int b, int n,
                             float* out, void* scratch, size_t scratch_bytes, void* stream) {
    if (!kw || !kpts || !scores || !out || !scratch) return fail("null argument");
    if (b < 1 || n < 2) return fail("keypoint encoder needs b >= 1 and n >= 2 (InstanceNorm1d)");
    if (kw->inp_dim != 3 && kw->inp_dim != 4) return fail("inp_dim must be 3 or 4");
    for (int i = 0; i < 4; ++i)
        if (!kw->w[i] || !kw->b[i]) return fail("null encoder weight");
    if (scratch_bytes < kenc_scratch_bytes(b, n)) return fail("scratch too small");
    launch_kenc(kw->w, kw->b, kw->inp_dim, kpts, scores, b, n, out, scratch, static_cast<hipStream_t>(stream));
    return check_launch("keypoint_encoder");
}

}  // extern "C"
