// Host-side launcher prototypes shared between the kernel translation units and the C ABI.
#pragma once
#include "gatsspg_common.h"

namespace gatsspg {

// Kernel ids (also the `kernel_id` of gatsspg_forward_profiled, include/gatsspg.h)
enum KernelId {
    KID_LOAD_STATE = 0, KID_GATS = 1, KID_QKV_KV = 2, KID_KV_FINAL = 3, KID_ATTN_APPLY = 4 /* retired: folded into kv_final + mlp0 */, KID_MLP0 = 5,
    KID_STAT_FINAL = 6, KID_MLP3 = 7, KID_FINAL_PROJ = 8, KID_SCORE_EXP = 9, KID_CONF_FINALIZE = 10,
    KID_MATCH_TAIL = 11, KID_GATS_WLT = 12, KID_SOFTMAX_STATS = 13, KID_COUNT = 14
};

// Optional HIP-event bracket around the `occurrence`-th launch of kernel `kernel_id` inside one forward
// (events live on the same stream as the kernels).  nullptr = no instrumentation.
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
#define GATSSPG_LAUNCH(hook, kid, stream, ...)      \
    do {                                            \
        hook_before(hook, kid, stream);             \
        hipLaunchKernelGGL(__VA_ARGS__);            \
        hook_after(hook, kid, stream);              \
    } while (0)

// Tuning knobs.  The product library has NO environment lookups: every knob is its compiled-in default.  A library built
// with -DGATSSPG_TUNING (python -m onepose_amd.build_ext --tuning -> libgatsspg_hip_tuning.so, never loaded by the
// package) reads GATSSPG_<NAME> from the environment once, for A/B runs of alternative tile shapes (tools/ab_tuning.py).
int tuning_knob(const char* name, int dflt);

// gatsspg_gemm_kernels.hip
#ifdef GATSSPG_PROFILING_BUILD
extern unsigned long long* g_trace;  // per-workgroup timeline buffer of mlp0_kernel (nullptr = off)
#endif
// packedb: the split-bf16 weight planes of this layer (AttnWB offsets), used when w.prec == 1
void launch_qkv_kv(const float* Wqkv, const float* bqkv, const unsigned short* packedb, const Workspace& w, hipStream_t s,
                   ProfileHook* hk = nullptr);
// KV / ksum sums of every active SOURCE segment + the message operators M_t of their TARGET segments (cross: the other side of
// the frame) for mlp.0.  W0: the layer's packed mlp.0 operator [512][512].  kv_src: nullptr, or (amortised mode) the cached final
// KV sums [b][4][KVP] of the 3D-side sources, used instead of the partials.
void launch_kv_final(const float* W0, const Workspace& w, int cross, const float* kv_src, hipStream_t s, ProfileHook* hk = nullptr);
void launch_mlp(const float* W0, const float* b0, const float* W3, const float* b3, const unsigned short* packedb,
                const Workspace& w, hipStream_t s, ProfileHook* hk = nullptr);
// gatsspg_split_kernels.hip: the same three GEMMs on the LDS-DMA split-16-bit loop (gemm_split_glds.h); sc = the layer's AttnW::SC
void launch_qkv_kv_sp(const float* sc, const float* bqkv, const unsigned short* packedb, const Workspace& w, hipStream_t s, ProfileHook* hk = nullptr);
void launch_mlp0_sp(const float* sc, const float* b0, const unsigned short* packedb, const Workspace& w, hipStream_t s, ProfileHook* hk = nullptr);
void launch_mlp3_sp(const float* sc, const float* b3, const unsigned short* packedb, const Workspace& w, hipStream_t s, ProfileHook* hk = nullptr);
// score contraction + exp on the split loop (fp32-class modes only: bf16x6, fp16x4): A = the 16-bit planes of the query descriptors
// written by final_proj_norm_kernel (w.MDTp), B = the fp32 3D descriptors.  Same outputs and partial layout as launch_score_exp.
bool score_on_split_loop(int prec, int shifted);
void launch_score_exp_sp(const Workspace& w, float* conf, float scale, hipStream_t s, ProfileHook* hk = nullptr);
constexpr int SCORE_SPLIT_SCALE_LOG2 = 10;   // unit-norm descriptors (|x| <= 1) are multiplied by 2^10 before the fp16 split
// the fp32 arithmetic (exact v_mfma_f32_32x32x2_f32) on the LDS-DMA loop: same kernels, MODE 0, the fp32 operators as the A operand
void launch_qkv_kv_dma(const float* Wqkv, const float* bqkv, const Workspace& w, hipStream_t s, ProfileHook* hk = nullptr);
void launch_mlp0_dma(const float* W0, const float* b0, const Workspace& w, hipStream_t s, ProfileHook* hk = nullptr);
void launch_mlp3_dma(const float* W3, const float* b3, const Workspace& w, hipStream_t s, ProfileHook* hk = nullptr);
// true if the split-precision launch goes to the kernels above (fp16 modes: always; bf16 modes: unless a tuning build says otherwise)
bool split_loop_glds(int prec);
// true (tuning builds, GATSSPG_STAT_FUSED=1): the InstanceNorm statistics are finished inside the mlp.0 launch by its last workgroups
// (stat_last_block); the product path is the separate stat_final launch (measured faster one frame at a time: DESIGN.md 14e)
bool stat_fused();
// true: mlp.0 of the split loop writes U point-major (U^T [ld][512]) with InstanceNorm partials per 32-point strip and mlp.3 reads it as
// a transposed B operand (fp16 modes; gatsspg_split_kernels.hip).  stat_final then merges 32-column partials.
bool sp_ut_on(int prec);
void launch_final_proj_norm(const float* Wf, const float* bf, const Workspace& w, hipStream_t s, ProfileHook* hk = nullptr);
// shifted = 0: E = exp(S) into conf + row/col sum partials (|S| <= 80); 1: raw scores S into conf (max-subtracting path)
void launch_score_exp(const Workspace& w, float* conf, float scale, int shifted, hipStream_t s, ProfileHook* hk = nullptr);
int score_tile_rows();   // rows / columns of a score tile = what one column / row partial sums over (conf_finalize needs the counts)
int score_tile_cols();
void launch_gats_wlt(const float* W, const float* P, const Workspace& w, int add_h, hipStream_t s, ProfileHook* hk = nullptr);
void launch_split_weights(float* packed, unsigned short* packedb, hipStream_t s);   // also writes the fp16 plane scales (AttnW::SC) into `packed`

// gatsspg_stream_kernels.hip
void launch_load_state(const float* dq, const float* d3, const Workspace& w, hipStream_t s, ProfileHook* hk = nullptr);
// copies compact [b,256,n] columns of one or both sides into `dst` ([256][ld]); a null side is left untouched
void launch_load_columns(const float* c2, const float* c3, float* dst, const Workspace& w, hipStream_t s);
void launch_store_state(const float* src, float* out2d, float* out3d, const Workspace& w, hipStream_t s, ProfileHook* hk = nullptr);
// h3: where the layer reads the 3D-point descriptors from: nullptr = the state Z; otherwise the caller's compact
// [b,256,n2] tensor (first layer of a forward: the state load is fused, `dq` is copied into the 2D side by spare workgroups)
void launch_gats(const float* u1, const float* u2, const float* leaves, int num_leaf, int flags, float* dst,
                 const Workspace& w, hipStream_t s, ProfileHook* hk = nullptr, const float* h3 = nullptr,
                 const float* dq = nullptr, const float* cached_logits = nullptr);
// per-object cache of the leaf logits (amortised mode): [nlayers][b][tiles][32] floats, gats_leaf_logit_floats per layer;
// launch_gats(..., cached_logits = that layer's block) then skips the leaf . u1 products
bool gats_caches_leaf_logits(int num_leaf, int flags);
size_t gats_leaf_logit_floats(int b, int n2);
void launch_gats_leaf_logits(const float* u1_first, int u1_stride, int nlayers, const float* leaves, float* cl,
                             const Workspace& w, hipStream_t s);
// true if launch_gats can take h3 / dq for this configuration (fused state load)
bool gats_fuses_state_load(int num_leaf, int flags, const Workspace& w);
void launch_dual_softmax_match(const Workspace& w, float* conf, float scale, int shifted, float match_threshold,
                               int64_t* matches0, int64_t* matches1, float* mscores0, float* mscores1, hipStream_t s,
                               ProfileHook* hk = nullptr);
void launch_pack_weights(const void* raw_struct_host, float* packed, hipStream_t s);
size_t kenc_scratch_bytes(int b, int n);
void launch_kenc(const float* const* w, const float* const* bias, int inp_dim, const float* kpts, const float* scores,
                 int b, int n, float* out, void* scratch, hipStream_t s);

}  // namespace gatsspg
