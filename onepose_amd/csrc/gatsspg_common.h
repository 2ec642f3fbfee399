// Shared host/device definitions: the column layout of the activation state in HBM, the
// workspace carve-up and the packed-weights blob.  See DESIGN.md "Data layout in HBM".
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace gatsspg {

constexpr int D = 256;    // descriptor_dim (hard-coded by the reference GNN, GATs_SuperGlue.py:35-36)
constexpr int H = 4;      // heads (GATs_SuperGlue.py:43)
constexpr int DH = 64;    // channels per head
constexpr int CP = 128;   // column padding granule: every (frame, side) segment starts on a multiple of CP
constexpr int BK = 32;    // K tile of every MFMA GEMM
constexpr int KVMAX = 8;           // bound data of a KV partial (message-operator scale of the fp16 modes): [0..3] per-wave largest key sum of the tile, [4..7] max |V| of the tile
constexpr int KVP = DH * DH + DH + KVMAX;  // one KV partial: the 64x64 KV matrix TRANSPOSED, [d][q] (kv_final owns 4-row d blocks), + 64 ksum + maxima
constexpr int MOP_LD = 512;        // row stride of the per-segment message operators M_seg [512][256] (two segments share a [512][512] block)
constexpr int MPL_PLANE = 512 * 256;   // bf16 elements of one split plane of M_seg (slab-major like the weight planes)

// tile widths baked into the partial-sum buffers
constexpr int QKV_BN = 64;    // column tile of the QKV+KV-partial kernel (one KV partial per tile)
constexpr int MLP0_BN = 64;   // column tile of the mlp.0 kernel (one InstanceNorm partial per tile)
constexpr int SC_BM = 128;    // score kernel row tile (over n1)
constexpr int SC_BN = 64;     // score kernel column tile (over n2)
constexpr int CF_ROWS = 16;   // conf-finalize strip height: 4 waves x 4 rows, all loaded before any is processed
constexpr int CF_COLS = 512;  // conf-finalize chunk width: 2 x 16 B per lane and row

// Activation state: channel-major [channels][ld] fp32; frame f owns columns [f*np, (f+1)*np):
// its N_2D query points at [0, n1) of that range (padded to n1p), its N_3D points at
// [n1p, n1p + n2) (padded to n2p).  A "segment" is one (frame, side) pair: seg = 2*f + side.
struct ColLayout {
    int b, n1, n2, n1p, n2p, np, ld;
    // active column window of a launch, per frame, in 64-column tiles: tiles [tw_first, tw_first + tw_count) of
    // every frame (default: the whole frame).  Lets the per-point kernels run on the query side or the 3D side only.
    int tw_first, tw_count;
    int side_mask;   // bit 0: 2D-side segments active, bit 1: 3D-side segments (segment-level reduction kernels)
    int xgs;         // XCD granule of a launch's tile map, log2 of column tiles (xcd_tile_map_g; set by the launcher, 0 = one tile)
};

__host__ __device__ inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

__host__ __device__ inline ColLayout make_layout(int b, int n1, int n2) {
    ColLayout L;
    L.b = b; L.n1 = n1; L.n2 = n2;
    L.n1p = round_up(n1, CP); L.n2p = round_up(n2, CP);
    L.np = L.n1p + L.n2p; L.ld = b * L.np;
    L.tw_first = 0; L.tw_count = L.np / 64; L.side_mask = 3; L.xgs = 0;
    return L;
}

// layout restricted to one side: side 0 = query (2D) columns, 1 = 3D columns
__host__ __device__ inline ColLayout side_window(ColLayout L, int side) {
    L.tw_first = side ? L.n1p / 64 : 0;
    L.tw_count = (side ? L.n2p : L.n1p) / 64;
    L.side_mask = side ? 2 : 1;
    return L;
}
// number of active 64-column tiles of a launch, and the global tile index of active tile t
__host__ __device__ inline int active_tiles(const ColLayout& L) { return L.b * L.tw_count; }
__host__ __device__ inline int global_tile(const ColLayout& L, int t) {
    return (t / L.tw_count) * (L.np / 64) + L.tw_first + t % L.tw_count;
}

struct TileSeg {
    int frame, side, seg, seg_start, valid;  // valid = number of real (non-pad) columns in the tile
};

__host__ __device__ inline TileSeg tile_seg(const ColLayout& L, int c0, int bn) {
    TileSeg t;
    t.frame = c0 / L.np;
    int r = c0 - t.frame * L.np;
    t.side = r >= L.n1p ? 1 : 0;
    t.seg = t.frame * 2 + t.side;
    t.seg_start = t.frame * L.np + (t.side ? L.n1p : 0);
    int v = t.seg_start + (t.side ? L.n2 : L.n1) - c0;
    t.valid = v < 0 ? 0 : (v > bn ? bn : v);
    return t;
}

// ---- packed weights blob (floats) -------------------------------------------------------------
struct AttnW {  // offsets inside one attention layer block
    static constexpr size_t WQKV = 0;                          // [768][256] rows: Q head-major, then per head K_h(64) V_h(64)
    static constexpr size_t BQKV = WQKV + 768 * 256;           // [768]
    static constexpr size_t W0 = BQKV + 768;                   // [512][512]: [:, :256] = mlp.0 x-half, [:, 256:] = W0b @ Wm (head-major cols)
    static constexpr size_t B0 = W0 + 512 * 512;               // [512] = b0 + W0b @ bm
    static constexpr size_t W3 = B0 + 512;                     // [256][512]
    static constexpr size_t B3 = W3 + 256 * 512;               // [256]
    // [8]: [0..2] the exact powers of two the fp16 planes of WQKV, W0[:, :256] and W3 are multiplied by before the split
    // (weight_scale_kernel: the matrix maximum lands in [2^13, 2^14)); [3] spare; [4 + h] = max_r sum_q |(W0b Wm)[r][h*64 + q]|, the
    // row-L1 norm of head h's message half: kv_final bounds the operator M_h with it.  The consumers scale the accumulators back.
    static constexpr size_t SC = B3 + 256;
    static constexpr size_t SIZE = SC + 8;
};
struct GatsW {
    static constexpr size_t U1 = 0;              // [256] = W @ a[:256]   (leaf logit vector)
    static constexpr size_t U2 = 256;            // [256] = W @ a[256:]   (3D-point logit vector)
    static constexpr size_t W = 512;             // [256][256] raw W (only used with_linear_transform)
    static constexpr size_t SIZE = 512 + 256 * 256;
};
constexpr size_t PW_ATTN = 0;
constexpr size_t PW_GATS = PW_ATTN + 8 * AttnW::SIZE;
constexpr size_t PW_FINAL_W = PW_GATS + 4 * GatsW::SIZE;
constexpr size_t PW_FINAL_B = PW_FINAL_W + 256 * 256;
constexpr size_t PW_TOTAL = PW_FINAL_B + 256;   // floats

// Split-bf16 planes of the three big GEMM operators (GATSSPG_FLAG_PREC_BF16X3 / _BF16X6): appended to the fp32 blob, in bf16
// elements from (unsigned short*)(packed + PW_TOTAL).  w = hi + lo with hi = RNE_bf16(w), lo = RNE_bf16(w - hi);
// rows in the order of the fp32 matrices they are split from, elements slab-major: (m, k) at ((k / 32) * M + m) * 32 + k % 32.
struct AttnWB {
    static constexpr size_t QKV_HI = 0;                          // [768][256]
    static constexpr size_t QKV_LO = QKV_HI + 768 * 256;
    static constexpr size_t W0_HI = QKV_LO + 768 * 256;          // [512][512]
    static constexpr size_t W0_LO = W0_HI + 512 * 512;
    static constexpr size_t W3_HI = W0_LO + 512 * 512;           // [256][512]
    static constexpr size_t W3_LO = W3_HI + 256 * 512;
    // third planes (GATSSPG_FLAG_PREC_BF16X6): lo2 = RNE_bf16(w - hi - lo), w = hi + lo + lo2 exactly
    static constexpr size_t QKV_LO2 = W3_LO + 256 * 512;
    static constexpr size_t W0_LO2 = QKV_LO2 + 768 * 256;
    static constexpr size_t W3_LO2 = W0_LO2 + 512 * 512;
    // fp16 hi / lo planes (GATSSPG_FLAG_PREC_FP16X3 / _FP16X4) of s * w, s = the matrix's power-of-two scale (AttnW::SC):
    // hi = RNE_fp16(s w), lo = RNE_fp16(s w - hi), same slab-major layout
    static constexpr size_t QKV_H16 = W3_LO2 + 256 * 512;
    static constexpr size_t QKV_L16 = QKV_H16 + 768 * 256;
    static constexpr size_t W0_H16 = QKV_L16 + 768 * 256;
    static constexpr size_t W0_L16 = W0_H16 + 512 * 512;
    static constexpr size_t W3_H16 = W0_L16 + 512 * 512;
    static constexpr size_t W3_L16 = W3_H16 + 256 * 512;
    static constexpr size_t SIZE = W3_L16 + 256 * 512;
};
constexpr size_t PWB_TOTAL = 8 * AttnWB::SIZE;                   // bf16 elements
constexpr size_t PACKED_BYTES = sizeof(float) * PW_TOTAL + sizeof(unsigned short) * PWB_TOTAL;

// ---- workspace carve-up ---------------------------------------------------------------------------
struct Workspace {
    ColLayout L;
    int prec;          // 0: fp32 MFMA; 1: three-term split-bf16 (bf16x3), 2: six-term (bf16x6), 3 / 4: three- / four-term split-fp16 (fp16x3, fp16x4) main loops in qkv_kv / mlp0 / mlp3 (from the call's flags)
    int nt64;          // ld / 64 column tiles
    int nseg;          // 2*b
    int sc_nct, sc_nrt;   // score kernel tiles per frame (n2p/SC_BN, n1p/SC_BM)
    int cf_nst, cf_nch;   // conf-finalize strips / chunks per frame
    float *Z, *Q, *MSG, *U, *MD;     // MD aliases Q (Q is dead after the GNN); MSG: scratch of the with_linear_transform GATs path
    float *MDT;                      // query-side normalised descriptors, point-major [b][n1p][256]; aliases MSG
    unsigned short *MDTp;            // their 16-bit planes for the split score contraction (bf16x6 / fp16x4): [3][8 slabs][b * n1p][32], slab-major
    float *kvpart, *kvfin, *statpart, *stats;
    int *statcnt;                    // [nseg][8] arrival counters of the fused InstanceNorm statistics (stat_last_block)
    // linear attention folded into mlp.0 (kv_final_kernel -> mlp0_kernel): per TARGET segment t the operator
    // M_t = (W0b Wm)[:, head h] KV_h(source) for the four heads [512][4 x 64], and the source's ksum.
    float *Mop;                      // [b][512][512]: segment 2f at columns 256..511, segment 2f+1 at columns 0..255 of frame f's block
    unsigned short *Mpl;             // split-bf16 planes of M_t (prec != 0): [nseg][3][8 slabs][512][32]
    float *ksumT;                    // [nseg][4][64]
    float *zsc;                      // [nseg][4]: per head, fold factor of the target segment's operator planes = (scale of the W0 planes) / (scale of Mpl_h), a power of two
    float *rowpart, *colpart, *rs, *cs;
    float *rmax_v, *cmax_v, *rshift, *cshift;   // rshift / cshift: row / column maxima of the max-subtracting dual softmax
    int *rmax_i, *cmax_i;
    size_t bytes;
};

inline size_t align_up(size_t x) { return (x + 255) & ~size_t(255); }

// M_t of segment `seg`: element (r, c), c = h*64 + d, at mop_seg(Mop, seg)[r * MOP_LD + c]
__host__ __device__ inline float* mop_seg(float* Mop, int seg) { return Mop + (size_t)(seg >> 1) * 512 * MOP_LD + ((seg & 1) ? 0 : 256); }
__host__ __device__ inline const float* mop_seg(const float* Mop, int seg) {
    return Mop + (size_t)(seg >> 1) * 512 * MOP_LD + ((seg & 1) ? 0 : 256);
}

inline Workspace carve_workspace(void* base, int b, int n1, int n2) {
    Workspace w;
    w.L = make_layout(b, n1, n2);
    w.prec = 0;
    const ColLayout& L = w.L;
    w.nt64 = L.ld / 64;
    w.nseg = 2 * b;
    w.sc_nct = L.n2p / SC_BN;
    w.sc_nrt = L.n1p / SC_BM;
    w.cf_nst = (n1 + CF_ROWS - 1) / CF_ROWS;
    w.cf_nch = (n2 + CF_COLS - 1) / CF_COLS;
    char* p = static_cast<char*>(base);
    size_t off = 0;
    auto take = [&](size_t nbytes) { char* r = p ? p + off : nullptr; off += align_up(nbytes); return r; };
    const size_t ld = L.ld;
    w.Z = (float*)take(sizeof(float) * D * ld);
    w.Q = (float*)take(sizeof(float) * D * ld);
    w.MSG = (float*)take(sizeof(float) * D * ld);
    w.U = (float*)take(sizeof(float) * 2 * D * ld);
    w.MD = w.Q;
    w.MDT = w.MSG;   // b * n1p * 256 floats <= 256 * ld
    w.MDTp = (unsigned short*)take(sizeof(unsigned short) * 3 * (size_t)b * L.n1p * D);
    w.kvpart = (float*)take(sizeof(float) * (size_t)w.nt64 * H * KVP);
    w.kvfin = (float*)take(sizeof(float) * (size_t)w.nseg * H * KVP);
    w.Mop = (float*)take(sizeof(float) * (size_t)b * 512 * MOP_LD);
    w.Mpl = (unsigned short*)take(sizeof(unsigned short) * (size_t)w.nseg * 3 * MPL_PLANE);
    w.ksumT = (float*)take(sizeof(float) * (size_t)w.nseg * H * DH);
    w.zsc = (float*)take(sizeof(float) * (size_t)w.nseg * H);
    w.statpart = (float*)take(sizeof(float) * (size_t)w.nt64 * 2 * 2 * 512);   // one partial per 64 columns, or per 32 (mlp0_sp's transposed epilogue)
    w.stats = (float*)take(sizeof(float) * (size_t)w.nseg * 2 * 512);
    w.statcnt = (int*)take(sizeof(int) * (size_t)w.nseg * 8);
    w.rowpart = (float*)take(sizeof(float) * (size_t)b * w.sc_nct * L.n1p);
    w.colpart = (float*)take(sizeof(float) * (size_t)b * w.sc_nrt * L.n2p);
    w.rs = (float*)take(sizeof(float) * (size_t)b * L.n1p);
    w.cs = (float*)take(sizeof(float) * (size_t)b * L.n2p);
    w.rmax_v = (float*)take(sizeof(float) * (size_t)b * w.cf_nch * L.n1p);
    w.rmax_i = (int*)take(sizeof(int) * (size_t)b * w.cf_nch * L.n1p);
    w.cmax_v = (float*)take(sizeof(float) * (size_t)b * w.cf_nst * L.n2p);
    w.cmax_i = (int*)take(sizeof(int) * (size_t)b * w.cf_nst * L.n2p);
    w.rshift = (float*)take(sizeof(float) * (size_t)b * L.n1p);
    w.cshift = (float*)take(sizeof(float) * (size_t)b * L.n2p);
    w.bytes = off;
    return w;
}

}  // namespace gatsspg
