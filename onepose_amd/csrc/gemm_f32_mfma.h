// fp32 MFMA GEMM main loop for gfx950 (v_mfma_f32_32x32x2_f32: exact f32 FMA chain at the
// 157 TFLOP/s matrix rate).  C[BM x BN] += A[BM x K] * B[K x BN], one workgroup of 4 or 8 wave64s
// (arranged WM x WN), K consumed in BK=32 slabs, LDS double-buffered with register
// staging two slabs ahead (two register sets) -- one barrier per slab.
//
// Operand layouts
//   A row-major [M][K] (weights): LDS image [BM][BK+4]; the +4 pad makes the ds_read_b128
//     fragment reads (lane = row, 16 consecutive k) conflict-free (row stride 36 dwords -> 16-B
//     slot index 9*row mod 16 is a bijection over each 16-lane group).
//   A "KM" [K][M] (M contiguous, e.g. the normalised descriptors of the score GEMM): LDS [BK][BM].
//   B [K][N] (N contiguous; channel-major activations): LDS image [BK][BN]; fragment reads are
//     ds_read_b32 with lane = column -> 32 consecutive banks, conflict-free.
// k assignment inside a slab: the MFMA consumes 2 k per issue (lane-half 0 -> k, lane-half 1 -> k');
// half 0 takes slab k = s, half 1 takes k = 16 + s (s = 0..15), so a lane's 16 A values are
// contiguous in k and come in with 4 ds_read_b128.
#pragma once
#include "gatsspg_common.h"

namespace gatsspg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
// staging registers use a first-class vector type: HIP's float4 is a struct, and a struct copy
// global -> register -> LDS is forwarded by MemCpyOpt into a late global->LDS copy (load next to its use)
typedef float vf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ vf4 ldg4(const float* p) { return *reinterpret_cast<const vf4*>(p); }
// same registers, but the address is only 4-byte aligned (still one global_load_dwordx4 on gfx950)
typedef vf4 vf4u __attribute__((aligned(4)));

// BU_: the B slab pointers are only 4-byte aligned (column-shifted views of an activation plane, used by the
// 3x3 convolutions of the SuperPoint extractor)
// KS_: intra-workgroup K split (fp32 loop only).  KS = 2 doubles the waves of a workgroup: wave group kg = wave / (WM * WN)
// multiplies the kg-th half of every 32-deep slab (16 of its 32 k) into its own accumulators, ksplit_reduce() adds the groups
// at the end.  Same tile, same staging, half the MFMA chain per wave -- for launches that leave CUs empty (small N), where a
// lone workgroup's time is its waves' dependent MFMA chain, not the matrix pipe.
template <int BM_, int BN_, int WM_, int WN_, bool AKM_, bool BU_ = false, int KS_ = 1>
struct GemmTile {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, KS = KS_;
    static constexpr bool AKM = AKM_, BU = BU_;
    static constexpr int TM = BM / WM / 32;   // 32x32 MFMA tiles per wave along M
    static constexpr int TN = BN / WN / 32;   // ... along N
    static constexpr int A_STRIDE = AKM ? BM : (BK + 4);
    static constexpr int A_FLOATS = AKM ? BK * BM : BM * (BK + 4);
    static constexpr int B_FLOATS = BK * BN;
    static constexpr int STAGE_FLOATS = A_FLOATS + B_FLOATS;
    static constexpr int SMEM_FLOATS = 2 * STAGE_FLOATS;
    static constexpr int WAVES_MN = WM * WN;           // waves that tile the output; x KS wave groups
    static constexpr int THREADS = 64 * WM * WN * KS;  // 4, 8 or 16 waves
    static constexpr int A_VEC = BM * BK / 4 / THREADS;   // float4 per thread per slab
    static constexpr int B_PIECES = BK * BN / 4;          // float4 pieces of a B slab
    // a narrow tile on many waves has fewer B pieces than threads: every thread still moves one piece, the surplus threads
    // duplicate the piece of thread (tid mod B_PIECES) -- same bytes to the same LDS address, no guarded loads
    static constexpr int B_VEC = B_PIECES >= THREADS ? B_PIECES / THREADS : 1;
    static_assert(WM * WN * KS == 4 || WM * WN * KS == 8 || WM * WN * KS == 16, "workgroup = 4, 8 or 16 waves");
    static_assert(KS == 1 || KS == 2, "K split: one or two wave groups");
    static_assert(TM >= 1 && TN >= 1, "wave tile must hold at least one 32x32 MFMA tile");
    static_assert(A_VEC >= 1 && (B_PIECES % THREADS == 0 || THREADS % B_PIECES == 0), "tile / workgroup mismatch");
};

// row (within a 32x32 MFMA tile) held by accumulator register r of a lane in half `half`
__device__ __forceinline__ int mfma_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// Operand access is described by wave-uniform slab pointers plus loop-invariant per-thread byte
// offsets, so the steady-state loop carries no per-thread address arithmetic (the loads use the
// scalar-base + 32-bit-VGPR-offset form):
//   a_slab(kt): pointer to A slab kt.  row-major A: &A[row0][kt*32] (row stride lda);
//               KM A: &A_km[kt*32][m0] (row stride lda).
//   b_slab(kt): pointer to B slab kt: &B[kt*32][col0] (row stride ldb).
//   x_slab(kt): pointer to 32 per-row float2 (e.g. InstanceNorm mean / rstd) or nullptr-like dummy;
//   bxform(v, aux): applied to the B registers when they are written to LDS.
struct NoXform {
    __device__ __forceinline__ void operator()(vf4&, float2) const {}
};
// tile column (multiple of 4) -> element offset from the B slab pointer; identity for a contiguous column tile.
// The SuperPoint conv+pool kernels use a 2-row x 64-column image patch as their 128 "columns".
struct IdentityCol {
    __device__ __forceinline__ int operator()(int c) const { return c; }
};
__device__ __forceinline__ vf4 ldg4_off(const float* base, unsigned byte_off) {
    return *reinterpret_cast<const vf4*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ vf4 ldg4u_off(const float* base, unsigned byte_off) {
    return *reinterpret_cast<const vf4u*>(reinterpret_cast<const char*>(base) + byte_off);
}


// ---------------------------------------------------------------------------------------------------------------------
// Main-loop hooks of the mlp.0 kernel with the linear-attention apply folded in (GATs_SuperGlue.py:78-79,101,113,122):
//   u = W0a x + sum_h z_h (.) (M_h Qf_h) + b,   M_h = (W0b Wm)[:, head h] KV_h  (kv_final_kernel),   Qf = elu(q) + 1,
//   z_h[n] = 1 / (sum_d Qf_h[d][n] ksum_h[d] + 1e-6).
// The K loop runs over [x ; Qf] (16 slabs of 32): slabs 0..7 accumulate the x part, slabs 8 + 2h, 9 + 2h the product of
// head h into a zeroed accumulator, which is folded into the kept sum with the per-column z_h (a per-lane scalar in the
// 32x32 MFMA C layout) when the pair ends.  The denominators come from the staged Qf values themselves: every thread
// multiplies the 4 consecutive k rows of ONE column it sees (fp32 loop: from the LDS slab being computed; split-bf16
// loops: the registers it is about to split) with the source's ksum, the two slabs of a head are added in the thread, the
// eight per-wave partials go to LDS and are summed in wave order at the fold: fixed order, no atomics.
// Requires a 64-column tile on 8 waves (thread = (k group = wave, column = lane)) and one 32x32 MFMA tile per wave.
// ---------------------------------------------------------------------------------------------------------------------
struct NoHooks {
    static constexpr bool ENABLED = false;
};
struct AttnFoldHooks {
    static constexpr bool ENABLED = true;
    static constexpr int SPLIT = 8;          // first slab of the head phase
    static constexpr int ZP_FLOATS = 2 * 8 * 64;
    const float* ks;                         // ksum of the source segment, [4][64] (global, wave-uniform reads)
    float* zp;                               // LDS [2 (head parity)][8 waves][64 columns]
    f32x16 kept;                             // x part + folded heads
    float carry;                             // this thread's partial of the first slab of the current head
    int wave, lane, col;                     // col = this lane's column in the MFMA C layout (wn * 32 + l31)
    __device__ __forceinline__ void init(const float* ksum_src, float* zp_lds, int wn) {
        ks = ksum_src; zp = zp_lds; carry = 0.f;
        wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        lane = threadIdx.x & 63;
        col = wn * 32 + (lane & 31);
    }
    // j = slab index within the head phase (0..7); v0..v3 = rows 4 * wave .. + 3 of that slab, column `lane` of the tile
    __device__ __forceinline__ void partial(int j, float v0, float v1, float v2, float v3) {
        const int h = j >> 1;
        const float* k = ks + h * 64 + (j & 1) * 32 + wave * 4;
        float p = v0 * k[0];
        p = fmaf(v1, k[1], p);
        p = fmaf(v2, k[2], p);
        p = fmaf(v3, k[3], p);
        if (j & 1) zp[((h & 1) * 8 + wave) * 64 + lane] = carry + p;
        else carry = p;
    }
    // called after the barrier that ends the slab pair (i, i + 1)
    __device__ __forceinline__ void pair_end(int i, f32x16& acc) {
        if (i < SPLIT - 2) return;
        if (i == SPLIT - 2) {
            kept = acc;
        } else {
            const int h = (i - SPLIT) >> 1;
            const float* zr = zp + (h & 1) * 8 * 64 + col;
            float d = zr[0];
#pragma unroll
            for (int w = 1; w < 8; ++w) d += zr[w * 64];
            const float z = 1.f / (d + 1e-6f);
#pragma unroll
            for (int r = 0; r < 16; ++r) kept[r] = fmaf(z, acc[r], kept[r]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    }
};

// ABLATE (profiling only, wrong results): 1 = no global loads in the steady-state loop,
//   2 = no global loads and no LDS writes, 3 = steady-state loop cut to one step pair,
//   6 = every load of a wave hits the same 1 KiB (always L1-hot)
// QF (quarter fragments, one 32x32 MFMA tile per wave only): the operand fragments of a slab are read in four quarters of 4 k-steps (one
//   ds_read_b128 of A + four ds_read_b32 of B each) into TWO alternating register sets instead of two halves of 8: 16 live fragment
//   registers instead of 32.  Same MFMAs in the same order (bit-identical results); the point is the register budget -- mlp0_kernel
//   drops from 126 to <= 112 VGPRs, so that a 64-register wave of ANOTHER frame's HBM-bound kernel fits beside two of its workgroups
//   on a SIMD (DESIGN 12 item 5, round-5 judge item 2).
template <class T, class ASlab, class BSlab, class XSlabA, class XSlabB, class BXform, bool HAS_AUX, int ABLATE = 0,
          class BCol = IdentityCol, class Hooks = NoHooks, int QF = 0>
__device__ __forceinline__ void gemm_mainloop_ex(f32x16 (&acc)[T::TM][T::TN], float* smem, int KT, ASlab a_slab, int lda,
                                                 BSlab b_slab, int ldb, XSlabA x_mean, XSlabB x_rstd, BXform bxform,
                                                 BCol bcol = BCol(), Hooks* hooks = nullptr) {
    constexpr int BM = T::BM, BN = T::BN, TM = T::TM, TN = T::TN;
    if constexpr (Hooks::ENABLED)
        static_assert(BN == 64 && T::THREADS == 512 && TM == 1 && TN == 1 && !T::AKM, "fold hooks: 64-column tile, 8 waves, one MFMA tile per wave");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int kg = wave / T::WAVES_MN, wq = wave % T::WAVES_MN;   // K-split group, wave within the output tiling
    const int wm = wq / T::WN, wn = wq % T::WN;
    const int half = lane >> 5, l31 = lane & 31;
    static_assert(T::KS == 1 || (ABLATE == 0 && !T::AKM), "K split: plain row-major fp32 loop only");

    // loop-invariant per-thread byte offsets (global) and LDS float offsets of the staging slots
    unsigned a_goff[T::A_VEC], b_goff[T::B_VEC], x_goff[T::B_VEC];
    int a_soff[T::A_VEC], b_soff[T::B_VEC];
#pragma unroll
    for (int p = 0; p < T::A_VEC; ++p) {
        const int idx = p * T::THREADS + tid;
        if constexpr (T::AKM) {
            const int k = idx / (BM / 4), m = (idx % (BM / 4)) * 4;
            a_goff[p] = 4u * (unsigned)(k * lda + m);
            a_soff[p] = k * BM + m;
        } else {
            const int r = idx / (BK / 4), c = (idx % (BK / 4)) * 4;
            a_goff[p] = 4u * (unsigned)(r * lda + c);
            a_soff[p] = r * T::A_STRIDE + c;
        }
    }
#pragma unroll
    for (int p = 0; p < T::B_VEC; ++p) {
        const int idx = (p * T::THREADS + tid) % T::B_PIECES;
        const int k = idx / (BN / 4), c = (idx % (BN / 4)) * 4;
        b_goff[p] = 4u * (unsigned)(k * ldb + bcol(c));
        x_goff[p] = 4u * (unsigned)k;
        b_soff[p] = T::A_FLOATS + k * BN + c;
    }
    // fragment read offsets (floats)
    int afrag[TM], bfrag[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
        afrag[tm] = T::AKM ? (half * 16) * BM + (wm * TM + tm) * 32 + l31 : ((wm * TM + tm) * 32 + l31) * T::A_STRIDE + half * 16;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) bfrag[tn] = T::A_FLOATS + (half * 16) * BN + (wn * TN + tn) * 32 + l31;

    // two staging register sets: slab t+1 and slab t+2 are both in flight while slab t is computed
    vf4 ra0[T::A_VEC], rb0[T::B_VEC], ra1[T::A_VEC], rb1[T::B_VEC];
    float2 rx0[T::B_VEC], rx1[T::B_VEC];

    // piece q of the staging loads of slab kt: q < A_VEC -> A piece, else B piece (+ its per-row aux)
    constexpr int NPIECE = T::A_VEC + T::B_VEC;
    auto gload_piece = [&](int kt, int q, vf4(&ra)[T::A_VEC], vf4(&rb)[T::B_VEC], float2(&rx)[T::B_VEC]) {
        const int ks = ABLATE == 6 ? 0 : kt;
        if (q < T::A_VEC) {
            ra[q] = ldg4_off(a_slab(ks), ABLATE == 6 ? 16u * lane : a_goff[q]);   // 6: always L1-hot (profiling)
        } else {
            const int p = q - T::A_VEC;
            if constexpr (T::BU) rb[p] = ldg4u_off(b_slab(ks), b_goff[p]);
            else rb[p] = ldg4_off(b_slab(ks), ABLATE == 6 ? 16u * lane : b_goff[p]);
            if constexpr (HAS_AUX)
                rx[p] = make_float2(*reinterpret_cast<const float*>(reinterpret_cast<const char*>(x_mean(kt)) + x_goff[p]),
                                    *reinterpret_cast<const float*>(reinterpret_cast<const char*>(x_rstd(kt)) + x_goff[p]));
        }
    };
    auto gload = [&](int kt, vf4(&ra)[T::A_VEC], vf4(&rb)[T::B_VEC], float2(&rx)[T::B_VEC]) {
#pragma unroll
        for (int q = 0; q < NPIECE; ++q) gload_piece(kt, q, ra, rb, rx);
    };
    auto swrite = [&](float* stage, const vf4(&ra)[T::A_VEC], const vf4(&rb)[T::B_VEC], const float2(&rx)[T::B_VEC]) {
#pragma unroll
        for (int p = 0; p < T::A_VEC; ++p) *reinterpret_cast<vf4*>(stage + a_soff[p]) = ra[p];
#pragma unroll
        for (int p = 0; p < T::B_VEC; ++p) {
            vf4 v = rb[p];
            if constexpr (HAS_AUX) bxform(v, rx[p]);
            *reinterpret_cast<vf4*>(stage + b_soff[p]) = v;
        }
    };
    // fragments of one half slab (8 k-steps per lane half)
    auto read_frags = [&](const float* stage, int h, float (&a)[TM][8], float (&b)[TN][8]) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            if constexpr (T::AKM) {
#pragma unroll
                for (int s = 0; s < 8; ++s) a[tm][s] = stage[afrag[tm] + (h * 8 + s) * BM];
            } else {
                const vf4* ap = reinterpret_cast<const vf4*>(stage + afrag[tm] + h * 8);
                const vf4 x = ap[0], y = ap[1];
                a[tm][0] = x[0]; a[tm][1] = x[1]; a[tm][2] = x[2]; a[tm][3] = x[3];
                a[tm][4] = y[0]; a[tm][5] = y[1]; a[tm][6] = y[2]; a[tm][7] = y[3];
            }
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int s = 0; s < 8; ++s) b[tn][s] = stage[bfrag[tn] + (h * 8 + s) * BN];
    };
    auto mfma4 = [&](const float (&a)[TM][8], const float (&b)[TN][8], int s0, int cnt = 4) {
#pragma unroll
        for (int s = s0; s < s0 + cnt; ++s)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][s], b[tn][s], acc[tm][tn], 0, 0, 0);
    };
    // One step: compute slab from `cur`; start the global loads of the slab three steps ahead into the
    // register set just freed; write the slab one step ahead (registers filled two steps ago) into `nxt`.
    auto step = [&](const float* cur, float* nxt, int kt_cur, int kt_load, vf4(&ra)[T::A_VEC], vf4(&rb)[T::B_VEC],
                    float2(&rx)[T::B_VEC]) {
        float a0[TM][8], b0[TN][8], a1[TM][8], b1[TN][8];
        // Eight MFMA groups (2 k-steps each) with the memory work of the step placed between them (a 32x32x2
        // f32 MFMA occupies the matrix pipe for 64 cycles while the wave may issue independent instructions);
        // sched_barrier(0) between the groups keeps hipcc from regrouping them.  The global loads are NOT
        // issued as one burst: measured on mlp0, loads that hit L1 are free while L1-missing ones cost ~9 %
        // (a burst of 6 x 8 lines per wave fills the CU's miss path and the wave blocks at the load), so the
        // pieces are spread one or two per gap over the second half of the step.
        constexpr int GAPS = 5;                                   // gaps 3..7 carry the loads
        read_frags(cur, 0, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        mfma4(a0, b0, 0, 2);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(cur, 1, a1, b1);                               // gap 1
        if constexpr (Hooks::ENABLED) {
            if (kt_cur >= Hooks::SPLIT) {                         // denominators of the folded linear attention (block-uniform)
                const float* bq = cur + T::A_FLOATS + hooks->wave * 4 * BN + hooks->lane;
                hooks->partial(kt_cur - Hooks::SPLIT, bq[0], bq[BN], bq[2 * BN], bq[3 * BN]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma4(a0, b0, 2, 2);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ABLATE <= 1 || ABLATE == 6) swrite(nxt, ra, rb, rx);   // gap 2 (frees the register set)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 6; ++g) {                             // MFMA groups 3..8, gaps 3..7 between them
            if (g < 2) mfma4(a0, b0, 4 + 2 * g, 2);
            else mfma4(a1, b1, 2 * (g - 2), 2);
            __builtin_amdgcn_sched_barrier(0);
            if (g < GAPS) {
                if constexpr (ABLATE == 0 || ABLATE == 6) {
#pragma unroll
                    for (int q = 0; q < NPIECE; ++q)
                        if (q % GAPS == g) gload_piece(kt_load, q, ra, rb, rx);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // The same step with quarter fragments (QF): MFMA groups, gaps and the placement of the memory work as in `step`; the fragments of
    // quarter q + 2 are read into the set quarter q has just finished with.
    auto step_qf = [&](const float* cur, float* nxt, int kt_cur, int kt_load, vf4(&ra)[T::A_VEC], vf4(&rb)[T::B_VEC],
                       float2(&rx)[T::B_VEC]) {
        static_assert(!QF || (TM == 1 && TN == 1 && !T::AKM && T::KS == 1), "quarter fragments: one MFMA tile per wave, row-major A");
        float a0[4], b0[4], a1[4], b1[4];
        auto read_q = [&](int q, float (&a)[4], float (&b)[4]) {
            const vf4 x = *reinterpret_cast<const vf4*>(cur + afrag[0] + q * 4);
            a[0] = x[0]; a[1] = x[1]; a[2] = x[2]; a[3] = x[3];
#pragma unroll
            for (int s = 0; s < 4; ++s) b[s] = cur[bfrag[0] + (q * 4 + s) * BN];
        };
        auto mfma2 = [&](const float (&a)[4], const float (&b)[4], int s0) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s0], b[s0], acc[0][0], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s0 + 1], b[s0 + 1], acc[0][0], 0, 0, 0);
        };
        constexpr int GAPS = 5;
        read_q(0, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        mfma2(a0, b0, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_q(1, a1, b1);                                        // gap 1
        if constexpr (Hooks::ENABLED) {
            if (kt_cur >= Hooks::SPLIT) {
                const float* bq = cur + T::A_FLOATS + hooks->wave * 4 * BN + hooks->lane;
                hooks->partial(kt_cur - Hooks::SPLIT, bq[0], bq[BN], bq[2 * BN], bq[3 * BN]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma2(a0, b0, 2);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ABLATE <= 1 || ABLATE == 6) swrite(nxt, ra, rb, rx);   // gap 2 (frees the staging set)
        read_q(2, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 6; ++g) {                             // MFMA groups 3..8, gaps 3..7 between them
            if (g == 0) mfma2(a1, b1, 0);
            else if (g == 1) mfma2(a1, b1, 2);
            else if (g == 2) mfma2(a0, b0, 0);
            else if (g == 3) mfma2(a0, b0, 2);
            else if (g == 4) mfma2(a1, b1, 0);
            else mfma2(a1, b1, 2);
            __builtin_amdgcn_sched_barrier(0);
            if (g == 1) read_q(3, a1, b1);
            if (g < GAPS) {
                if constexpr (ABLATE == 0 || ABLATE == 6) {
#pragma unroll
                    for (int q = 0; q < NPIECE; ++q)
                        if (q % GAPS == g) gload_piece(kt_load, q, ra, rb, rx);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // K-split step (T::KS == 2): this wave group's half of the slab (8 k-steps per lane half = 8 MFMAs per 32x32 tile), the
    // memory work of the step between the four MFMA pairs
    auto step_ks = [&](const float* cur, float* nxt, int kt_cur, int kt_load, vf4(&ra)[T::A_VEC], vf4(&rb)[T::B_VEC],
                       float2(&rx)[T::B_VEC]) {
        float a0[TM][8], b0[TN][8];
        read_frags(cur, kg, a0, b0);
        if constexpr (Hooks::ENABLED) {
            if (kt_cur >= Hooks::SPLIT) {
                const float* bq = cur + T::A_FLOATS + hooks->wave * 4 * BN + hooks->lane;
                hooks->partial(kt_cur - Hooks::SPLIT, bq[0], bq[BN], bq[2 * BN], bq[3 * BN]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma4(a0, b0, 0, 2);
        __builtin_amdgcn_sched_barrier(0);
        swrite(nxt, ra, rb, rx);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            mfma4(a0, b0, 2 + 2 * g, 2);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < NPIECE; ++q)
                if (q % 3 == g) gload_piece(kt_load, q, ra, rb, rx);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // Pipeline (prefetch distance 2).  During step i the LDS buffer (i+1)&1 is free (every wave passed the
    // barrier that ended step i-1), so slab i+1 -- sitting in register set i&1 since step i-2 -- is written
    // there while slab i is computed from buffer i&1, and the freed registers start loading slab i+3.
    float* buf0 = smem;
    float* buf1 = smem + T::STAGE_FLOATS;
    const int last = KT - 1;
    // slab 0 goes through set 1 so that slab 1 (set 0) is requested in the same breath: the LDS write of slab 0 waits for the
    // first group only (loads retire in order), and step 0 no longer stalls a full round trip on slab 1
    gload(0, ra1, rb1, rx1);
    gload(min(1, last), ra0, rb0, rx0);   // set 0 <- slab 1
    swrite(buf0, ra1, rb1, rx1);
    gload(min(2, last), ra1, rb1, rx1);   // set 1 <- slab 2
    __syncthreads();
    // step i (even): compute buf0; write slab i+1 (set 0) -> buf1; then reload set 0 <- slab i+3
    // step i+1 (odd): compute buf1; write slab i+2 (set 1) -> buf0; then reload set 1 <- slab i+4
    // Slab indices are clamped (the last loads / writes are redundant but harmless); KT is even for every GEMM
    // here (2, 8, 16 slabs); the body is branch-free and the step pair is unrolled so both register sets are
    // statically indexed (hipcc would otherwise sink the loads into a conditional block next to their use).
    const int KTL = ABLATE == 3 ? 2 : KT;
    // (a static s_setprio 1 for the younger half of the workgroup -- the guide's two-waves-per-SIMD tip for attention loops -- was
    //  A/B-timed here: 1234 -> 1187 frames/s in flight, 1024 -> 978 one at a time, mlp0 40.0 -> 42.7 us event-timed; not kept)
    for (int i = 0; i < KTL; i += 2) {
        if constexpr (T::KS == 2) step_ks(buf0, buf1, i, min(i + 3, last), ra0, rb0, rx0);
        else if constexpr (QF) step_qf(buf0, buf1, i, min(i + 3, last), ra0, rb0, rx0);
        else step(buf0, buf1, i, min(i + 3, last), ra0, rb0, rx0);
        __syncthreads();
        if constexpr (T::KS == 2) step_ks(buf1, buf0, i + 1, min(i + 4, last), ra1, rb1, rx1);
        else if constexpr (QF) step_qf(buf1, buf0, i + 1, min(i + 4, last), ra1, rb1, rx1);
        else step(buf1, buf0, i + 1, min(i + 4, last), ra1, rb1, rx1);
        __syncthreads();
        if constexpr (Hooks::ENABLED) hooks->pair_end(i, acc[0][0]);
    }
}

// K split: add the accumulators of the wave groups (after the main loop, whose last barrier freed `smem`).  Every group ends
// with the full sums (group 0's value + group 1's value on both), so the epilogues need not know about the split: the waves
// of group 1 redo group 0's epilogue stores with identical values.  smem must hold THREADS * 16 floats per 32x32 tile.
template <class T>
__device__ __forceinline__ void ksplit_reduce(f32x16 (&acc)[T::TM][T::TN], float* smem) {
    if constexpr (T::KS == 2) {
        static_assert(T::TM * T::TN * T::THREADS * 16 <= T::SMEM_FLOATS, "the partial accumulators must fit the operand buffers");
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        const int kg = wave / T::WAVES_MN, wq = wave % T::WAVES_MN;
#pragma unroll
        for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < T::TN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    smem[((((kg * T::WAVES_MN + wq) * T::TM + tm) * T::TN + tn) * 16 + r) * 64 + lane] = acc[tm][tn][r];
        __syncthreads();
#pragma unroll
        for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < T::TN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float o = smem[(((((kg ^ 1) * T::WAVES_MN + wq) * T::TM + tm) * T::TN + tn) * 16 + r) * 64 + lane];
                    acc[tm][tn][r] = kg ? o + acc[tm][tn][r] : acc[tm][tn][r] + o;   // group 0 + group 1 on both
                }
        __syncthreads();
    }
}

// =====================================================================================================
// Split-bf16 variant of the main loop ("bf16x3", selected per call by GATSSPG_FLAG_PREC_BF16X3): every fp32 operand is
// x = x1 + x2 with x1 = RNE_bf16(x), x2 = RNE_bf16(x - x1); the product is a1*b1 + a1*b2 + a2*b1 on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation (3 x 32 cycles per 32x32x16 block instead of 8 x 64 for the f32 MFMA).
// tests/studies/split_bf16_study.py: conf within 5e-7 of the fp32 forward and every match identical at the headline shape.
//   A (weights): split ONCE at pack time into two bf16 planes [M][K] (gatsspg_pack_weights); a slab is copied
//     global -> LDS as 16-byte pieces, no VALU work.
//   B (activations, fp32 [K][N] in HBM): a thread owns ONE column and KPT = 4 or 8 consecutive k rows of the slab
//     (dword loads, coalesced across the wave), applies the optional per-row transform, splits with v_cvt_pk_bf16_f32
//     and writes its k run of both planes with one 8- or 16-byte LDS store each.
// LDS images (bf16): A planes [BM][40], B planes [BN][40] (k contiguous, 80-byte rows: the 16-byte fragment reads of
// 16 consecutive rows hit 16 distinct 4-bank groups).  Pipeline as in gemm_mainloop_ex: two register sets, prefetch
// distance 2, one barrier per slab.
// =====================================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned bf16_rne_bits(float x) {
    const unsigned u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
// two floats -> packed (hi-term) bf16 pair and packed (lo-term) bf16 pair; element 0 in the low half
__device__ __forceinline__ void bf16_split2(float a, float b, unsigned& hi, unsigned& lo) {
    const bf16x2 h = __builtin_convertvector((f32x2){a, b}, bf16x2);
    hi = __builtin_bit_cast(unsigned, h);
    const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xFFFF0000u);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){ra, rb}, bf16x2));
}

// The same two-term idea on IEEE fp16 terms (GATSSPG_FLAG_PREC_FP16X3 / _FP16X4): x ~ x1 + x2 with x1 = RNE_fp16(x), x2 = RNE_fp16(x - x1):
// 2 x 11 significand bits with signed remainders, i.e. a representation error <= 2^-23 |x| -- what rounding to fp32 itself costs is
// 2^-24 -- as long as x2 stays a normal fp16 number (|x| >~ 0.06; below that the absolute error is the fp16 subnormal spacing,
// 6e-8).  Both conversions SATURATE at +-65504 (MODE.FP16_OVFL, below): an out-of-range operand gives two terms that reach
// +-131008 instead of infinity and then NaN.  (Round-toward-zero conversions, one instruction per pair and saturating
// by themselves, were measured first: the one-sided first term doubles the remainder and the mode kept a near-tie flip that the
// RNE form does not have in the four-product mode -- tests/studies/split_bf16_study.py.)
//   fp16x3: a1 b1 + a1 b2 + a2 b1              three v_mfma_f32_32x32x16_f16 per 32x32x16 block: the matrix-pipe time of bf16x3
//   fp16x4: + a2 b2 (first, smallest)          four: the exact product of the split operands -- fp32-class
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
// MODE.FP16_OVFL (hwreg 1, bit 23): an overflowing fp16 RESULT is clamped to +-65504 instead of becoming infinity.  Set once per wave
// by every kernel that splits into fp16 terms (checked on the GPU: 1e5 -> 65504 + 34496, -3e38 -> -65504 - 65504); it replaces four
// v_med3_f32 per operand pair in the main loops (measured with the explicit clamps: mlp0 +1.5 us).
__device__ __forceinline__ void fp16_saturate_mode() { __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1); }
// requires fp16_saturate_mode() earlier in the wave
__device__ __forceinline__ void fp16_split2(float a, float b, unsigned& hi, unsigned& lo) {
    const f16x2 h = __builtin_convertvector((f32x2){a, b}, f16x2);
    hi = __builtin_bit_cast(unsigned, h);
    const f32x2 r = {a - (float)h[0], b - (float)h[1]};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}

template <class T>
struct Bf3Layout {
    static constexpr int KS = 40;                                       // bf16 per LDS row (32 + 8 pad)
    static constexpr int A_PLANE = T::BM * KS, B_PLANE = T::BN * KS;    // in bf16 elements
    static constexpr int STAGE = 2 * A_PLANE + 2 * B_PLANE;             // hi + lo of both operands
    static constexpr size_t SMEM_BYTES = 2 * (size_t)STAGE * 2;         // two stages
    static constexpr int A_PIECES = T::BM * 4 * 2 / T::THREADS;         // 16-byte pieces per thread per slab (both planes)
    static constexpr int KPT = BK * T::BN / T::THREADS;                 // consecutive k rows per thread (one B column)
    static_assert(KPT == 4 || KPT == 8, "a thread owns 4 or 8 consecutive k of one column");
    static_assert(A_PIECES >= 1 && ((T::BM * 4) % T::THREADS == 0 || T::THREADS % (T::BM * 4) == 0), "A plane of a piece is static");
    static_assert(T::BN % 64 == 0, "a wave covers 64 consecutive columns of one k group");
};

// a_hi(kt) / a_lo(kt): bf16 plane pointers of A slab kt (&A[row0][kt*32], row stride lda elements).
// b_slab(kt): fp32 pointer &B[kt*32][col0], row stride ldb.  x_mean / x_rstd / bxform: per-k-row aux values applied to
// the B values before the split (mlp.3: InstanceNorm + ReLU on the operand load).
// F16: the two planes of each operand hold fp16 terms (fp16_split2) and the products run on v_mfma_f32_32x32x16_f16; layouts,
// staging and pipeline are those of the bf16 form (16-bit elements either way).
// NP = 4 (F16 only): the lo x lo product is kept as well (fp16x4).
// BCol: tile column -> element offset from the B slab pointer (identity, or the image-patch map of the SuperPoint convolutions; the B
// loads of this loop are dword loads, so column-shifted / 4-byte-aligned views (T::BU) need nothing else).
template <class T, class AHi, class ALo, class BSlab, class XSlabA, class XSlabB, class BXform, bool HAS_AUX, class Hooks = NoHooks,
          bool F16 = false, int NP = 3, class BCol = IdentityCol>
__device__ __forceinline__ void gemm_mainloop_bf3_ex(f32x16 (&acc)[T::TM][T::TN], unsigned short* smem, int KT, AHi a_hi, ALo a_lo,
                                                     int lda, BSlab b_slab, int ldb, XSlabA x_mean, XSlabB x_rstd, BXform bxform,
                                                     Hooks* hooks = nullptr, BCol bcolmap = BCol()) {
    static_assert(!T::AKM, "row-major A");
    using LY = Bf3Layout<T>;
    if constexpr (Hooks::ENABLED)
        static_assert(T::BN == 64 && T::THREADS == 512 && T::TM == 1 && T::TN == 1 && LY::KPT == 4, "fold hooks: 64-column tile, 8 waves");
    constexpr int BM = T::BM, BN = T::BN, TM = T::TM, TN = T::TN, KS = LY::KS, AP = LY::A_PIECES, KPT = LY::KPT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / T::WN, wn = wave % T::WN;
    const int half = lane >> 5, l31 = lane & 31;
    // A pieces: idx -> (plane, row, 8-element group); the plane of piece p is a compile-time constant
    unsigned a_goff[AP];
    int a_soff[AP];
#pragma unroll
    for (int p = 0; p < AP; ++p) {
        const int idx = p * T::THREADS + tid;
        const int rem = idx % (BM * 4), r = rem >> 2, c8 = rem & 3;
        a_goff[p] = 2u * (unsigned)(r * lda + c8 * 8);
        a_soff[p] = r * KS + c8 * 8;
    }
    // plane (hi / lo) of A piece p: a compile-time constant when a plane is a whole number of thread rounds; with more threads than
    // pieces per plane (64-row tile on 8 waves) the waves split between the planes (wave-uniform)
    auto a_plane_lo = [&](int p) -> bool {
        if constexpr ((BM * 4) % T::THREADS == 0) return (p * T::THREADS) / (BM * 4) != 0;
        else return __builtin_amdgcn_readfirstlane((p * T::THREADS + tid) / (BM * 4)) != 0;
    };
    // B: one column, KPT consecutive k; the k group is wave-uniform (BN is a multiple of 64)
    const int bcol = tid % BN;
    const int k0 = __builtin_amdgcn_readfirstlane(tid / BN) * KPT;
    unsigned b_goff[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) b_goff[j] = 4u * (unsigned)((k0 + j) * ldb + bcolmap(bcol));
    const int b_soff = bcol * KS + k0;

    u32x4 ra0[AP], ra1[AP];
    float rb0[KPT], rb1[KPT];
    float2 rx0[KPT], rx1[KPT];
    auto gload = [&](int kt, u32x4(&ra)[AP], float(&rb)[KPT], float2(&rx)[KPT]) {
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            const bool lo = a_plane_lo(p);
            const unsigned short* base = lo ? a_lo(kt) : a_hi(kt);
            ra[p] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(base) + a_goff[p]);
        }
        const float* bb = b_slab(kt);
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            rb[j] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(bb) + b_goff[j]);
            if constexpr (HAS_AUX) rx[j] = make_float2(x_mean(kt)[k0 + j], x_rstd(kt)[k0 + j]);
        }
    };
    // kw: index of the slab being written (fold hooks: its Qf values enter the denominators before they are split)
    auto swrite = [&](unsigned short* stage, int kw, const u32x4(&ra)[AP], const float(&rb)[KPT], const float2(&rx)[KPT]) {
        if constexpr (Hooks::ENABLED) {
            if (kw >= Hooks::SPLIT && kw < KT) hooks->partial(kw - Hooks::SPLIT, rb[0], rb[1], rb[2], rb[3]);
        }
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            const bool lo = a_plane_lo(p);
            *reinterpret_cast<u32x4*>(stage + (lo ? LY::A_PLANE : 0) + a_soff[p]) = ra[p];
        }
        unsigned short* Bhi = stage + 2 * LY::A_PLANE;
        unsigned short* Blo = Bhi + LY::B_PLANE;
        unsigned h[KPT / 2], l[KPT / 2];
#pragma unroll
        for (int j = 0; j < KPT; j += 2) {
            float v0 = rb[j], v1 = rb[j + 1];
            if constexpr (HAS_AUX) {
                v0 = bxform(v0, rx[j]);
                v1 = bxform(v1, rx[j + 1]);
            }
            if constexpr (F16) fp16_split2(v0, v1, h[j / 2], l[j / 2]);
            else bf16_split2(v0, v1, h[j / 2], l[j / 2]);
        }
        if constexpr (KPT == 8) {
            *reinterpret_cast<u32x4*>(Bhi + b_soff) = (u32x4){h[0], h[1], h[2], h[3]};
            *reinterpret_cast<u32x4*>(Blo + b_soff) = (u32x4){l[0], l[1], l[2], l[3]};
        } else {
            *reinterpret_cast<u32x2*>(Bhi + b_soff) = (u32x2){h[0], h[1]};
            *reinterpret_cast<u32x2*>(Blo + b_soff) = (u32x2){l[0], l[1]};
        }
    };
    auto compute = [&](const unsigned short* stage) {
        const unsigned short* Ahi = stage;
        const unsigned short* Alo = stage + LY::A_PLANE;
        const unsigned short* Bhi = stage + 2 * LY::A_PLANE;
        const unsigned short* Blo = Bhi + LY::B_PLANE;
        // the fragments of BOTH k16 steps of the slab are requested first (one exposed LDS latency per slab instead of one per
        // MFMA group), then 2 x 3 products per 32x32 block, small terms first
        bf16x8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int ko = 16 * s + 8 * half;                     // the same k assignment for A and B fragments
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const int row = (wm * TM + tm) * 32 + l31;
                ah[s][tm] = *reinterpret_cast<const bf16x8*>(Ahi + row * KS + ko);
                al[s][tm] = *reinterpret_cast<const bf16x8*>(Alo + row * KS + ko);
            }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int col = (wn * TN + tn) * 32 + l31;
                bh[s][tn] = *reinterpret_cast<const bf16x8*>(Bhi + col * KS + ko);
                bl[s][tn] = *reinterpret_cast<const bf16x8*>(Blo + col * KS + ko);
            }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    if constexpr (F16) {
                        auto h8 = [](bf16x8 v) { return __builtin_bit_cast(f16x8, v); };   // the registers hold fp16 terms in this mode
                        if constexpr (NP == 4)
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(al[s][tm]), h8(bl[s][tn]), acc[tm][tn], 0, 0, 0);
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(al[s][tm]), h8(bh[s][tn]), acc[tm][tn], 0, 0, 0);
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(ah[s][tm]), h8(bl[s][tn]), acc[tm][tn], 0, 0, 0);
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(ah[s][tm]), h8(bh[s][tn]), acc[tm][tn], 0, 0, 0);
                    } else {
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[s][tm], bh[s][tn], acc[tm][tn], 0, 0, 0);
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s][tm], bl[s][tn], acc[tm][tn], 0, 0, 0);
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s][tm], bh[s][tn], acc[tm][tn], 0, 0, 0);
                    }
                }
    };
    unsigned short* buf0 = smem;
    unsigned short* buf1 = smem + LY::STAGE;
    const int last = KT - 1;
    gload(0, ra1, rb1, rx1);              // slab 0 through set 1, slab 1 (set 0) requested together with it (see gemm_mainloop_ex)
    gload(min(1, last), ra0, rb0, rx0);
    swrite(buf0, 0, ra1, rb1, rx1);
    gload(min(2, last), ra1, rb1, rx1);
    __syncthreads();
    // step i: MFMAs of slab i, slab i+1 (in registers since step i-2) is split and written to the free buffer, the freed
    // registers start loading slab i+3
    for (int i = 0; i < KT; i += 2) {
        compute(buf0);
        swrite(buf1, i + 1, ra0, rb0, rx0);
        asm volatile("" ::: "memory");
        gload(min(i + 3, last), ra0, rb0, rx0);
        __syncthreads();
        compute(buf1);
        swrite(buf0, i + 2, ra1, rb1, rx1);
        asm volatile("" ::: "memory");
        gload(min(i + 4, last), ra1, rb1, rx1);
        __syncthreads();
        if constexpr (Hooks::ENABLED) hooks->pair_end(i, acc[0][0]);
    }
}

// =====================================================================================================
// Six-term split-bf16 ("bf16x6"): x = x1 + x2 + x3 EXACTLY (three bf16 planes hold the 24 mantissa bits of an fp32 value),
// product = a1b1 + a1b2 + a2b1 + a2b2 + a1b3 + a3b1; the dropped terms (a2b3, a3b2, a3b3) are <= 2^-24 |ab| -- the size of
// ONE fp32 rounding of the product -- and the sum is accumulated in fp32 by the MFMA exactly like the f32 path does.
// 6 x 32 cycles per 32x32x16 block against 8 x 64 for v_mfma_f32_32x32x2_f32.
// LDS images: three planes per operand, rows of 32 bf16 (64 bytes, NO pad: with the 40-element rows of the three-term loop
// two stages of six planes would not leave room for two workgroups per CU); the 16-byte chunk c of row r is stored at chunk
// c ^ ((r >> 2) & 3), so the fragment reads of 16 consecutive rows still cover the 16 four-bank groups once.
// =====================================================================================================
__device__ __forceinline__ void bf16_split3(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
    p0 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf16x2));
    const float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xFFFF0000u);
    p1 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){ra, rb}, bf16x2));
    const float sa = ra - __uint_as_float(p1 << 16), sb = rb - __uint_as_float(p1 & 0xFFFF0000u);
    p2 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){sa, sb}, bf16x2));
}

template <class T>
struct Bf6Layout {
    static constexpr int KS = 32;                                       // bf16 per LDS row (swizzled, no pad)
    static constexpr int A_PLANE = T::BM * KS, B_PLANE = T::BN * KS;    // in bf16 elements
    static constexpr int STAGE = 3 * A_PLANE + 3 * B_PLANE;
    static constexpr size_t SMEM_BYTES = 2 * (size_t)STAGE * 2;         // two stages
    static constexpr int A_PIECES = T::BM * 4 * 3 / T::THREADS;         // 16-byte pieces per thread per slab (three planes)
    static constexpr int KPT = BK * T::BN / T::THREADS;                 // consecutive k rows per thread (one B column)
    static_assert(KPT == 4 || KPT == 8, "a thread owns 4 or 8 consecutive k of one column");
    static_assert(A_PIECES >= 1 && ((T::BM * 4) % T::THREADS == 0 || T::THREADS % (T::BM * 4) == 0), "A plane of a piece is static");
    static_assert(T::BN % 64 == 0, "a wave covers 64 consecutive columns of one k group");
};
__device__ __forceinline__ int bf6_swz(int row, int chunk) { return chunk ^ ((row >> 2) & 3); }

// a_pl(kt, plane): bf16 plane pointer of A slab kt (&A_plane[row0][kt*32], row stride lda elements); the rest as in
// gemm_mainloop_bf3_ex.
template <class T, class APlane, class BSlab, class XSlabA, class XSlabB, class BXform, bool HAS_AUX, class Hooks = NoHooks>
__device__ __forceinline__ void gemm_mainloop_bf6_ex(f32x16 (&acc)[T::TM][T::TN], unsigned short* smem, int KT, APlane a_pl, int lda,
                                                     BSlab b_slab, int ldb, XSlabA x_mean, XSlabB x_rstd, BXform bxform,
                                                     Hooks* hooks = nullptr) {
    static_assert(!T::AKM && !T::BU, "row-major A, aligned B");
    using LY = Bf6Layout<T>;
    if constexpr (Hooks::ENABLED)
        static_assert(T::BN == 64 && T::THREADS == 512 && T::TM == 1 && T::TN == 1 && LY::KPT == 4, "fold hooks: 64-column tile, 8 waves");
    constexpr int BM = T::BM, BN = T::BN, TM = T::TM, TN = T::TN, KS = LY::KS, AP = LY::A_PIECES, KPT = LY::KPT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / T::WN, wn = wave % T::WN;
    const int half = lane >> 5, l31 = lane & 31;
    unsigned a_goff[AP];
    int a_soff[AP];
#pragma unroll
    for (int p = 0; p < AP; ++p) {
        const int idx = p * T::THREADS + tid;
        const int rem = idx % (BM * 4), r = rem >> 2, c8 = rem & 3;
        a_goff[p] = 2u * (unsigned)(r * lda + c8 * 8);
        a_soff[p] = r * KS + bf6_swz(r, c8) * 8;
    }
    const int bcol = tid % BN;
    const int k0 = __builtin_amdgcn_readfirstlane(tid / BN) * KPT;
    unsigned b_goff[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) b_goff[j] = 4u * (unsigned)((k0 + j) * ldb + bcol);
    const int b_soff = bcol * KS + bf6_swz(bcol, k0 >> 3) * 8 + (k0 & 7);

    u32x4 ra0[AP], ra1[AP];
    float rb0[KPT], rb1[KPT];
    float2 rx0[KPT], rx1[KPT];
    auto gload = [&](int kt, u32x4(&ra)[AP], float(&rb)[KPT], float2(&rx)[KPT]) {
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            const int plane = (p * T::THREADS) / (BM * 4);
            ra[p] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(a_pl(kt, plane)) + a_goff[p]);
        }
        const float* bb = b_slab(kt);
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            rb[j] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(bb) + b_goff[j]);
            if constexpr (HAS_AUX) rx[j] = make_float2(x_mean(kt)[k0 + j], x_rstd(kt)[k0 + j]);
        }
    };
    auto swrite = [&](unsigned short* stage, int kw, const u32x4(&ra)[AP], const float(&rb)[KPT], const float2(&rx)[KPT]) {
        if constexpr (Hooks::ENABLED) {
            if (kw >= Hooks::SPLIT && kw < KT) hooks->partial(kw - Hooks::SPLIT, rb[0], rb[1], rb[2], rb[3]);
        }
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            const int plane = (p * T::THREADS) / (BM * 4);
            *reinterpret_cast<u32x4*>(stage + plane * LY::A_PLANE + a_soff[p]) = ra[p];
        }
        unsigned short* B0 = stage + 3 * LY::A_PLANE;
        unsigned q0[KPT / 2], q1[KPT / 2], q2[KPT / 2];
#pragma unroll
        for (int j = 0; j < KPT; j += 2) {
            float v0 = rb[j], v1 = rb[j + 1];
            if constexpr (HAS_AUX) {
                v0 = bxform(v0, rx[j]);
                v1 = bxform(v1, rx[j + 1]);
            }
            bf16_split3(v0, v1, q0[j / 2], q1[j / 2], q2[j / 2]);
        }
        if constexpr (KPT == 8) {
            *reinterpret_cast<u32x4*>(B0 + b_soff) = (u32x4){q0[0], q0[1], q0[2], q0[3]};
            *reinterpret_cast<u32x4*>(B0 + LY::B_PLANE + b_soff) = (u32x4){q1[0], q1[1], q1[2], q1[3]};
            *reinterpret_cast<u32x4*>(B0 + 2 * LY::B_PLANE + b_soff) = (u32x4){q2[0], q2[1], q2[2], q2[3]};
        } else {
            *reinterpret_cast<u32x2*>(B0 + b_soff) = (u32x2){q0[0], q0[1]};
            *reinterpret_cast<u32x2*>(B0 + LY::B_PLANE + b_soff) = (u32x2){q1[0], q1[1]};
            *reinterpret_cast<u32x2*>(B0 + 2 * LY::B_PLANE + b_soff) = (u32x2){q2[0], q2[1]};
        }
    };
    auto gload_a = [&](int kt, u32x4(&ra)[AP]) {
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            const int plane = (p * T::THREADS) / (BM * 4);
            ra[p] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(a_pl(kt, plane)) + a_goff[p]);
        }
    };
    auto gload_b = [&](int kt, float(&rb)[KPT], float2(&rx)[KPT]) {
        const float* bb = b_slab(kt);
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            rb[j] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(bb) + b_goff[j]);
            if constexpr (HAS_AUX) rx[j] = make_float2(x_mean(kt)[k0 + j], x_rstd(kt)[k0 + j]);
        }
    };
    // fragments of one k16 step: k = 16 s + 8 half .. + 7, the same k assignment for A and B
    auto read_frags = [&](const unsigned short* stage, int s, bf16x8 (&af)[3][TM], bf16x8 (&bf)[3][TN]) {
        const unsigned short* Ap = stage;
        const unsigned short* Bp = stage + 3 * LY::A_PLANE;
        const int chunk = 2 * s + half;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int row = (wm * TM + tm) * 32 + l31;
            const int off = row * KS + bf6_swz(row, chunk) * 8;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) af[pl][tm] = *reinterpret_cast<const bf16x8*>(Ap + pl * LY::A_PLANE + off);
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int col = (wn * TN + tn) * 32 + l31;
            const int off = col * KS + bf6_swz(col, chunk) * 8;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) bf[pl][tn] = *reinterpret_cast<const bf16x8*>(Bp + pl * LY::B_PLANE + off);
        }
    };
    // term pair g of one k16 step, small terms first: (a3 b1, a1 b3), (a2 b2, a2 b1), (a1 b2, a1 b1)
    auto mfma_pair = [&](const bf16x8 (&af)[3][TM], const bf16x8 (&bf)[3][TN], int g) {
        const int pa0 = g == 0 ? 2 : g == 1 ? 1 : 0, pb0 = g == 0 ? 0 : 1;
        const int pa1 = g == 0 ? 0 : g == 1 ? 1 : 0, pb1 = g == 0 ? 2 : 0;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[pa0][tm], bf[pb0][tn], acc[tm][tn], 0, 0, 0);
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[pa1][tm], bf[pb1][tn], acc[tm][tn], 0, 0, 0);
            }
    };
    // One step = one read phase (all twelve fragments), twelve MFMAs, then the split + LDS write of slab i+1 and the global loads
    // of slab i+3.  (Placing the second-step reads, the LDS writes and the loads BETWEEN the MFMA pairs, as the fp32 loop does,
    // was measured on this loop: mlp0 34.9 vs 32.5 us, 1337 vs 1435 frames/s -- the bf16 MFMAs are too short to hide them;
    // write + loads BEFORE the MFMAs: 1137 frames/s -- the write then waits for loads that have had one step, not two, to land.)
    auto step = [&](const unsigned short* cur, unsigned short* nxt, int kw, int kt_load, u32x4(&ra)[AP], float(&rb)[KPT], float2(&rx)[KPT]) {
        bool two_reads = false;
        if constexpr (Hooks::ENABLED) two_reads = kw > Hooks::SPLIT;   // the slab being computed (kw - 1) belongs to the head phase
        if (two_reads) {
            // head phase of the attention fold: a second accumulator (16 registers) is alive, so the fragments of the second
            // k16 step re-use the registers of the first (read -> 6 MFMAs -> read -> 6 MFMAs) instead of being fetched up front
            // -- the 128-VGPR budget of two workgroups per CU has no room for both (spilled 28 registers: mlp0 29.6 -> 36.9 us)
            bf16x8 af[3][TM], bf[3][TN];
            read_frags(cur, 0, af, bf);
#pragma unroll
            for (int g = 0; g < 3; ++g) mfma_pair(af, bf, g);
            __builtin_amdgcn_sched_barrier(0);
            read_frags(cur, 1, af, bf);
#pragma unroll
            for (int g = 0; g < 3; ++g) mfma_pair(af, bf, g);
        } else {
            bf16x8 af0[3][TM], bf0[3][TN], af1[3][TM], bf1[3][TN];
            read_frags(cur, 0, af0, bf0);
            read_frags(cur, 1, af1, bf1);
#pragma unroll
            for (int g = 0; g < 3; ++g) mfma_pair(af0, bf0, g);
#pragma unroll
            for (int g = 0; g < 3; ++g) mfma_pair(af1, bf1, g);
        }
        swrite(nxt, kw, ra, rb, rx);
        asm volatile("" ::: "memory");
        gload_a(kt_load, ra);
        gload_b(kt_load, rb, rx);
    };
    unsigned short* buf0 = smem;
    unsigned short* buf1 = smem + LY::STAGE;
    const int last = KT - 1;
    gload(0, ra1, rb1, rx1);
    gload(min(1, last), ra0, rb0, rx0);
    swrite(buf0, 0, ra1, rb1, rx1);
    gload(min(2, last), ra1, rb1, rx1);
    __syncthreads();
    for (int i = 0; i < KT; i += 2) {
        step(buf0, buf1, i + 1, min(i + 3, last), ra0, rb0, rx0);
        __syncthreads();
        step(buf1, buf0, i + 2, min(i + 4, last), ra1, rb1, rx1);
        __syncthreads();
        if constexpr (Hooks::ENABLED) hooks->pair_end(i, acc[0][0]);
    }
}

struct NoXform1 {
    __device__ __forceinline__ float operator()(float v, float2) const { return v; }
};
template <class T, class AHi, class ALo, class BSlab, class Hooks = NoHooks, bool F16 = false, int NP = 3, class BCol = IdentityCol>
__device__ __forceinline__ void gemm_mainloop_bf3(f32x16 (&acc)[T::TM][T::TN], unsigned short* smem, int KT, AHi a_hi, ALo a_lo,
                                                  int lda, BSlab b_slab, int ldb, Hooks* hooks = nullptr, BCol bcolmap = BCol()) {
    auto nox = [](int) { return static_cast<const float*>(nullptr); };
    gemm_mainloop_bf3_ex<T, AHi, ALo, BSlab, decltype(nox), decltype(nox), NoXform1, false, Hooks, F16, NP, BCol>(
        acc, smem, KT, a_hi, a_lo, lda, b_slab, ldb, nox, nox, NoXform1(), hooks, bcolmap);
}

template <class T, class APlane, class BSlab, class Hooks = NoHooks>
__device__ __forceinline__ void gemm_mainloop_bf6(f32x16 (&acc)[T::TM][T::TN], unsigned short* smem, int KT, APlane a_pl, int lda,
                                                  BSlab b_slab, int ldb, Hooks* hooks = nullptr) {
    auto nox = [](int) { return static_cast<const float*>(nullptr); };
    gemm_mainloop_bf6_ex<T, APlane, BSlab, decltype(nox), decltype(nox), NoXform1, false, Hooks>(acc, smem, KT, a_pl, lda, b_slab, ldb, nox,
                                                                                               nox, NoXform1(), hooks);
}

// convenience wrapper without per-row aux / transform
template <class T, class ASlab, class BSlab, int ABLATE = 0, class BCol = IdentityCol, class Hooks = NoHooks, int QF = 0>
__device__ __forceinline__ void gemm_mainloop(f32x16 (&acc)[T::TM][T::TN], float* smem, int KT, ASlab a_slab, int lda,
                                              BSlab b_slab, int ldb, BCol bcol = BCol(), Hooks* hooks = nullptr) {
    auto nox = [](int) { return static_cast<const float*>(nullptr); };
    gemm_mainloop_ex<T, ASlab, BSlab, decltype(nox), decltype(nox), NoXform, false, ABLATE, BCol, Hooks, QF>(
        acc, smem, KT, a_slab, lda, b_slab, ldb, nox, nox, NoXform(), bcol, hooks);
}

// Epilogue helper: the accumulator tile leaves through LDS as 16-byte stores (16 lanes cover one 256-byte row segment
// of a 64-column tile) instead of 4-byte stores straight from the MFMA layout (measured on mlp0: -3 %).
// One LDS-DMA request: lane i moves 16 bytes from its global address to lds + 16 i (global_load_lds_dwordx4; M0 = the wave-uniform LDS base).
__device__ __forceinline__ void glds16(const void* g, void* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds, 16, 0,
                                     0);
}
// A lane's 16 bias values per 32-row MFMA tile (rows 8 k + 4 half + 0..3) from an LDS table of the workgroup's BM bias values.  The table is
// filled by HALF an LDS-DMA piece (32 lanes x 16 bytes) requested at kernel entry -- older than every operand load of the main loop, so the
// loop's own waits and barriers cover and publish it -- and read behind the loop: no bias registers across the loop, no per-lane global loads
// in front of the first operand requests (frames in flight: +3..4 % on the split loop, profiles/r04_ab_live_bias_table.txt).
template <class T>
__device__ __forceinline__ void read_bias16(const float* tab, int wm, int half, float (&bias)[T::TM][16]) {
#pragma unroll
    for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const vf4 b4 = *reinterpret_cast<const vf4*>(tab + (wm * T::TM + tm) * 32 + 8 * k + 4 * half);
            bias[tm][4 * k + 0] = b4[0]; bias[tm][4 * k + 1] = b4[1]; bias[tm][4 * k + 2] = b4[2]; bias[tm][4 * k + 3] = b4[3];
        }
}

// f(row, v) is applied on the way in; dst points at the tile origin, ld is its row stride; both 16-byte aligned.
// smem must be free (the main loop ends on a barrier) and hold BM * (BN + 4) floats.
template <class T, class F>
__device__ __forceinline__ void store_tile_via_lds(const f32x16 (&acc)[T::TM][T::TN], float* smem, float* dst, int ld, F f) {
    constexpr int TS = T::BN + 4;
    static_assert(T::BM * TS <= T::SMEM_FLOATS, "staging tile must fit the operand buffers");
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) % T::WAVES_MN;
    const int wm = wave / T::WN, wn = wave % T::WN, half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < T::TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (wm * T::TM + tm) * 32 + mfma_row(r, half);
                smem[row * TS + (wn * T::TN + tn) * 32 + l31] = f(row, acc[tm][tn][r]);
            }
    __syncthreads();
#pragma unroll
    for (int idx = tid; idx < T::BM * (T::BN / 4); idx += T::THREADS) {   // compile-time trip count: all LDS reads first, then the stores
        const int row = idx / (T::BN / 4), c4 = (idx % (T::BN / 4)) * 4;
        *reinterpret_cast<vf4*>(dst + (size_t)row * ld + c4) = *reinterpret_cast<const vf4*>(smem + row * TS + c4);
    }
}

// A plain output tile straight from the accumulators: in the 32 x 32 C layout a lane's 16 values of one product sit in ONE column (lane & 31)
// and 16 rows, so each dword store instruction covers two rows x 32 consecutive columns = two full 128-byte lines -- no LDS round trip and
// no barrier in front of the stores (store_tile_via_lds: 16 scalar LDS writes per 32 x 32 block, a barrier, row reads, 16-byte stores).
template <class T, class F>
__device__ __forceinline__ void store_tile_regs(const f32x16 (&acc)[T::TM][T::TN], float* dst, int ld, F f) {
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) % T::WAVES_MN;
    const int wm = wave / T::WN, wn = wave % T::WN, half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < T::TN; ++tn) {
            float* d = dst + (wn * T::TN + tn) * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (wm * T::TM + tm) * 32 + mfma_row(r, half);
                d[(size_t)row * ld] = f(row, acc[tm][tn][r]);
            }
        }
}

template <int TM, int TN>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[TM][TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}

// ---------------------------------------------------------------------------------------------------------------------
// InstanceNorm statistics fused into the mlp.0 launch (replaces the stat_final_kernel launch between mlp.0 and mlp.3): the LAST
// workgroup of a (segment, row tile) to finish turns the per-tile partials of its rows into mean / rstd.
//   * every workgroup leaves its partials with write-through (agent-scope) stores, waits for them (vmcnt(0) + barrier) and draws a
//     ticket from the (segment, row tile) counter with one relaxed agent-scope atomic -- no L2 write-back fence (guide G16, sc1 form);
//   * the workgroup that draws the last ticket reads ALL partials of its rows with agent-scope loads and merges them exactly like
//     stat_final_kernel did (Chan's formula in double precision, tile ranges summed in tile order, ranges combined in range order):
//     the result does not depend on WHICH workgroup is last nor on the order of arrival -- run-to-run bit-identical;
//   * the counters are zeroed by kv_final_kernel (always enqueued before mlp.0 on the same segments) and reset by the reducer.
// smem: 2 * THREADS doubles + one int, free at the call.  stats: [seg][2][512] (mean, 1 / sqrt(var + 1e-5)), GATs_SuperGlue.py:126.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int STATCNT_PER_SEG = 8;   // row tiles of mlp.0 per segment (512 / 64 at most)
__device__ __forceinline__ void stat_partial_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// PENDING: vector-memory instructions this thread issued AFTER its partial stores (the tile's own stores), which may stay in flight
template <class T, int PENDING>
__device__ __forceinline__ void stat_last_block(const float* statpart, float* stats, int* cnt, const ColLayout& L, const TileSeg& ts, int rt,
                                                void* smem_v) {
    constexpr int BM = T::BM, PARTS = T::THREADS / BM;
    static_assert(T::THREADS % BM == 0 && 512 / BM <= STATCNT_PER_SEG, "row tile / counter layout");
    double* red = reinterpret_cast<double*>(smem_v);   // [2][PARTS][BM]
    int* flag = reinterpret_cast<int*>(red + 2 * PARTS * BM);
    const int tid = threadIdx.x;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PENDING) : "memory");   // this wave's partial stores (and everything before them) have been acknowledged
    __syncthreads();                                                  // ... every wave's; nobody still uses the staged tile in LDS
    if (tid == 0) {
        const int nwg = (ts.side ? L.n2p : L.n1p) / T::BN;
        const int old = __hip_atomic_fetch_add(cnt + ts.seg * STATCNT_PER_SEG + rt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = old == nwg - 1;
    }
    __syncthreads();
    if (!*flag) return;   // block-uniform
    const int row = tid % BM, part = tid / BM;
    const int t0 = (ts.frame * L.np + (ts.side ? L.n1p : 0)) / MLP0_BN;
    const int nt = (ts.side ? L.n2p : L.n1p) / MLP0_BN;
    const int n = ts.side ? L.n2 : L.n1;
    const int per = (nt + PARTS - 1) / PARTS;
    const int tb = part * per, te = min(nt, tb + per);
    const int ch = rt * BM + row;
    double S = 0.0, QP = 0.0;
    constexpr int CH = 32;   // tiles per round trip: this workgroup is the last one running in its group, so latency is all that counts
    for (int tt = tb; tt < te; tt += CH) {   // 2 x 32 loads in flight at a time on clamped addresses
        float xs[CH], xm[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const size_t tile = (size_t)(t0 + min(tt + u, nt - 1));
            xs[u] = __hip_atomic_load(statpart + (tile * 2 + 0) * 512 + ch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            xm[u] = __hip_atomic_load(statpart + (tile * 2 + 1) * 512 + ch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int t = tt + u;
            const int nv = min(MLP0_BN, n - t * MLP0_BN);   // real columns of tile t of this segment (<= 0: pad-only tile)
            if (t < te && nv > 0) {
                const double st = (double)xs[u], mt = (double)xm[u];
                const double inv = nv == MLP0_BN ? 1.0 / MLP0_BN : 1.0 / nv;
                S += st;
                QP += mt + st * st * inv;
            }
        }
    }
    red[(0 * PARTS + part) * BM + row] = S;
    red[(1 * PARTS + part) * BM + row] = QP;
    __syncthreads();
    if (part == 0) {
        S = red[row];
        QP = red[PARTS * BM + row];
#pragma unroll
        for (int p = 1; p < PARTS; ++p) {
            S += red[p * BM + row];
            QP += red[(PARTS + p) * BM + row];
        }
        const double mean = S / n;
        double var = (QP - S * mean) / n;
        if (var < 0.0) var = 0.0;
        stats[((size_t)ts.seg * 2 + 0) * 512 + ch] = (float)mean;
        stats[((size_t)ts.seg * 2 + 1) * 512 + ch] = (float)(1.0 / sqrt(var + 1e-5));
    }
    if (tid == 0) __hip_atomic_store(cnt + ts.seg * STATCNT_PER_SEG + rt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// XCD-aware block -> (row tile, column tile) map (cdna guide T1): workgroup id g is dispatched
// to XCD g % 8; give each XCD whole column tiles and walk that tile's row tiles back to back so
// the B panel [K x BN] is fetched into one XCD's L2 once and re-used by all M/BM row tiles.
// Launch with grid = 8 * MT * ceil(NT / 8); returns false for the padding blocks.
__device__ __forceinline__ bool xcd_tile_map(int MT, int NT, int& rt, int& ct) {
    const int g = blockIdx.x;
    const int xcd = g & 7, slot = g >> 3;
    rt = slot % MT;
    ct = (slot / MT) * 8 + xcd;
    return ct < NT;
}
inline int xcd_grid(int MT, int NT) { return 8 * MT * ((NT + 7) / 8); }
// The same with an XCD owning GROUPS of 2^gs consecutive column tiles: kernels whose column tiles are half as wide as a neighbour kernel's
// (qkv_kv / mlp3 on 64 columns beside mlp0_sp on 128) then keep a column range on the XCD whose L2 its producer wrote it into.
__device__ __forceinline__ bool xcd_tile_map_g(int MT, int NT, int gs, int& rt, int& ct) {
    const int g = blockIdx.x;
    const int xcd = g & 7, slot = g >> 3;
    rt = slot % MT;
    const int cs = slot / MT;
    ct = ((((cs >> gs) << 3) + xcd) << gs) + (cs & ((1 << gs) - 1));
    return ct < NT;
}
inline int xcd_grid_g(int MT, int NT, int gs) { return (8 * MT * (((NT + (8 << gs) - 1) / (8 << gs)))) << gs; }

__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x : expm1f(x); }
// same values, branch-free: both sides are evaluated and selected (epilogues that apply elu to 16-32 accumulator values per
// lane: a divergent branch per value serialises whatever sits next to it)
__device__ __forceinline__ float elu1_select(float x) {
    const float e = expm1f(fminf(x, 0.f));
    return x > 0.f ? x : e;
}

}  // namespace gatsspg
