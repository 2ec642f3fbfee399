// Split-16-bit GEMM main loop for gfx950, second form (round 4): C[BM x BN] += A[BM x K] * B[K x BN] with every fp32 operand
// evaluated as a sum of 16-bit terms on v_mfma_f32_32x32x16_{f16,bf16} (the arithmetics of gemm_f32_mfma.h: bf16x3 / bf16x6 /
// fp16x3 / fp16x4), restructured around what bounded the first form (profiles/r03_pmc_fp16x4_sq_lds.txt: LDS busy 44 % of the launch,
// 30 % of that bank conflicts, matrix pipe 29 %, the phases adding up instead of overlapping):
//
//   * BOTH operand slabs arrive by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass, no VALU in the
//     staging path): the pre-split weight planes as 64-byte rows with the 16-byte chunk c of row r at chunk c ^ ((r >> 2) & 3) --
//     the swizzle is applied on the per-lane GLOBAL address, the LDS image of a piece is lane-linear as the DMA requires -- and the
//     activation slab as RAW fp32 [32 k][BN] rows;
//   * a wave owns a 64-row x 32-column output tile (TM = 2): its B fragment (8 consecutive k of ONE column per lane) is read from
//     the fp32 image with conflict-free ds_read_b32, SPLIT IN REGISTERS into the 16-bit planes and fed to the MFMAs directly -- the
//     B planes never exist in LDS; per 16 MFMAs a wave reads 8 x 16 B of A fragments and 16 x 4 B of B (first form: 16 x 16 B);
//   * a ring of NST stages; the slab NST steps ahead is requested right after the barrier that frees its stage and waited for with
//     a COUNTED vmcnt one or two steps later (raw s_barrier: __syncthreads() would drain the DMA queue);
//   * the barrier of step i sits BETWEEN the two k16 halves of the step's MFMAs, and the fragments of each half are read one half
//     step ahead into a second register set, so no wave waits for LDS latency right after a barrier;
//   * the K loop is fully unrolled (KT is 8 or 16): stage addresses, wait counts, the tail and the attention-fold phases are
//     compile-time.
//
// fp16 modes: operands are pre-scaled by exact powers of two before the split (weights: per matrix at pack time, to a maximum in
// [2^13, 2^14); activations: 2^4 in the loop; the message operator: by the source count, kv_final_kernel) and the accumulators scaled
// back in the epilogue, so that the SECOND fp16 term stays a normal number down to |x| ~ 2^-7 (|w| ~ 2^-24 max|W|): without it the
// second term of every operand below 2^-3 is an fp16 subnormal with a fixed 2^-25 absolute error (round-3 advisor finding).
#pragma once
#include <type_traits>

#include "gemm_f32_mfma.h"

namespace gatsspg {

template <int I>
using IC = std::integral_constant<int, I>;
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(IC<B>{});
        static_for<B + 1, E>(f);
    }
}

// (glds16: gemm_f32_mfma.h)
// counted wait for this wave's LDS-DMA pieces + its own LDS reads, then the workgroup barrier (no vmcnt(0) drain)
template <int VM>
__device__ __forceinline__ void wait_dma_barrier() {
    static_assert(VM >= 0 && VM < 64, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(VM) : "memory");
}

// the same, executed only by the waves with `on` != 0 (wave-uniform): a branch INSIDE the statement, so the compiler's control flow
// (and with it register allocation across the step) stays a straight line
template <int VM>
__device__ __forceinline__ void wait_dma_barrier_if(int on) {
    static_assert(VM >= 0 && VM < 64, "vmcnt immediate");
    asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_waitcnt vmcnt(%1) lgkmcnt(0)\n\ts_barrier\n1:" ::"s"(__builtin_amdgcn_readfirstlane(on)),
                 "n"(VM)
                 : "memory", "scc");
}

constexpr int FP16_ACT_SCALE_LOG2 = 4;   // activations are multiplied by 2^4 before the fp16 split (exact), range +-8188

// MODE = Workspace::prec: 1 bf16x3, 2 bf16x6, 3 fp16x3, 4 fp16x4; 0 = the exact fp32 MFMA (v_mfma_f32_32x32x2_f32) on the same loop: the A
// operand is then the fp32 row-major matrix itself (128-byte LDS rows, chunk c of row r at c ^ ((r >> 1) & 7)), B is fed to the MFMAs unsplit.
// ASL: log2 of the power of two the B operand (activations) is multiplied by before an fp16 split (ignored by the bf16 modes)
template <int BM_, int WM_, int WN_, int NST_, int MODE_, int ASL_ = FP16_ACT_SCALE_LOG2>
struct SpTile {
    static constexpr int BM = BM_, WM = WM_, WN = WN_, NST = NST_, MODE = MODE_;
    static constexpr int TM = BM / WM / 32, TN = 1, BN = 32 * WN;
    static constexpr int WAVES = WM * WN, WAVES_MN = WAVES, THREADS = 64 * WAVES, KS = 1;
    static constexpr bool F16 = MODE >= 3, F32 = MODE == 0;
    static constexpr int PA = F32 ? 1 : MODE == 2 ? 3 : 2;        // planes per operand (16-bit terms; fp32: the matrix itself)
    static constexpr int ROW_BYTES = F32 ? 128 : 64;              // one LDS row = 32 k of one matrix row
    static constexpr int A_PLANE_BYTES = BM * ROW_BYTES;          // [BM][32], unpadded, swizzled
    static constexpr int A_BYTES = PA * A_PLANE_BYTES;
    static constexpr int B_BYTES = BK * BN * 4;                   // raw fp32 [32][BN]
    static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int RING_BYTES = NST * STAGE_BYTES;
    static constexpr int NA = A_BYTES / 1024, NB = B_BYTES / 1024;   // 1 KiB DMA pieces (one wave-instruction each) per slab
    static constexpr int G = (NA + NB) / WAVES;                      // pieces per wave per slab
    static constexpr int SMEM_FLOATS = RING_BYTES / 4;               // what an epilogue may re-use after the loop
    static constexpr float ACT_SCALE = F16 ? (float)(1 << ASL_) : 1.f;
    static_assert(TM >= 1 && BM % (32 * WM) == 0, "wave tile");
    static_assert(BM % 16 == 0 && (NA + NB) % WAVES == 0, "DMA pieces must divide evenly over the waves");
    static_assert(NST >= 2 && NST <= 4, "two to four stages");
    static_assert(BN == 64 || BN == 128, "a B piece is 2 or 4 whole k rows");
};

// Two-term fp16 split of (s a, s b), s an exact power of two, in five VALU instructions per pair: the first terms come out of
// v_fma_mixlo/hi_f16 (scale and round in one step: s x is exact, one RNE rounding to fp16), the remainders s x - x1 out of v_fma_mix_f32
// with the fp16 term as its third source (exact: the difference has at most 13 significant bits), the second terms out of one
// v_cvt_pk_f16_f32.  (hipcc's own lowering of the same arithmetic: mul, mul, cvt_pk, cvt, cvt, fma, fma, cvt_pk = eight.)
// Overflowing results saturate at +-65504 under fp16_saturate_mode() like the conversions do.
__device__ __forceinline__ void fp16_split2_scaled(float a, float b, float s, unsigned& hi, unsigned& lo) {
    float ra, rb;
    unsigned h;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(a), "s"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(b), "s"(s));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(ra) : "v"(a), "s"(s), "v"(h));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(rb) : "v"(b), "s"(s), "v"(h));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo) : "v"(ra), "v"(rb));
    hi = h;
}
// the same without a scale (four instructions per pair)
__device__ __forceinline__ void fp16_split2_mix(float a, float b, unsigned& hi, unsigned& lo) {
    float ra, rb;
    unsigned h;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(a), "v"(b));
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(ra) : "v"(a), "v"(h));
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(rb) : "v"(b), "v"(h));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo) : "v"(ra), "v"(rb));
    hi = h;
}

// 8 fp32 values (consecutive k of one column) -> the PA 16-bit planes of an MFMA operand register.  scale (fp16 modes): an exact
// power of two applied on the way (1: none).
template <int MODE>
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8 (&out)[MODE == 2 ? 3 : 2], float scale = 1.f) {
    unsigned p[3][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if constexpr (MODE >= 3) {
            if (scale != 1.f) fp16_split2_scaled(v[2 * q], v[2 * q + 1], scale, p[0][q], p[1][q]);
            else fp16_split2_mix(v[2 * q], v[2 * q + 1], p[0][q], p[1][q]);
        }
        else if constexpr (MODE == 1) bf16_split2(v[2 * q], v[2 * q + 1], p[0][q], p[1][q]);
        else bf16_split3(v[2 * q], v[2 * q + 1], p[0][q], p[1][q], p[2][q]);
    }
#pragma unroll
    for (int pl = 0; pl < (MODE == 2 ? 3 : 2); ++pl) out[pl] = __builtin_bit_cast(bf16x8, ((u32x4){p[pl][0], p[pl][1], p[pl][2], p[pl][3]}));
}

// the term products of one 32x32x16 block, small terms first (gemm_f32_mfma.h); FRESH: the first one starts from zero.
// SWAP: the two fragments trade places in the instruction (the MFMA's A and B operand registers have the same shape: 8 consecutive
// k of row / column lane & 31) -- the block comes out TRANSPOSED in the accumulators, lane = row of the a[] fragment, same bits.
template <int MODE, bool FRESH, bool SWAP = false>
__device__ __forceinline__ void mfma_terms(f32x16& c, const bf16x8 (&a)[MODE == 2 ? 3 : 2], const bf16x8 (&b)[MODE == 2 ? 3 : 2]) {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    auto f16 = [](bf16x8 x, bf16x8 y, f32x16 acc) {
        const f16x8 hx = __builtin_bit_cast(f16x8, x), hy = __builtin_bit_cast(f16x8, y);
        return SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_f16(hy, hx, acc, 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_f16(hx, hy, acc, 0, 0, 0);
    };
    auto b16 = [](bf16x8 x, bf16x8 y, f32x16 acc) {
        return SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, acc, 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc, 0, 0, 0);
    };
    if constexpr (MODE >= 3) {
        if constexpr (MODE == 4) {
            c = f16(a[1], b[1], FRESH ? z : c);
            c = f16(a[1], b[0], c);
        } else {
            c = f16(a[1], b[0], FRESH ? z : c);
        }
        c = f16(a[0], b[1], c);
        c = f16(a[0], b[0], c);
    } else if constexpr (MODE == 1) {
        c = b16(a[1], b[0], FRESH ? z : c);
        c = b16(a[0], b[1], c);
        c = b16(a[0], b[0], c);
    } else {
        c = b16(a[2], b[0], FRESH ? z : c);
        c = b16(a[0], b[2], c);
        c = b16(a[1], b[1], c);
        c = b16(a[1], b[0], c);
        c = b16(a[0], b[1], c);
        c = b16(a[0], b[0], c);
    }
}

// ---- hooks / operand transform interfaces ---------------------------------------------------------------------------------
// Hooks: target<I>(acc) -> the accumulator array slab I multiplies into; fresh<I, P>() -> its first product starts from zero;
//        bvals<I, P>(v) sees the RAW fp32 B values of (slab I, k16 half P): lane (column l31, half) holds k = 16 P + 8 half + j;
//        in_step<I>(acc) runs beside the first half of slab I's products (VALU work that should overlap them).
struct SpNoHooks {
    static constexpr bool ENABLED = false;
    template <int I, int TM>
    __device__ __forceinline__ f32x16 (&target(f32x16 (&acc)[TM]))[TM] { return acc; }
    template <int I, int P>
    static constexpr bool fresh() { return false; }
    template <int I, int P>
    __device__ __forceinline__ void bvals(const float (&)[8]) {}
    template <int I, int TM>
    __device__ __forceinline__ void in_step(f32x16 (&)[TM]) {}
};
// BX: per-k-row transform of the B values before the split (mlp.3: InstanceNorm + ReLU); fetch(k, x) reads the 8 per-row aux pairs
// of rows k .. k + 7 (the lane's own k run), apply(v, x) returns the value to split (the activation pre-scale is applied by the split).
struct SpNoBx {
    static constexpr bool ON = false;
    __device__ __forceinline__ void fetch(int, float2 (&)[8]) const {}
    __device__ __forceinline__ float apply(float v, float2) const { return v; }
};

// Wave-level time stamps (profiling builds; tools/trace_sp.py): s_memtime at the group boundaries of ONE step of the K loop and at
// the ends of prologue and loop.  Kept in SGPRs until the kernel writes them out: a handful of scalar instructions in total.
struct SpTrace {
    unsigned long long t[14];   // 0 after the prologue; 1..7 the boundaries of step STEP; 8 after the loop; 9..13 epilogue stages of the caller
};
constexpr int SP_TRACE_STEP = 5;

// a_pl(kt, plane): 16-bit plane pointer of A slab kt at the tile's first row, slab-major (row stride 32 elements = 64 bytes);
// b_slab(kt): fp32 pointer &B[kt * 32][col0], row stride ldb.  smem: T::RING_BYTES, 16-byte aligned.  All NST stages are free
// again when the function returns (it ends on a barrier behind the last fragment read).
// ABL (tuning builds only, WRONG results, timing only): bit 0 no DMA requests after the prologue, bit 1 no MFMAs, bit 2 no split VALU,
// bit 3 no fragment reads
// SCHED 0: every wave runs the schedule below (barrier between the two k16 halves, fragments read half a step ahead).
// SCHED 1 ("ping-pong"): the waves of a workgroup form two groups (first / second half of the wave ids: with eight waves the pairs
//   that share a SIMD); every wave runs  m(0) c(0) m(1) c(1) ...  with m(I) = DMA requests of slab I + NST - 1 and ALL fragment reads
//   of slab I, c(I) = splits + products of slab I, and ONE barrier per slab -- the early group behind c(I), the late group behind
//   m(I).  Between two barriers each wave does one m and one c, the groups in opposite order: while one wave of a SIMD multiplies,
//   its partner requests and reads, and only half of the workgroup loads the CU's DMA / LDS paths at a time.
// pre(): called once, right BEFORE the DMA requests of the first NST slabs -- the place for LDS-DMA fills of small per-workgroup tables
// (glds16 pieces: being older than every slab piece they are covered by the first counted wait and published by the first barrier; a
// table filled through registers would stall the wave on its load in front of the first slab requests).
struct SpNoPre {
    __device__ __forceinline__ void operator()() const {}
};
// OPT bit 0 (SP_OPT_SWAP): the MFMA operands trade places -- the accumulators hold the TRANSPOSED tile: lane = row (wm TM + tm) 32 + l31 of
//   the A operand, register r = column wn 32 + mfma_row(r, half) of the B operand (mlp.0 writes U^T point-major straight from them and
//   sums the InstanceNorm statistics inside a lane).  The products and their order are unchanged: the same bits, transposed.
// OPT bit 1 (SP_OPT_BT): the B operand is given TRANSPOSED in memory, B^T [column][k] fp32 with row stride ldb (floats): b_slab(kt) =
//   &BT[col0][32 kt].  Its slab arrives as 128-byte LDS rows (one per column, chunk c of row r at c ^ ((r >> 1) & 7): the fp32 A rows of
//   MODE 0) and a lane reads its 8 consecutive k with TWO 16-byte reads instead of eight 4-byte ones.
constexpr int SP_OPT_SWAP = 1, SP_OPT_BT = 2;
template <class T, int KT, class APlane, class BSlab, class Hooks, class BX, int ABL = 0, int SCHED = 0, class Pre = SpNoPre, int OPT = 0>
__device__ __forceinline__ void gemm_mainloop_sp(f32x16 (&acc)[T::TM], char* smem, APlane a_pl, BSlab b_slab, int ldb, Hooks& hooks,
                                                 BX& bx, SpTrace* tr_ = nullptr, Pre pre = Pre(), bool tron = false, int lda_bytes = 64) {
    constexpr bool SWAP = (OPT & SP_OPT_SWAP) != 0, BT = (OPT & SP_OPT_BT) != 0;
    // lda_bytes: row stride of an A slab in bytes (slab-major 16-bit planes: 64; a row-major fp32 matrix: 4 x its leading dimension)
    // (profiling builds: the stamps go into the caller's SpTrace through a reference and a separate on/off flag -- a conditional pointer
    //  keeps the object in scratch memory and costs the traced kernel 50 spilled registers)
    SpTrace tr_dummy;
    SpTrace& trr = tr_ ? *tr_ : tr_dummy;
    (void)trr; (void)tron;
    constexpr int TM = T::TM, PA = T::PA, BN = T::BN, NST = T::NST, G = T::G, MODE = T::MODE;
    static_assert(KT >= NST, "at least NST slabs");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / T::WN, wn = wave % T::WN;
    const int half = lane >> 5, l31 = lane & 31;

    // ---- LDS-DMA pieces: piece q of a slab = 1 KiB of its LDS image; lane i supplies bytes [16 i, 16 i + 16) of it.
    // A piece: 16 rows x 64 B of one plane; LDS chunk position i & 3 of row i >> 2 holds source chunk (i & 3) ^ ((row >> 2) & 3),
    // and (row >> 2) & 3 = (i >> 4) & 3 because pieces start on multiples of 16 rows.
    // (fp32 rows are 128 bytes: a piece is 8 rows, the swizzle (row >> 1) & 7 = (4 (piece & 1)) | (lane >> 4): the lane part below, the
    //  piece-parity part -- bit 2 of the chunk index = byte 64 -- is XORed in per piece)
    constexpr int CPR = T::ROW_BYTES / 16;   // 16-byte chunks per row (4 or 8)
    constexpr int RPQ = 64 / CPR;            // rows per piece (16 or 8)
    const unsigned a_lane_row = (unsigned)(lane / CPR), a_lane_chunk = (unsigned)((lane % CPR) ^ ((lane >> 4) & 3));
    const unsigned a_lane = a_lane_row * (unsigned)lda_bytes + a_lane_chunk * 16;
    constexpr int LPR = BN / 4;        // lanes per k row of a B piece (16 B each)
    constexpr int KPP = 64 / LPR;      // k rows per B piece
    // (BT: a piece is 8 columns x 128 B of B^T; lane i supplies LDS chunk position i & 7 of row i >> 3, which holds source chunk
    //  (i & 7) ^ ((row >> 1) & 7), (row >> 1) & 7 = 4 (piece & 1) | (i >> 4) inside an 8-row piece: the piece-parity bit is XORed in per piece)
    const unsigned b_lane = BT ? (unsigned)((lane >> 3) * ldb * 4 + (((lane & 7) ^ ((lane >> 4) & 3)) * 16))
                               : (unsigned)(((lane / LPR) * ldb + (lane % LPR) * 4) * 4);
    constexpr int RPP = T::BM / RPQ;   // A pieces per plane
    // pieces [j0, j1) of this wave's G pieces of slab kt
    auto issue_pieces = [&](int kt, int stage, int j0, int j1) {
        char* st = smem + stage * T::STAGE_BYTES;
#pragma unroll
        for (int j = 0; j < G; ++j) {
            if (j < j0 || j >= j1) continue;
            const int q = j * T::WAVES + wave;   // wave-uniform
            const bool is_a = (T::NA % T::WAVES == 0) ? (j < T::NA / T::WAVES) : (q < T::NA);
            if (is_a) {
                const int plane = (RPP % T::WAVES == 0) ? (j * T::WAVES) / RPP : q / RPP;
                const int qi = q % RPP;
                const unsigned flip = T::F32 ? (unsigned)(qi & 1) * 64u : 0u;
                glds16(reinterpret_cast<const char*>(a_pl(kt, plane)) + (size_t)(qi * RPQ) * lda_bytes + (a_lane ^ flip),
                       st + plane * T::A_PLANE_BYTES + qi * 1024);
            } else {
                const int qb = q - T::NA;
                if constexpr (BT)
                    glds16(reinterpret_cast<const char*>(b_slab(kt)) + (size_t)(qb * 8) * ldb * 4 + (b_lane ^ ((unsigned)(qb & 1) * 64u)),
                           st + T::A_BYTES + qb * 1024);
                else
                    glds16(reinterpret_cast<const char*>(b_slab(kt)) + (size_t)(qb * KPP) * ldb * 4 + b_lane, st + T::A_BYTES + qb * 1024);
            }
        }
    };
    auto issue = [&](int kt, int stage) { issue_pieces(kt, stage, 0, G); };

    // ---- fragment read offsets (bytes inside a stage)
    // 16-bit planes: a lane's 8 k of (k16 half P) are ONE chunk; fp32: two consecutive chunks (NCH = 2)
    constexpr int NCH = T::F32 ? 2 : 1;
    int a_off[TM][2][NCH];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int row = (wm * TM + tm) * 32 + l31;
        const int x = T::F32 ? (row >> 1) & 7 : (row >> 2) & 3;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int e = 0; e < NCH; ++e) a_off[tm][s][e] = row * T::ROW_BYTES + ((((2 * s + half) * NCH + e) ^ x) * 16);
    }
    const int b_off = T::A_BYTES + ((8 * half) * BN + wn * 32 + l31) * 4;
    int bt_off[2][2];   // BT: [k16 half][16-byte chunk] of the lane's column row
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int row = wn * 32 + l31;
            bt_off[s][e] = T::A_BYTES + row * 128 + ((((2 * s + half) * 2 + e) ^ ((row >> 1) & 7)) * 16);
        }

    bf16x8 Af[2][TM][PA * NCH];   // [k16 half][tm][plane] (fp32: the lane's 8 k as two 16-byte chunks)
    float Br[2][8];           // raw B values of the k16 half
    float2 Bx[2][8];          // their per-row aux pairs (BX::ON)
    bf16x8 Bf[2][PA];         // split B planes
    float Bv[2][8];           // fp32 mode: the (transformed) B values themselves

    auto read_b = [&](int stage, int slab, auto Pc) {
        constexpr int P = decltype(Pc)::value;
        const char* st = smem + stage * T::STAGE_BYTES;
        if constexpr (ABL & 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(Br[P][j]));
            return;
        }
        if constexpr (BT) {
            const vf4 lo = *reinterpret_cast<const vf4*>(st + bt_off[P][0]), hi = *reinterpret_cast<const vf4*>(st + bt_off[P][1]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                Br[P][j] = lo[j];
                Br[P][4 + j] = hi[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) Br[P][j] = *reinterpret_cast<const float*>(st + b_off + (P * 16 + j) * BN * 4);
        }
        if constexpr (BX::ON) bx.fetch(slab * BK + P * 16 + 8 * half, Bx[P]);
    };
    auto read_a = [&](int stage, auto Pc) {
        constexpr int P = decltype(Pc)::value;
        const char* st = smem + stage * T::STAGE_BYTES;
        if constexpr (ABL & 8) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int pl = 0; pl < PA * NCH; ++pl) asm volatile("" : "+v"(Af[P][tm][pl]));
            return;
        }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int pl = 0; pl < PA; ++pl)
#pragma unroll
                for (int e = 0; e < NCH; ++e)
                    Af[P][tm][pl * NCH + e] = *reinterpret_cast<const bf16x8*>(st + pl * T::A_PLANE_BYTES + a_off[tm][P][e]);
    };
    auto split_part = [&](auto Ic, auto Pc) {
        constexpr int I = decltype(Ic)::value, P = decltype(Pc)::value;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = Br[P][j];
        if constexpr (Hooks::ENABLED) hooks.template bvals<I, P>(v);
        if constexpr ((ABL & 4) && !T::F32) {
#pragma unroll
            for (int pl = 0; pl < PA; ++pl)
                Bf[P][pl] = __builtin_bit_cast(bf16x8, ((vf4){v[pl], v[pl + 2], v[pl + 4], v[7 - pl]}));
            return;
        }
        if constexpr (BX::ON) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = bx.apply(v[j], Bx[P][j]);
        }
        if constexpr (T::F32) {
#pragma unroll
            for (int j = 0; j < 8; ++j) Bv[P][j] = v[j];
        } else {
            split8<MODE>(v, Bf[P], T::ACT_SCALE);   // fp16 modes: scaled inside the split; bf16 modes: ACT_SCALE = 1
        }
    };
    auto mfma_part = [&](auto Ic, auto Pc, int tm0, int tm1) {
        constexpr int I = decltype(Ic)::value, P = decltype(Pc)::value;
        f32x16(&dst)[TM] = hooks.template target<I, TM>(acc);
        if constexpr (ABL & 2) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
                if (tm >= tm0 && tm < tm1) {
#pragma unroll
                    for (int pl = 0; pl < PA; ++pl) asm volatile("" ::"v"(Af[P][tm][pl]), "v"(Bf[P][pl]));
                    if constexpr (Hooks::template fresh<I, P>()) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) dst[tm][r] = 0.f;
                    }
                }
            return;
        }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
            if (tm >= tm0 && tm < tm1) {
                if constexpr (T::F32) {
                    // eight exact fp32 products: lane half h multiplies k = 16 P + 8 h + j (the same k assignment for A and B)
                    const vf4 a0 = __builtin_bit_cast(vf4, Af[P][tm][0]), a1 = __builtin_bit_cast(vf4, Af[P][tm][1]);
                    f32x16 c = dst[tm];
                    if constexpr (Hooks::template fresh<I, P>()) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) c[r] = 0.f;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        c = SWAP ? __builtin_amdgcn_mfma_f32_32x32x2f32(Bv[P][j], a0[j], c, 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], Bv[P][j], c, 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        c = SWAP ? __builtin_amdgcn_mfma_f32_32x32x2f32(Bv[P][4 + j], a1[j], c, 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], Bv[P][4 + j], c, 0, 0, 0);
                    dst[tm] = c;
                } else {
                    mfma_terms<MODE, Hooks::template fresh<I, P>(), SWAP>(dst[tm], Af[P][tm], Bf[P]);
                }
            }
    };

    // Schedule of step I (P0 / P1 = the two k16 halves of a slab; every fragment set is read half a step before its products):
    //   [ products (I, P0) || VALU: split of B (I, P1), hooks ]
    //   counted wait + barrier(I)          -- slab I + 1 landed for every wave, every wave has read slab I
    //   DMA requests of slab I + NST;  reads: raw B (I + 1, P0), raw B (I + 1, P1), A fragments (I + 1, P0)
    //   [ products (I, P1), first half of the wave's rows ]            -- covers the latency of those reads
    //   [ products (I, P1), second half || VALU: split of B (I + 1, P0) ]
    //   reads: A fragments (I + 1, P1)                                 -- land under the products (I + 1, P0)
    // sched_barrier(0) pins the bracketed groups; inside a group hipcc interleaves VALU, LDS reads and MFMAs by itself.
    auto stamp = [&](int I, int k) {   // (compiled out unless the caller passes a trace object: profiling builds)
#ifdef GATSSPG_PROFILING_BUILD
        if (tron && (I == SP_TRACE_STEP || k == 0 || k == 8)) trr.t[k] = __builtin_readcyclecounter();
#else
        (void)I; (void)k;
#endif
    };
    pre();
    static_for<0, NST>([&](auto Ic) { issue(decltype(Ic)::value, decltype(Ic)::value); });
    wait_dma_barrier<(NST - 1) * G>();
    if constexpr (SCHED == 1) {
        const int late = wave >= T::WAVES / 2 ? 1 : 0;   // wave-uniform (wave is an SGPR value)
        stamp(-1, 0);
        static_for<0, KT>([&](auto Ic) {
            constexpr int I = decltype(Ic)::value;
            // barrier #I lets slab I + 1 be read: this wave's pieces of the slabs requested after it may stay in flight
            constexpr int LATER = I + 1 < KT ? (I + NST - 1 < KT - 1 ? I + NST - 1 : KT - 1) - (I + 1) : 0;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (I == SP_TRACE_STEP) stamp(I, 1);
            // ---- m(I)
            if constexpr (I >= 1 && I + NST - 1 < KT && !(ABL & 1)) issue(I + NST - 1, (I - 1) % NST);   // the stage slab I - 1 left
            read_b(I % NST, I, IC<0>{});
            read_a(I % NST, IC<0>{});
            read_b(I % NST, I, IC<1>{});
            read_a(I % NST, IC<1>{});
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (I == SP_TRACE_STEP) stamp(I, 2);
            wait_dma_barrier_if<LATER * G>(late);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (I == SP_TRACE_STEP) stamp(I, 3);
            // ---- c(I)
            split_part(Ic, IC<0>{});
            mfma_part(Ic, IC<0>{}, 0, TM);
            split_part(Ic, IC<1>{});
            if constexpr (Hooks::ENABLED) hooks.template in_step<I, TM>(acc);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (I == SP_TRACE_STEP) stamp(I, 4);
            mfma_part(Ic, IC<1>{}, 0, TM);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (I == SP_TRACE_STEP) stamp(I, 5);
            wait_dma_barrier_if<LATER * G>(late ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (I == SP_TRACE_STEP) { stamp(I, 6); stamp(I, 7); }
        });
        __builtin_amdgcn_sched_barrier(0);
        stamp(-1, 8);
        wait_dma_barrier<0>();   // every wave is done with the ring: the epilogue may re-use it
        return;
    }
    if constexpr (SCHED == 2) {
        // SCHED 0 with the DMA requests of a step SPREAD over it instead of issued in one burst behind the barrier (where all waves of
        // the workgroup hit the CU's DMA path at once: ~140 cycles per piece in the trace): slot 0 behind the barrier, slot 1 between
        // the two halves of products (I, P1), slot 2 behind them, slot 3 (three-stage rings only: the slab has a whole further step to
        // land) inside the next step's first product group.  The raw B values of (I + 1, P1) are read between the halves as well.
        constexpr int NSLOT = NST >= 3 ? 4 : 3;
        constexpr int E0 = (G + NSLOT - 1) / NSLOT, E1 = E0 + (G - E0 + NSLOT - 2) / (NSLOT - 1);
        constexpr int E2 = NSLOT == 3 ? G : E1 + (G - E1 + 1) / 2;
        read_b(0, 0, IC<0>{});
        read_b(0, 0, IC<1>{});
        read_a(0, IC<0>{});
        read_a(0, IC<1>{});
        split_part(IC<0>{}, IC<0>{});
        stamp(-1, 0);
        static_for<0, KT>([&](auto Ic) {
            constexpr int I = decltype(Ic)::value;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (I == SP_TRACE_STEP) stamp(I, 1);
            mfma_part(Ic, IC<0>{}, 0, TM / 2);
            if constexpr (E2 < G && I >= 1 && I - 1 + NST < KT && !(ABL & 1)) {   // slot 3 of the slab requested in step I - 1
                __builtin_amdgcn_sched_barrier(0);
                issue_pieces(I - 1 + NST, (I - 1) % NST, E2, G);
                __builtin_amdgcn_sched_barrier(0);
            }
            mfma_part(Ic, IC<0>{}, TM / 2, TM);
            split_part(Ic, IC<1>{});
            if constexpr (Hooks::ENABLED) hooks.template in_step<I, TM>(acc);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (I == SP_TRACE_STEP) stamp(I, 2);
            if constexpr (I + 1 < KT) {
                constexpr int LATER = (I + NST - 1 < KT - 1 ? I + NST - 1 : KT - 1) - (I + 1);
                wait_dma_barrier<LATER * G>();
                if constexpr (I == SP_TRACE_STEP) stamp(I, 3);
                if constexpr (I + NST < KT && !(ABL & 1)) issue_pieces(I + NST, I % NST, 0, E0);
                read_b((I + 1) % NST, I + 1, IC<0>{});
                read_a((I + 1) % NST, IC<0>{});
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (I == SP_TRACE_STEP) stamp(I, 4);
            mfma_part(Ic, IC<1>{}, 0, TM / 2);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (I + 1 < KT) {
                if constexpr (I + NST < KT && !(ABL & 1)) issue_pieces(I + NST, I % NST, E0, E1);
                read_b((I + 1) % NST, I + 1, IC<1>{});
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (I == SP_TRACE_STEP) stamp(I, 5);
            mfma_part(Ic, IC<1>{}, TM / 2, TM);
            if constexpr (I + 1 < KT) split_part(IC<I + 1>{}, IC<0>{});
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (I == SP_TRACE_STEP) stamp(I, 6);
            if constexpr (I + 1 < KT) {
                if constexpr (I + NST < KT && !(ABL & 1)) issue_pieces(I + NST, I % NST, E1, E2);
                read_a((I + 1) % NST, IC<1>{});
            }
            if constexpr (I == SP_TRACE_STEP) {
                __builtin_amdgcn_sched_barrier(0);
                stamp(I, 7);
            }
        });
        __builtin_amdgcn_sched_barrier(0);
        stamp(-1, 8);
        wait_dma_barrier<0>();
        return;
    }
    if constexpr (SCHED == 3 || SCHED == 4) {
        // SCHED 2's data flow with the VALU work of a step pinned UNDER its MFMAs, one slot per MFMA (tools/microbench/mfma_valu_overlap.hip:
        // a wave that follows every 32-cycle MFMA with up to ~5 VALU instructions runs at the matrix pipe's rate, alone or with a partner
        // wave on its SIMD; 16 MFMAs followed by 80 VALU cost the sum -- and the per-step barrier keeps the two waves of a SIMD in phase, so
        // a partner cannot fill the gap).  hipcc left to itself keeps the four term products of an accumulator back to back and packs a
        // half step's 20-28 split instructions under four of its eight MFMAs; here every slot is fenced:
        //   products (I, P0) m = 0 .. HM-1 : pair q of split (I, P1) under m = q HM/4 (hooks: the raw-value dot under m = 0, the head fold under m = HM/2)
        //   counted wait + barrier(I)
        //   products (I, P1) m = 0 .. HM-1 : DMA pieces + reads of slab I + 1 under m = 0, HM/4, HM-2; pair q of split (I + 1, P0) under m = HM-4+q
        static_assert(!T::F32 && MODE >= 3, "the slot schedule is written for the two-plane fp16 modes");
        constexpr int NT = MODE == 4 ? 4 : 3, HM = TM * NT, QA = HM / 4, S0 = HM - 4, PP = 1;
        constexpr bool SPREAD = SCHED == 4;   // the reads of slab I + 1 one group per slot instead of two groups under product 1 of (I, P1)
        static_assert(QA >= 1 && S0 >= 0, "four pairs per half step");
        constexpr int NSLOT = NST >= 3 ? 4 : 3;
        constexpr int E0 = (G + NSLOT - 1) / NSLOT, E1 = E0 + (G - E0 + NSLOT - 2) / (NSLOT - 1);
        constexpr int E2 = NSLOT == 3 ? G : E1 + (G - E1 + 1) / 2;
        unsigned Bw[2][2][4];   // [k16 half][plane][dword]: the split B planes, written pair by pair
        auto one = [&](auto Ic, auto Pc, auto Mc) {
            constexpr int I = decltype(Ic)::value, P = decltype(Pc)::value, M = decltype(Mc)::value, tm = M / NT, t = M % NT;
            f32x16(&dst)[TM] = hooks.template target<I, TM>(acc);
            // term order of mfma_terms (small terms first): fp16x4 (a1 b1) (a1 b0) (a0 b1) (a0 b0); fp16x3 (a1 b0) (a0 b1) (a0 b0)
            constexpr int ap = MODE == 4 ? (t < 2 ? 1 : 0) : (t == 0 ? 1 : 0);
            constexpr int bp = MODE == 4 ? ((t & 1) ? 0 : 1) : (t == 1 ? 1 : 0);
            const f16x8 a = __builtin_bit_cast(f16x8, Af[P][tm][ap]);
            const f16x8 b = __builtin_bit_cast(f16x8, ((u32x4){Bw[P][bp][0], Bw[P][bp][1], Bw[P][bp][2], Bw[P][bp][3]}));
            if constexpr (t == 0 && Hooks::template fresh<I, P>()) {
                f32x16 z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.f;
                dst[tm] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, z, 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, z, 0, 0, 0);
            } else {
                dst[tm] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, dst[tm], 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, dst[tm], 0, 0, 0);
            }
        };
        auto pair = [&](auto Pc, auto Qc) {
            constexpr int P = decltype(Pc)::value, q = decltype(Qc)::value;
            float a = Br[P][2 * q], b = Br[P][2 * q + 1];
            if constexpr (BX::ON) {
                a = bx.apply(a, Bx[P][2 * q]);
                b = bx.apply(b, Bx[P][2 * q + 1]);
            }
            if (T::ACT_SCALE != 1.f) fp16_split2_scaled(a, b, T::ACT_SCALE, Bw[P][0][q], Bw[P][1][q]);
            else fp16_split2_mix(a, b, Bw[P][0][q], Bw[P][1][q]);
        };
        read_b(0, 0, IC<0>{});
        read_b(0, 0, IC<1>{});
        read_a(0, IC<0>{});
        read_a(0, IC<1>{});
        if constexpr (Hooks::ENABLED) hooks.template bvals<0, 0>(Br[0]);
        static_for<0, 4>([&](auto Qc) { pair(IC<0>{}, Qc); });
        stamp(-1, 0);
        static_for<0, KT>([&](auto Ic) {
            constexpr int I = decltype(Ic)::value;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (I == SP_TRACE_STEP) stamp(I, 1);   // (profiling builds) 1 step start, 2 / 3 behind the first / second half of products (I, P0)
            static_for<0, HM>([&](auto Mc) {
                constexpr int M = decltype(Mc)::value;
                if constexpr (I == SP_TRACE_STEP && M == HM / 2) stamp(I, 2);
                one(Ic, IC<0>{}, Mc);
                if constexpr (M == 0 && Hooks::ENABLED) hooks.template bvals<I, 1>(Br[1]);
                if constexpr (PP == 1) {
                    if constexpr (M % QA == 0 && M / QA < 4) pair(IC<1>{}, IC<M / QA>{});
                } else {   // two pairs per slot: their instructions fill each other's hazard slots (three s_nop per lone pair)
                    if constexpr (M % (2 * QA) == 0 && M / (2 * QA) < 2) {
                        pair(IC<1>{}, IC<2 * (M / (2 * QA))>{});
                        pair(IC<1>{}, IC<2 * (M / (2 * QA)) + 1>{});
                    }
                }
                if constexpr (M == 1 && E2 < G && I >= 1 && I - 1 + NST < KT && !(ABL & 1)) issue_pieces(I - 1 + NST, (I - 1) % NST, E2, G);
                if constexpr (M == HM / 2 && Hooks::ENABLED) hooks.template in_step<I, TM>(acc);
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (I == SP_TRACE_STEP) stamp(I, 3);
            if constexpr (I + 1 < KT) {
                constexpr int LATER = (I + NST - 1 < KT - 1 ? I + NST - 1 : KT - 1) - (I + 1);
                wait_dma_barrier<LATER * G>();
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (I == SP_TRACE_STEP) stamp(I, 4);   // 4 behind the barrier, 5 / 6 behind products 2 / 4 of (I, P1), 7 behind the last
            static_for<0, HM>([&](auto Mc) {
                constexpr int M = decltype(Mc)::value;
                if constexpr (I == SP_TRACE_STEP && M == 2) stamp(I, 5);
                if constexpr (I == SP_TRACE_STEP && M == 4) stamp(I, 6);
                if constexpr (I + 1 < KT) {
                    if constexpr (M == 0) {
                        read_b((I + 1) % NST, I + 1, IC<0>{});
                        if constexpr (!SPREAD) read_a((I + 1) % NST, IC<0>{});
                        if constexpr (I + NST < KT && !(ABL & 1)) issue_pieces(I + NST, I % NST, 0, E0);
                    }
                    if constexpr (SPREAD && M == 1) read_a((I + 1) % NST, IC<0>{});
                    if constexpr (M == (SPREAD ? 2 : QA)) {
                        if constexpr (I + NST < KT && !(ABL & 1)) issue_pieces(I + NST, I % NST, E0, E1);
                        if constexpr (!SPREAD) read_b((I + 1) % NST, I + 1, IC<1>{});
                    }
                    if constexpr (SPREAD && M == 3) read_b((I + 1) % NST, I + 1, IC<1>{});
                }
                one(Ic, IC<1>{}, Mc);
                if constexpr (I + 1 < KT) {
                    if constexpr (M == S0 && Hooks::ENABLED) hooks.template bvals<I + 1, 0>(Br[0]);
                    if constexpr (PP == 1) {
                        if constexpr (M >= S0) pair(IC<0>{}, IC<M - S0>{});
                    } else if constexpr (M == S0 || M == S0 + 2) {
                        pair(IC<0>{}, IC<M - S0>{});
                        pair(IC<0>{}, IC<M - S0 + 1>{});
                    }
                    if constexpr (M == HM - 2 && I + NST < KT && !(ABL & 1)) issue_pieces(I + NST, I % NST, E1, E2);
                    if constexpr (M == HM - 1) read_a((I + 1) % NST, IC<1>{});
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (I == SP_TRACE_STEP) stamp(I, 7);
        });
        __builtin_amdgcn_sched_barrier(0);
        stamp(-1, 8);
        wait_dma_barrier<0>();
        return;
    }
    read_b(0, 0, IC<0>{});
    read_b(0, 0, IC<1>{});
    read_a(0, IC<0>{});
    read_a(0, IC<1>{});
    split_part(IC<0>{}, IC<0>{});
    stamp(-1, 0);
    static_for<0, KT>([&](auto Ic) {
        constexpr int I = decltype(Ic)::value;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (I == SP_TRACE_STEP) stamp(I, 1);
        mfma_part(Ic, IC<0>{}, 0, TM);
        split_part(Ic, IC<1>{});
        if constexpr (Hooks::ENABLED) hooks.template in_step<I, TM>(acc);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (I == SP_TRACE_STEP) stamp(I, 2);
        if constexpr (I + 1 < KT) {
            // this wave's pieces still wanted in flight behind slab I + 1: those of the slabs requested after it
            constexpr int LATER = (I + NST - 1 < KT - 1 ? I + NST - 1 : KT - 1) - (I + 1);
            wait_dma_barrier<LATER * G>();
            if constexpr (I == SP_TRACE_STEP) stamp(I, 3);
            if constexpr (I + NST < KT && !(ABL & 1)) issue(I + NST, I % NST);
            read_b((I + 1) % NST, I + 1, IC<0>{});
            read_b((I + 1) % NST, I + 1, IC<1>{});
            read_a((I + 1) % NST, IC<0>{});
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (I == SP_TRACE_STEP) stamp(I, 4);
        mfma_part(Ic, IC<1>{}, 0, TM / 2);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (I == SP_TRACE_STEP) stamp(I, 5);
        mfma_part(Ic, IC<1>{}, TM / 2, TM);
        if constexpr (I + 1 < KT) split_part(IC<I + 1>{}, IC<0>{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (I == SP_TRACE_STEP) stamp(I, 6);
        if constexpr (I + 1 < KT) read_a((I + 1) % NST, IC<1>{});
        if constexpr (I == SP_TRACE_STEP) {
            __builtin_amdgcn_sched_barrier(0);
            stamp(I, 7);
        }
    });
    __builtin_amdgcn_sched_barrier(0);
    stamp(-1, 8);
    wait_dma_barrier<0>();   // every wave is done with the ring: the epilogue may re-use it
}

}  // namespace gatsspg
