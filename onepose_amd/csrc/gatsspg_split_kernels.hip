// The three big GEMMs of an attention layer on the split-16-bit main loop of gemm_split_glds.h (round 4): qkv_kv, mlp.0 (merge and
// the linear-attention apply folded in) and mlp.3, for the arithmetics GATSSPG_FLAG_PREC_FP16X4 / _FP16X3 (always) and _BF16X3 /
// _BF16X6.  Same maths, same buffers and the same epilogues as the fp32 kernels of gatsspg_gemm_kernels.hip
// (GATs_SuperGlue.py:69-128); what differs is how the operands reach the matrix pipe.
#include "gemm_split_glds.h"
#include "gatsspg_launch.h"

namespace gatsspg {

#ifndef GATSSPG_PROFILING_BUILD
static constexpr unsigned long long* g_trace = nullptr;   // (profiling builds: the buffer of gatsspg_debug_set_trace, tools/trace_sp.py)
#define SP_TRACE_ON(ptr) false
#else
#define SP_TRACE_ON(ptr) ((ptr) != nullptr)
#endif

template <auto Kernel>
static void allow_big_lds_sp() {
    static bool done[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !done[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
        if (dev >= 0 && dev < 64) done[dev] = true;
    }
}

// plain hooks: no side work; the first product of the K loop starts from zero (FRESH0) or from the caller's accumulators
template <bool FRESH0>
struct SpPlainHooks {
    static constexpr bool ENABLED = false;
    template <int I, int TM>
    __device__ __forceinline__ f32x16 (&target(f32x16 (&acc)[TM]))[TM] { return acc; }
    template <int I, int P>
    static constexpr bool fresh() { return FRESH0 && I == 0 && P == 0; }
    template <int I, int P>
    __device__ __forceinline__ void bvals(const float (&)[8]) {}
    template <int I, int TM>
    __device__ __forceinline__ void in_step(f32x16 (&)[TM]) {}
};

// this lane's 16 bias values per 32-row MFMA tile (rows 8 k + 4 half + 0..3: four 16-byte loads)
template <class T>
__device__ __forceinline__ void load_bias16(const float* b, int row0, int wm, int half, float (&bias)[T::TM][16]) {
#pragma unroll
    for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const vf4 b4 = ldg4(b + row0 + (wm * T::TM + tm) * 32 + 8 * k + 4 * half);
            bias[tm][4 * k + 0] = b4[0]; bias[tm][4 * k + 1] = b4[1]; bias[tm][4 * k + 2] = b4[2]; bias[tm][4 * k + 3] = b4[3];
        }
}

// =====================================================================================================
// K1  QKV projection + KV / ksum partials (qkv_kv_kernel of gatsspg_gemm_kernels.hip on the split loop).
//     128 x 64 tile on 4 waves (64 x 32 per wave), two stages (48 KiB): three workgroups per CU, so the 756 tiles of the headline
//     shape are resident at once and every SIMD holds three waves of three different workgroups (no common barrier).
// =====================================================================================================
template <int MODE>
using QkvSpTile = SpTile<128, 2, 2, 2, MODE>;
// the exact fp32 MFMA on the same loop (MODE 0): 128 x 64 on 8 waves of 32 x 32 like the fp32 kernels of gatsspg_gemm_kernels.hip (two
// workgroups per CU, <= 128 registers), three stages of 24 KiB
using Fp32SpTile = SpTile<128, 4, 2, 3, 0>;

// A plain output tile straight from the accumulators: in the 32 x 32 C layout a lane's 16 values of one product sit in ONE column
// (lane & 31) and 16 rows, so each dword store instruction covers two rows x 32 consecutive columns = two full 128-byte lines --
// no LDS round trip, no barrier in front of the stores (store_tile_via_lds: 32 scalar LDS writes, a barrier, 8 row reads, 8 16-byte stores).
template <class T, class F>
__device__ __forceinline__ void store_tile_direct(const f32x16 (&acc)[T::TM][T::TN], float* dst, int ld, F f) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / T::WN, wn = wave % T::WN, half = lane >> 5, l31 = lane & 31;
    float* d = dst + wn * 32 + l31;
#pragma unroll
    for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (wm * T::TM + tm) * 32 + mfma_row(r, half);
            d[(size_t)row * ld] = f(row, acc[tm][0][r]);
        }
}

template <class T, int SCHED = 0, int EPI = 0>
__global__ __launch_bounds__(T::THREADS, (T::F32 ? 4 : 3)) void qkv_kv_sp_kernel(const float* __restrict__ sc, const float* __restrict__ bqkv,
                                                                const unsigned short* __restrict__ P0, const unsigned short* __restrict__ P1,
                                                                const unsigned short* __restrict__ P2, const float* __restrict__ Z,
                                                                float* __restrict__ Qbuf, float* __restrict__ kvpart, ColLayout L) {
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    float* smem = reinterpret_cast<float*>(smem_c);
    if constexpr (T::F16) fp16_saturate_mode();
    int rt, ct;
    if (!xcd_tile_map_g(6, active_tiles(L), L.xgs, rt, ct)) return;
    ct = global_tile(L, ct);
    const int c0 = ct * T::BN, ld = L.ld;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / T::WN, wn = wave % T::WN, half = lane >> 5, l31 = lane & 31;
    constexpr bool BIAS_TAB = (EPI & 2) != 0;   // bias through an LDS table behind the ring (T::BM floats)
    static_assert(T::BM == 128, "one half piece of bias values");
    float* btab = reinterpret_cast<float*>(smem_c + T::RING_BYTES);
    float bias[T::TM][16];
    if constexpr (!BIAS_TAB) load_bias16<T>(bqkv, rt * 128, wm, half, bias);
    const float inv = T::F16 ? 1.f / (sc[0] * T::ACT_SCALE) : 1.f;
    f32x16 acc[T::TM][T::TN];
    const size_t ro = (size_t)rt * 128 * BK;
    // 16-bit modes: slab-major planes; fp32 (MODE 0): P0 is the row-major fp32 operator [768][256] itself
    auto apl = [&](int kt, int pl) -> const void* {
        if constexpr (T::F32) return reinterpret_cast<const float*>(P0) + (size_t)rt * 128 * D + kt * BK;
        else return (pl == 0 ? P0 : pl == 1 ? P1 : P2) + ro + (size_t)kt * 768 * BK;
    };
    auto bsl = [&](int kt) { return Z + (size_t)kt * BK * ld + c0; };
    SpPlainHooks<true> hooks;
    SpNoBx nobx;
    auto pre = [&]() {
        if constexpr (BIAS_TAB) {
            if (wave == 0 && lane < 32) glds16(bqkv + rt * 128 + 4 * lane, btab);   // 128 floats: half a piece
        }
    };
    gemm_mainloop_sp<T, D / BK, decltype(apl), decltype(bsl), SpPlainHooks<true>, SpNoBx, 0, SCHED, decltype(pre)>(
        reinterpret_cast<f32x16(&)[T::TM]>(acc), smem_c, apl, bsl, ld, hooks, nobx, nullptr, pre, false, T::F32 ? D * 4 : 64);
    if constexpr (BIAS_TAB) read_bias16<T>(btab, wm, half, bias);

    if (rt < 2) {
#pragma unroll
        for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][0][r] = elu1_select(fmaf(acc[tm][0][r], inv, bias[tm][r])) + 1.f;
        if constexpr (EPI & 1) store_tile_direct<T>(acc, Qbuf + (size_t)rt * 128 * ld + c0, ld, [](int, float v) { return v; });
        else store_tile_via_lds<T>(acc, smem, Qbuf + (size_t)rt * 128 * ld + c0, ld, [](int, float v) { return v; });
        return;
    }
    // ---- K_h / V_h tile -> LDS -> KV partial (second MFMA pass, fp32: exact like the fp32 kernel's)
    const int h = rt - 2;
    const TileSeg ts = tile_seg(L, c0, T::BN);
    constexpr int TS = T::BN + 4;
    static_assert(T::BN == QKV_BN && T::WAVES >= 4 && 128 * TS * 4 <= T::RING_BYTES, "one KV partial per 64-column tile");
    float* Tl = smem;
    float opmx = 0.f;   // largest K (> 0) or |V| entry this wave holds: its 64 rows are all K_h or all V_h
#pragma unroll
    for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (wm * T::TM + tm) * 32 + mfma_row(r, half);  // 0..63 = K_h channel d, 64..127 = V_h channel q
            const int col = wn * 32 + l31;
            float v = fmaf(acc[tm][0][r], inv, bias[tm][r]);
            const float kf = elu1_select(v) + 1.f;
            v = row < 64 ? kf : v;
            v = col >= ts.valid ? 0.f : v;  // pad columns must not enter the sums
            opmx = fmaxf(opmx, fabsf(v));
            Tl[row * TS + col] = v;
        }
    {
        // bound data for kv_final's scale of the message operator (|KV_h[q][d]| <= max |V| * ksum[d]): slots [4..7] max |V| per wave here,
        // slots [0..3] the per-wave largest key sum of the tile (ksum pass below).  Written in every arithmetic (a partial / database cache
        // must never carry uninitialised slots: round-5 advisor)
        static_assert((T::WAVES == 4 && T::TM == 2) || (T::WAVES == 8 && T::TM == 1), "4 waves: 0, 1 hold K_h, 2, 3 V_h; 8 waves: 0..3 K_h, 4..7 V_h");
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) opmx = fmaxf(opmx, __shfl_xor(opmx, o));
        if (lane == 0) {   // slots 4..7: max |V| per V-holding wave (0 for the K-holding ones); slots 0..3: the ksum pass below
            float* mx = kvpart + ((size_t)ct * H + h) * KVP + DH * DH + DH;
            if constexpr (T::WAVES == 8) {
                if (wave >= 4) mx[wave] = opmx;
            } else {
                mx[4 + wave] = wm == 0 ? 0.f : opmx;
            }
        }
    }
    __syncthreads();
    if (wave < 4) {   // (an 8-wave workgroup leaves this short pass to its first four waves)
        const int qi = wave >> 1, di = wave & 1;
        f32x16 kv;
#pragma unroll
        for (int r = 0; r < 16; ++r) kv[r] = 0.f;
        const float4* ap = reinterpret_cast<const float4*>(Tl + (di * 32 + l31) * TS + half * 32);
        const float4* bp = reinterpret_cast<const float4*>(Tl + (64 + qi * 32 + l31) * TS + half * 32);
#pragma unroll
        for (int v4 = 0; v4 < 8; ++v4) {
            const float4 a = ap[v4], b = bp[v4];
            kv = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, kv, 0, 0, 0);
            kv = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, kv, 0, 0, 0);
            kv = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, kv, 0, 0, 0);
            kv = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, kv, 0, 0, 0);
        }
        float* out = kvpart + ((size_t)ct * H + h) * KVP;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = di * 32 + mfma_row(r, half);
            out[d * DH + qi * 32 + l31] = kv[r];
        }
        {   // ksum[d] = sum_m K[d][m]: 4 lanes per row (16 columns each, fixed order), combined by 2 shuffles
            const int d = tid >> 2, qtr = tid & 3;
            const float* kr = Tl + d * TS + qtr * 16;
            float s = 0.f;
#pragma unroll
            for (int m = 0; m < 16; ++m) s += kr[m];
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            if (qtr == 0) out[DH * DH + d] = s;
            // slot `wave` of the bound data (kv_final_kernel): the largest of this wave's 16 key sums
            float m = s;
#pragma unroll
            for (int o = 4; o <= 32; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
            if (lane == 0) out[DH * DH + DH + wave] = m;
        }
    }
}

// =====================================================================================================
// K4  mlp.0 with merge and the linear-attention apply folded in (mlp0_kernel / AttnFoldHooks of the fp32 path):
//         u = W0a x + sum_h z_h (.) (M_h Qf_h) + b,   z_h[n] = 1 / (Qf_h[:, n] . ksum_h + 1e-6)
//     K loop over [x ; Qf]: slabs 0..7 accumulate the x part, slabs 8 + 2h, 9 + 2h head h into one of TWO alternating head
//     accumulators that start from zero (the first product takes C = 0); head h - 1 is folded into the kept sum with its per-column
//     z while head h multiplies, so the fold's VALU work never waits for the matrix pipe.  The denominators come from the RAW B
//     values a lane holds anyway (8 consecutive k of its column per k16 half): 8 FMAs against the source's ksum (an LDS table),
//     added over the head's four halves in a fixed order, the two lane halves combined by one exchange -- per-lane in exactly the
//     32x32 C layout the fold needs, no LDS round trip.
// =====================================================================================================
// UT (transposed accumulators, SP_OPT_SWAP of the main loop: lane = channel, register r = point mfma_row(r, half) of the wave's 32-point
// strip): the denominator of point p is still summed by lane l31 = p (the raw B values a lane holds are those of ITS point), so a fold
// takes the factors of its registers' points from one rank-1 fp32 MFMA (z x 1: the accumulator layout of the tile itself).
template <int TM, bool UT = false>
struct AttnFoldSp {
    static constexpr bool ENABLED = true;
    static constexpr int SPLIT = 8;
    f32x16 hacc[2][TM];
    float dpart[2];
    const float* ks;   // LDS: ksum of the source segment [4][64]
    float zfac[4];     // per head: (scale of the W0 planes) / (scale of the head's operator planes), an exact power of two (1 in the bf16 modes)
    int half;
    template <int I, int TM_>
    __device__ __forceinline__ f32x16 (&target(f32x16 (&acc)[TM_]))[TM_] {
        if constexpr (I < SPLIT) return acc;
        else return hacc[((I - SPLIT) >> 1) & 1];
    }
    template <int I, int P>
    static constexpr bool fresh() { return P == 0 && (I == 0 || (I >= SPLIT && ((I - SPLIT) & 1) == 0)); }
    template <int I, int P>
    __device__ __forceinline__ void bvals(const float (&v)[8]) {
        if constexpr (I >= SPLIT) {
            constexpr int h = (I - SPLIT) >> 1;
            const float* k = ks + h * 64 + ((I - SPLIT) & 1) * 32 + P * 16 + 8 * half;
            const float4 k0 = *reinterpret_cast<const float4*>(k), k1 = *reinterpret_cast<const float4*>(k + 4);
            float p = v[0] * k0.x;
            p = fmaf(v[1], k0.y, p); p = fmaf(v[2], k0.z, p); p = fmaf(v[3], k0.w, p);
            p = fmaf(v[4], k1.x, p); p = fmaf(v[5], k1.y, p); p = fmaf(v[6], k1.z, p); p = fmaf(v[7], k1.w, p);
            if constexpr (((I - SPLIT) & 1) == 0 && P == 0) dpart[h & 1] = p;
            else dpart[h & 1] += p;
        }
    }
    template <int HD>
    __device__ __forceinline__ void fold(f32x16 (&kept)[TM]) {
        float d = dpart[HD & 1];
        const float o = __shfl_xor(d, 32);
        d = half ? o + d : d + o;   // lane half 0's partial first on both halves
        const float z = zfac[HD] / (d + 1e-6f);
        if constexpr (!UT) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) kept[tm][r] = fmaf(z, hacc[HD & 1][tm][r], kept[tm][r]);
        } else {
            // register r of a lane holds point mfma_row(r, half) of the strip; the factor of point p sits in lane p.  ONE fp32 MFMA hands
            // every lane the 16 factors of its registers' points: D[i][j] = sum_k A[i][k] B[k][j] with A[i][0] = z_i, B[0][j] = 1 (k = 1
            // operands zero) = z_i for every column j, in the accumulator layout of the tile itself (exact: z * 1 + 0 * 0).  64 cycles of
            // matrix pipe per head and wave instead of 2 v_readlane + 2 v_mov + 1 v_cndmask per register (measured: +1.4 % per frame).
            f32x16 zm;
#pragma unroll
            for (int r = 0; r < 16; ++r) zm[r] = 0.f;
            zm = __builtin_amdgcn_mfma_f32_32x32x2f32(half ? 0.f : z, half ? 0.f : 1.f, zm, 0, 0, 0);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) kept[tm][r] = fmaf(zm[r], hacc[HD & 1][tm][r], kept[tm][r]);
        }
    }
    // head h - 1 is folded beside the first products of head h (its own products were issued a whole slab earlier)
    template <int I, int TM_>
    __device__ __forceinline__ void in_step(f32x16 (&acc)[TM_]) {
        if constexpr (I >= SPLIT + 2 && ((I - SPLIT) & 1) == 0) fold<((I - SPLIT) >> 1) - 1>(acc);
    }
};

template <int MODE>
using Mlp0SpTileW = SpTile<128, 2, 4, 3, MODE>;   // 128 x 128 on 8 waves: 252 workgroups at the headline shape, one per CU, 96 KiB ring
template <int MODE>
using Mlp0SpTileN = SpTile<128, 2, 2, 3, MODE>;   // 128 x 64 on 4 waves: twice the workgroups (small shapes), two per CU
template <int MODE>
using Mlp0SpTileN2 = SpTile<128, 2, 2, 2, MODE>;  // the same on a two-stage ring (49 KiB): three workgroups per CU when the registers allow (<= 168)
template <int MODE>
using Mlp0SpTileW4 = SpTile<128, 2, 4, 4, MODE>;  // the 8-wave tile on a FOUR-stage ring (128 KiB): a slab has two and a half steps to land
template <int MODE>
using Mlp0SpTileT = SpTile<128, 1, 4, 3, MODE>;   // 128 x 128 on 4 waves, 128 x 32 per wave (one wave per SIMD, every B value split once)

template <class T, int ABL = 0, int SCHED = 0, int EPI = 0>
__global__ __launch_bounds__(T::THREADS, (T::F32 ? 4 : T::TM == 4 ? 1 : (T::WAVES == 4 && T::NST == 2) ? 3 : 2)) void mlp0_sp_kernel(const float* __restrict__ sc, const float* __restrict__ b0,
                                                              const unsigned short* __restrict__ P0, const unsigned short* __restrict__ P1,
                                                              const unsigned short* __restrict__ P2, const float* __restrict__ Z,
                                                              const float* __restrict__ Qbuf, const unsigned short* __restrict__ Mpl,
                                                              const float* __restrict__ ksumT, const float* __restrict__ zsc,
                                                              float* __restrict__ U, float* __restrict__ statpart, float* __restrict__ stats,
                                                              int* __restrict__ statcnt, ColLayout L, unsigned long long* trace) {
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    float* smem = reinterpret_cast<float*>(smem_c);
    if constexpr (T::F16) fp16_saturate_mode();
    SpTrace tr;
    const unsigned long long t_entry = SP_TRACE_ON(trace) ? __builtin_readcyclecounter() : 0;
    const unsigned long long w_entry = SP_TRACE_ON(trace) ? wall_clock64() : 0;   // 100 MHz constant clock: calibrates the s_memtime ticks
    int rt, ct;
    constexpr int TPW = T::BN / MLP0_BN;   // 64-column tiles (= InstanceNorm partials) per workgroup
    constexpr int MT = 512 / T::BM;
    if (!xcd_tile_map(MT, active_tiles(L) / TPW, rt, ct)) return;
    ct = global_tile(L, ct * TPW) / TPW;   // windows and segments are multiples of 128 columns
    const int c0 = ct * T::BN, ld = L.ld;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / T::WN, wn = wave % T::WN, half = lane >> 5, l31 = lane & 31;
    const TileSeg ts = tile_seg(L, c0, T::BN);
    // ksum of the source segment -> LDS table behind the ring (published by the first barrier of the main loop)
    float* tab = reinterpret_cast<float*>(smem_c + T::RING_BYTES);
    // (filled by ONE LDS-DMA piece in front of the first slab requests: no register hop, no wait of its own)
    // requested before the main loop (behind it the two dependent round trips would sit on the critical path: measured 3.9 k cycles of
    // a 39 k-cycle kernel); hipcc parks some of the 32 values in scratch across the loop, which costs two scratch instructions each
    constexpr bool BIAS_TAB = (EPI & 2) != 0;   // bias through a second LDS table (T::BM floats behind the ksum table)
    static_assert(T::BM == 128, "one half piece of bias values");
    float* btab = tab + 256;
    float bias[T::TM][16];
    if constexpr (!BIAS_TAB) load_bias16<T>(b0, rt * T::BM, wm, half, bias);
    const float inv = T::F16 ? 1.f / (sc[1] * T::ACT_SCALE) : 1.f;
    f32x16 acc[T::TM][T::TN];
    const size_t ro = (size_t)rt * T::BM * BK;   // slab-major planes: (m, k) at ((k / 32) * 512 + m) * 32 + k % 32
    const unsigned short* Mh = Mpl + (size_t)ts.seg * 3 * MPL_PLANE + ro;
    // 16-bit modes: slab-major planes of W0 (x half) and of the segment's message operator; fp32 (MODE 0): P0 = the fp32 [512][512] operator,
    // Mpl = the fp32 operator block Mop (the segment's M_t at mop_seg(), row stride MOP_LD = 512 like W0)
    auto apl = [&](int kt, int pl) -> const void* {
        if constexpr (T::F32) {
            const float* W0f = reinterpret_cast<const float*>(P0) + (size_t)rt * T::BM * 512;
            const float* Mf = mop_seg(reinterpret_cast<const float*>(Mpl), ts.seg) + (size_t)rt * T::BM * MOP_LD;
            return kt < 8 ? W0f + kt * BK : Mf + (kt - 8) * BK;
        } else {
            return kt < 8 ? (pl == 0 ? P0 : pl == 1 ? P1 : P2) + ro + (size_t)kt * 512 * BK
                          : Mh + (size_t)pl * MPL_PLANE + (size_t)(kt - 8) * 512 * BK;
        }
    };
    auto bsl = [&](int kt) { return (kt < 8 ? Z + (size_t)kt * BK * ld : Qbuf + (size_t)(kt - 8) * BK * ld) + c0; };
    // EPI bit 3 (UT): transposed accumulators -- U leaves POINT-major (U^T [ld][512]: mlp3_sp reads 8 consecutive channels of a point
    // with two 16-byte LDS reads) straight from the registers, and the InstanceNorm partials are summed inside a lane (one per 32-point
    // strip of a wave): no staging tile, no barrier, no statistics walk behind the loop (round-4 trace: 9 k of a wave's 39.5 k cycles)
    constexpr bool UT = (EPI & 8) != 0;
    AttnFoldSp<T::TM, UT> hooks;
    hooks.ks = tab; hooks.half = half;
    {
        const float4 zf = *reinterpret_cast<const float4*>(zsc + ts.seg * H);
        hooks.zfac[0] = zf.x; hooks.zfac[1] = zf.y; hooks.zfac[2] = zf.z; hooks.zfac[3] = zf.w;
    }
    SpNoBx nobx;
    auto pre = [&]() {
        if (wave == 0) glds16(ksumT + (size_t)ts.seg * H * DH + 4 * lane, tab);   // [4][64] floats = 1 KiB
        if constexpr (BIAS_TAB) {
            if (wave == 1 && lane < 32) glds16(b0 + rt * T::BM + 4 * lane, btab);   // 128 floats: half a piece
        }
    };
    gemm_mainloop_sp<T, 512 / BK, decltype(apl), decltype(bsl), AttnFoldSp<T::TM, UT>, SpNoBx, ABL, SCHED, decltype(pre), (UT ? SP_OPT_SWAP : 0)>(
        reinterpret_cast<f32x16(&)[T::TM]>(acc), smem_c, apl, bsl, ld, hooks, nobx, &tr, pre, SP_TRACE_ON(trace), T::F32 ? 512 * 4 : 64);
    if (SP_TRACE_ON(trace)) tr.t[9] = __builtin_readcyclecounter();    // behind the loop's last barrier
    hooks.template fold<3>(reinterpret_cast<f32x16(&)[T::TM]>(acc));
    if constexpr (ABL & 32) {   // timing only: no epilogue at all (one store keeps the accumulators alive)
        if (acc[0][0][0] == 123.456f) U[0] = acc[T::TM - 1][0][5];
        return;
    }
    if constexpr (UT) {
        static_assert(BIAS_TAB, "the transposed epilogue reads its bias (one value per lane and 32-channel block) from the LDS table");
        // lane = channel ch[tm] of the workgroup's 128, register r = point pt0 + mfma_row(r, half) of this wave's 32-point strip
        const int pt0 = wn * 32;
        const int vw = min(max(ts.valid - pt0, 0), 32);   // real points of the strip
        float* Ut = U + (size_t)(c0 + pt0) * 512 + rt * T::BM;
        const size_t t32 = (size_t)(c0 + pt0) / 32;
#pragma unroll
        for (int tm = 0; tm < T::TM; ++tm) {
            const int ch = (wm * T::TM + tm) * 32 + l31;
            const float bch = btab[ch];
            float u[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) u[r] = fmaf(acc[tm][0][r], inv, bch);
            // a store instruction = two points x 32 consecutive channels = two full 128-byte lines of U^T
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if constexpr (ABL & 16) { if (u[r] == 123.456f) U[0] = u[r]; }
                else Ut[(size_t)mfma_row(r, half) * 512 + ch] = u[r];
            }
            // (sum, pivot-shifted centred sum of squares) of the strip's real points: mlp0_kernel's partial on a 32-point tile, summed in
            // the lane, the two lane halves combined by one exchange (half 0 first on both)
            const float pivot = __shfl(u[0], l31);   // point pt0 (register 0 of lane half 0)
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float d = mfma_row(r, half) < vw ? u[r] - pivot : 0.f;
                s1 += d;
                s2 = fmaf(d, d, s2);
            }
            const float o1 = __shfl_xor(s1, 32), o2 = __shfl_xor(s2, 32);
            s1 = half ? o1 + s1 : s1 + o1;
            s2 = half ? o2 + s2 : s2 + o2;
            if (half == 0) {
                const float nv = (float)vw;
                statpart[(t32 * 2 + 0) * 512 + rt * T::BM + ch] = nv * pivot + s1;
                statpart[(t32 * 2 + 1) * 512 + rt * T::BM + ch] = nv > 0.f ? s2 - s1 * s1 / nv : 0.f;
            }
        }
        if (SP_TRACE_ON(trace)) { tr.t[10] = tr.t[11] = tr.t[12] = __builtin_readcyclecounter(); }
        (void)statcnt; (void)stats;
        if (SP_TRACE_ON(trace) && lane == 0) {
            unsigned long long* rr = trace + ((size_t)blockIdx.x * T::WAVES + wave) * 24;
            rr[0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
            rr[1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
            rr[2] = t_entry;
#pragma unroll
            for (int k = 0; k < 9; ++k) rr[3 + k] = tr.t[k];
            rr[12] = __builtin_readcyclecounter();
            rr[13] = rt; rr[14] = ct; rr[15] = wave;
#pragma unroll
            for (int k = 0; k < 4; ++k) rr[16 + k] = tr.t[9 + k];
            rr[20] = w_entry; rr[21] = wall_clock64();
        }
        return;
    }
    if constexpr (BIAS_TAB) read_bias16<T>(btab, wm, half, bias);

    // staging tile [BM][BN + 4]: the row stride (4 banks) keeps the scalar writes from the MFMA layout, the 16-byte row reads of the
    // store pass AND (with the walk skew below) the statistics reads free of bank conflicts (the [BN + 1] form of the fp32 kernel costs
    // a 4-way conflict on every read of the store pass: 12 % of this kernel's LDS cycles in profiles/r04_pmc_fp16x4_sq_lds.txt)
    constexpr int TS = T::BN + 4;
    static_assert(T::BM * TS * 4 <= T::RING_BYTES, "the output tile is staged in the ring");
    float* Tl = smem;
#pragma unroll
    for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (wm * T::TM + tm) * 32 + mfma_row(r, half);
            Tl[row * TS + wn * 32 + l31] = fmaf(acc[tm][0][r], inv, bias[tm][r]);
        }
    if (SP_TRACE_ON(trace)) tr.t[10] = __builtin_readcyclecounter();   // last fold + bias + tile written to LDS
    __syncthreads();
    if (SP_TRACE_ON(trace)) tr.t[11] = __builtin_readcyclecounter();
    auto tile_statistics = [&]() {   // per-row (sum, pivot-shifted centred sum of squares) of the real columns of each 64-column tile (mlp0_kernel's form)
        constexpr int LPR = T::THREADS / T::BM;    // lanes per row
        constexpr int LPS = LPR / TPW;             // lanes per (row, 64-column tile)
        constexpr int CPL = MLP0_BN / LPS;         // columns per lane
        static_assert(LPS >= 1, "at least one lane per row and 64-column tile");
        const int row = tid / LPR, q = tid % LPR, sub = q / LPS, part = q % LPS;
        const int valid = min(max(ts.valid - sub * MLP0_BN, 0), MLP0_BN);
        const float pivot = Tl[row * TS + sub * MLP0_BN];
        const float* trow = Tl + row * TS + sub * MLP0_BN + part * CPL;
        // the LPR lanes of a row start a multiple of 32 banks apart and rows are 4 banks apart: lane (row, q) starts its walk
        // q + LPR * ((row >> 3) mod (4 / LPR)) columns into its range, so that the 32 lanes of a read (32 / LPR rows) cover the 32 banks once
        static_assert(TS % 32 == 4, "walk skew of the statistics reads");
        constexpr bool SKEWED = CPL == 32 && (LPR == 4 || LPR == 2);   // (the one-wave-per-SIMD tuning tile walks 64 columns per lane: unskewed)
        const int skew = SKEWED ? q + (LPR == 2 ? 2 * ((row >> 3) & 1) : 0) : 0;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int m0 = 0; m0 < CPL; ++m0) {
            const int m = (m0 + skew) % CPL;
            const float t = trow[m];
            const float d = (part * CPL + m < valid) ? t - pivot : 0.f;
            s1 += d;
            s2 += d * d;
        }
#pragma unroll
        for (int o = 1; o < LPS; o <<= 1) {
            s1 += __shfl_xor(s1, o);
            s2 += __shfl_xor(s2, o);
        }
        if (part == 0) {
            const float nv = (float)valid;
            const size_t t64 = (size_t)ct * TPW + sub;
            stat_partial_store(statpart + (t64 * 2 + 0) * 512 + rt * T::BM + row, nv * pivot + s1);                      // sum
            stat_partial_store(statpart + (t64 * 2 + 1) * 512 + rt * T::BM + row, nv > 0.f ? s2 - s1 * s1 / nv : 0.f);   // M2
        }
    };
    // the tile leaves through LDS as 16-byte stores: 16 lanes cover one 256-byte row segment
    auto tile_stores = [&]() {
#pragma unroll
        for (int idx = tid; idx < T::BM * (T::BN / 4); idx += T::THREADS) {
            const int row = idx / (T::BN / 4), c4 = (idx % (T::BN / 4)) * 4;
            const vf4 v = *reinterpret_cast<const vf4*>(Tl + row * TS + c4);
            if constexpr (ABL & 16) {   // timing only: no global stores of the tile
                if (v[0] == 123.456f) U[0] = v[1];
            } else {
                *reinterpret_cast<vf4*>(U + (size_t)(rt * T::BM + row) * ld + c0 + c4) = v;
            }
        }
    };
    if constexpr (EPI & 1) {   // (stat_final launch only) the tile's stores drain while the statistics are summed
        tile_stores();
        asm volatile("" ::: "memory");
        tile_statistics();
    } else {
        // the partial stores go first; the tile's own stores follow them and may still be in flight when the ticket is drawn
        tile_statistics();
        asm volatile("" ::: "memory");
        tile_stores();
    }
    if (SP_TRACE_ON(trace)) tr.t[12] = __builtin_readcyclecounter();   // tile stores issued
    constexpr int TILE_STORES = T::BM * (T::BN / 4) / T::THREADS;   // per thread, behind its partial stores
    static_assert(T::BM * (T::BN / 4) % T::THREADS == 0, "whole stores per thread");
    if (statcnt) stat_last_block<T, TILE_STORES>(statpart, stats, statcnt, L, ts, rt, smem);   // (nullptr: tuning builds with the stat_final launch)
    if (SP_TRACE_ON(trace) && lane == 0) {   // 24 x u64 per wave: [hw_id, xcc_id, t_entry, t[0..8], t_end, rt, ct, wave, t[9..12], wall clock at entry / exit]
        unsigned long long* r = trace + ((size_t)blockIdx.x * T::WAVES + wave) * 24;
        r[0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        r[1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        r[2] = t_entry;
#pragma unroll
        for (int k = 0; k < 9; ++k) r[3 + k] = tr.t[k];
        r[12] = __builtin_readcyclecounter();
        r[13] = rt; r[14] = ct; r[15] = wave;
#pragma unroll
        for (int k = 0; k < 4; ++k) r[16 + k] = tr.t[9 + k];
        r[20] = w_entry; r[21] = wall_clock64();
    }
}

// =====================================================================================================
// K6  mlp.3:  Z = (Z + b3) + W3 relu((u - mean) * rstd)   (mlp3_kernel of the fp32 path).  The InstanceNorm statistics of the
//     tile's segment sit in an LDS table (mean, rstd x activation pre-scale) and are applied to the raw B values in registers.
// =====================================================================================================
template <int MODE>
using Mlp3SpTile = SpTile<128, 2, 2, 3, MODE>;   // 128 x 64 on 4 waves (252 workgroups at the headline shape)
template <int MODE>
using Mlp3SpTile2 = SpTile<128, 2, 2, 2, MODE>;  // two-stage ring (52 KiB with the statistics table): three workgroups per CU

struct InstNormBx {
    static constexpr bool ON = true;
    const float* tab;   // LDS: mean[512] | rstd[512]
    __device__ __forceinline__ void fetch(int k, float2 (&x)[8]) const {
        const vf4* pm = reinterpret_cast<const vf4*>(tab + k);         // k is a multiple of 8: 32-byte aligned
        const vf4* pr = reinterpret_cast<const vf4*>(tab + 512 + k);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const vf4 m = pm[q], r = pr[q];
#pragma unroll
            for (int e = 0; e < 4; ++e) x[4 * q + e] = make_float2(m[e], r[e]);
        }
    }
    __device__ __forceinline__ float apply(float v, float2 ms) const { return fmaxf((v - ms.x) * ms.y, 0.f); }
};

template <class T, int SCHED = 0, int EPI = 0>
__global__ __launch_bounds__(T::THREADS, (T::F32 ? 4 : T::NST == 2 ? 3 : 2)) void mlp3_sp_kernel(const float* __restrict__ sc, const float* __restrict__ b3,
                                                              const unsigned short* __restrict__ P0, const unsigned short* __restrict__ P1,
                                                              const unsigned short* __restrict__ P2, const float* __restrict__ U,
                                                              const float* __restrict__ stats, float* __restrict__ Z, ColLayout L) {
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    float* smem = reinterpret_cast<float*>(smem_c);
    if constexpr (T::F16) fp16_saturate_mode();
    int rt, ct;
    constexpr int MT = 256 / T::BM;
    constexpr int TPW = T::BN / 64;
    if (!xcd_tile_map_g(MT, active_tiles(L) / TPW, L.xgs, rt, ct)) return;
    ct = global_tile(L, ct * TPW) / TPW;
    const int c0 = ct * T::BN, ld = L.ld;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / T::WN, wn = wave % T::WN, half = lane >> 5, l31 = lane & 31;
    const TileSeg ts = tile_seg(L, c0, T::BN);
    const float sA = T::F16 ? sc[2] : 1.f;
    const float scale = sA * T::ACT_SCALE, inv = 1.f / scale;
    // statistics table behind the ring
    float* tab = reinterpret_cast<float*>(smem_c + T::RING_BYTES);   // mean[512] | rstd[512]: four LDS-DMA pieces in front of the first slabs
    static_assert(T::WAVES >= 4, "statistics table fill: one piece per wave");
    // start from (residual + bias) x the accumulator scale (exact: a power of two)
    f32x16 acc[T::TM][T::TN];
    {
        float bias[T::TM][16];   // (four 16-byte loads per 32-row tile instead of sixteen broadcast dword loads)
        load_bias16<T>(b3, rt * T::BM, wm, half, bias);
#pragma unroll
        for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rt * T::BM + (wm * T::TM + tm) * 32 + mfma_row(r, half);
                acc[tm][0][r] = (Z[(size_t)row * ld + c0 + wn * 32 + l31] + bias[tm][r]) * scale;
            }
    }
    const size_t ro = (size_t)rt * T::BM * BK;
    auto apl = [&](int kt, int pl) -> const void* {
        if constexpr (T::F32) return reinterpret_cast<const float*>(P0) + (size_t)rt * T::BM * 512 + kt * BK;   // fp32 W3 [256][512]
        else return (pl == 0 ? P0 : pl == 1 ? P1 : P2) + ro + (size_t)kt * 256 * BK;
    };
    // EPI bit 3 (BT): U arrives point-major (U^T [ld][512], written by mlp0_sp's transposed epilogue): SP_OPT_BT of the main loop
    constexpr bool BT = (EPI & 8) != 0;
    auto bsl = [&](int kt) { return BT ? U + (size_t)c0 * 512 + kt * BK : U + (size_t)kt * BK * ld + c0; };
    SpPlainHooks<false> hooks;
    InstNormBx bx;
    bx.tab = tab;
    auto pre = [&]() {
        if (wave < 4) glds16(stats + (size_t)ts.seg * 2 * 512 + wave * 256 + 4 * lane, tab + wave * 256);
    };
    gemm_mainloop_sp<T, 512 / BK, decltype(apl), decltype(bsl), SpPlainHooks<false>, InstNormBx, 0, SCHED, decltype(pre), (BT ? SP_OPT_BT : 0)>(
        reinterpret_cast<f32x16(&)[T::TM]>(acc), smem_c, apl, bsl, BT ? 512 : ld, hooks, bx, nullptr, pre, false, T::F32 ? 512 * 4 : 64);
    if constexpr (EPI & 1) store_tile_direct<T>(acc, Z + (size_t)rt * T::BM * ld + c0, ld, [inv](int, float v) { return v * inv; });
    else store_tile_via_lds<T>(acc, smem, Z + (size_t)rt * T::BM * ld + c0, ld, [inv](int, float v) { return v * inv; });
}

// =====================================================================================================
// K8  score contraction + exp (score_exp_kernel of the fp32 path, GATs_SuperGlue.py:217-218) on the split loop, for the fp32-class
//     arithmetics only (bf16x6, fp16x4):  E[n][m] = exp( (sum_d A[n][d] B[d][m]) / scale_factor ),  A = the 16-bit planes of the
//     normalised query descriptors (written by final_proj_norm_kernel: unit-norm rows, fp16 planes of 2^10 x), B = the fp32
//     normalised 3D descriptors, split in registers with the same 2^10.  Same tile (128 x 64), same partial-sum layout and the same
//     epilogue as the fp32 kernel: conf_finalize_kernel cannot tell them apart.
// =====================================================================================================
template <int MODE>
using ScoreSpTile = SpTile<SC_BM, 2, 2, 2, MODE, SCORE_SPLIT_SCALE_LOG2>;

template <class T>
__global__ __launch_bounds__(T::THREADS, 3) void score_exp_sp_kernel(const unsigned short* __restrict__ MDTp, const float* __restrict__ MD,
                                                                   float* __restrict__ conf, float* __restrict__ rowpart,
                                                                   float* __restrict__ colpart, ColLayout L, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    float* smem = reinterpret_cast<float*>(smem_c);
    static_assert(T::BN == SC_BN && T::BM == SC_BM, "partial sums are per 128 x 64 tile");
    if constexpr (T::F16) fp16_saturate_mode();
    const int nrt = L.n1p / T::BM, nct = L.n2p / T::BN;   // segments are padded to multiples of 128
    int rt, ct;
    const int frame = blockIdx.y;
    if (!xcd_tile_map(nrt, nct, rt, ct)) return;
    const int ld = L.ld;
    const size_t R = (size_t)L.b * L.n1p;
    const size_t m0 = (size_t)frame * L.n1p + (size_t)rt * T::BM;
    const float* Bp = MD + (size_t)frame * L.np + L.n1p + ct * T::BN;
    f32x16 acc[T::TM][T::TN];
    auto apl = [&](int kt, int pl) { return MDTp + (size_t)pl * R * D + ((size_t)kt * R + m0) * BK; };
    auto bsl = [&](int kt) { return Bp + (size_t)kt * BK * ld; };
    SpPlainHooks<true> hooks;
    SpNoBx nobx;
    constexpr int SS = T::F16 ? 4 : 0;   // fp16x4: the slot schedule (gemm_split_glds.h); bf16x6: the plain one
    gemm_mainloop_sp<T, D / BK, decltype(apl), decltype(bsl), SpPlainHooks<true>, SpNoBx, 0, SS>(reinterpret_cast<f32x16(&)[T::TM]>(acc), smem_c, apl, bsl, ld,
                                                                                                hooks, nobx);
    const float inv = T::F16 ? 1.f / (T::ACT_SCALE * T::ACT_SCALE) : 1.f;   // both operands carry the scale
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / T::WN, wn = wave % T::WN, half = lane >> 5, l31 = lane & 31;
    constexpr int TS = T::BN + 4;   // conflict-free 16-byte row reads (see mlp0_sp_kernel)
    static_assert(T::BM * TS * 4 <= T::RING_BYTES, "the output tile is staged in the ring");
    float* Tl = smem;  // [128][68]
    float* cf = conf + (size_t)frame * L.n1 * L.n2;
#pragma unroll
    for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (wm * T::TM + tm) * 32 + mfma_row(r, half);
            const int col = wn * 32 + l31;
            const int gi = rt * T::BM + row, gj = ct * T::BN + col;
            const float sc = (acc[tm][0][r] * inv) / scale;
            Tl[row * TS + col] = (gi < L.n1 && gj < L.n2) ? expf(sc) : 0.f;
        }
    __syncthreads();
    if ((L.n2 & 3) == 0 && (reinterpret_cast<uintptr_t>(cf) & 15) == 0) {
        for (int idx = tid; idx < T::BM * (T::BN / 4); idx += T::THREADS) {
            const int row = idx / (T::BN / 4), c4 = (idx % (T::BN / 4)) * 4;
            const int gi = rt * T::BM + row, gj = ct * T::BN + c4;
            if (gi < L.n1 && gj < L.n2) *reinterpret_cast<vf4*>(cf + (size_t)gi * L.n2 + gj) = *reinterpret_cast<const vf4*>(Tl + row * TS + c4);
        }
    } else {
        for (int idx = tid; idx < T::BM * T::BN; idx += T::THREADS) {
            const int row = idx / T::BN, col = idx % T::BN;
            const int gi = rt * T::BM + row, gj = ct * T::BN + col;
            if (gi < L.n1 && gj < L.n2) cf[(size_t)gi * L.n2 + gj] = Tl[row * TS + col];
        }
    }
    {   // row sums: THREADS / BM lanes per row; column sums: THREADS / BN row groups, one thread per (group, column); fixed order
        constexpr int LPR = T::THREADS / T::BM, CPL = T::BN / LPR;
        static_assert(LPR == 2 && CPL == 32, "walk skew of the row sums (rows 4 banks apart, the two lanes of a row 32 banks apart)");
        const int row = tid / LPR, hp = tid % LPR;
        const float* tr = Tl + row * TS + hp * CPL;
        const int skew = hp + 2 * ((row >> 3) & 1);
        float s = 0.f;
#pragma unroll 8
        for (int m = 0; m < CPL; ++m) s += tr[(m + skew) % CPL];
#pragma unroll
        for (int o = 1; o < LPR; o <<= 1) s += __shfl_xor(s, o);
        if (hp == 0 && rt * T::BM + row < L.n1p) rowpart[((size_t)frame * nct + ct) * L.n1p + rt * T::BM + row] = s;
        constexpr int NQ = T::THREADS / T::BN, RPQ = T::BM / NQ;
        const int c = tid % T::BN, qp = tid / T::BN;
        float t = 0.f;
#pragma unroll 8
        for (int m = 0; m < RPQ; ++m) t += Tl[(qp * RPQ + m) * TS + c];
        __syncthreads();
        Tl[qp * T::BN + c] = t;
        __syncthreads();
        if (tid < T::BN) {
            float tot = 0.f;
#pragma unroll
            for (int q = 0; q < NQ; ++q) tot += Tl[q * T::BN + tid];
            colpart[((size_t)frame * nrt + rt) * L.n2p + ct * T::BN + tid] = tot;
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------------------
template <int MODE>
static void launch_score_sp_t(const Workspace& w, float* conf, float scale, hipStream_t s, ProfileHook* hk) {
    using T = ScoreSpTile<MODE>;
    allow_big_lds_sp<score_exp_sp_kernel<T>>();
    GATSSPG_LAUNCH(hk, KID_SCORE_EXP, s, (score_exp_sp_kernel<T>), dim3(xcd_grid(w.L.n1p / T::BM, w.L.n2p / T::BN), w.L.b), dim3(T::THREADS),
                   (size_t)T::RING_BYTES, s, w.MDTp, w.MD, conf, w.rowpart, w.colpart, w.L, scale);
}
void launch_score_exp_sp(const Workspace& w, float* conf, float scale, hipStream_t s, ProfileHook* hk) {
    if (w.prec == 2) launch_score_sp_t<2>(w, conf, scale, s, hk);
    else launch_score_sp_t<4>(w, conf, scale, s, hk);
}

struct PlaneSet {
    const unsigned short *p0, *p1, *p2;
};
static PlaneSet planes(const unsigned short* wb, int prec, size_t hi, size_t lo, size_t lo2, size_t h16, size_t l16) {
    if (prec >= 3) return {wb + h16, wb + l16, wb + l16};
    return {wb + hi, wb + lo, wb + lo2};
}

// schedule of the split loop in the fp16 modes (gemm_split_glds.h): 4 (default) = the slot schedule -- every MFMA of a step carries its share of the
// step's VALU work, DMA requests and LDS reads, fenced slot by slot; 3 = the same with the next slab's reads in two bursts; 2 = DMA requests spread over
// the step, split work left to hipcc's grouping; 0 = DMA requests in one burst behind the barrier; 1 = ping-pong wave groups.
// The PRODUCT library instantiates only what it runs: the fp16 modes on schedule 4 with direct stores (and the score kernel).  Every alternative --
// other schedules, staged stores, ring depths, the bf16 modes and the exact fp32 arithmetic on this loop, the timing ablations -- is compiled into the
// tuning build only (TUNING_BUILD), where the knobs below are read from the environment per launch (tools/ab_live.py flips them inside one process).
#ifdef GATSSPG_TUNING
constexpr bool TUNING_BUILD = true;
#else
constexpr bool TUNING_BUILD = false;
#endif
constexpr int SP_SCHED_DEFAULT = 4;          // profiles/r04_ab_live_*.txt: 2 is 1.0-1.4 % faster per frame than 0, 3 another 1.0-1.6 %, 4 another 0.4 %
constexpr int SP_DIRECT_STORE_DEFAULT = 3;   // bit 0 the Q tiles of qkv_kv, bit 1 mlp3 leave straight from the accumulators (+0.5 % per frame); bit 2 = mlp0's
                                             // tile stores in front of its statistics (neutral)
static int sp_direct_store() { return tuning_knob("SP_DIRECT_STORE", SP_DIRECT_STORE_DEFAULT); }
static int sp_sched() { return tuning_knob("SP_SCHED", SP_SCHED_DEFAULT); }
// 1: qkv_kv / mlp0 read their bias from an LDS table filled by one LDS-DMA piece (no global loads in front of the first slab requests); 0: per-lane
// 16-byte global loads before the loop (schedule 4 only; the other schedules keep the loads)
static int sp_bias_table() { return tuning_knob("SP_BIAS_TABLE", 1); }
// fp16 modes, TUNING BUILDS ONLY (GATSSPG_SP_UT=1): mlp.0 leaves U point-major from TRANSPOSED accumulators (swapped MFMA operands: no staging tile,
// no barrier, InstanceNorm partials summed inside a lane per 32-point strip) and mlp.3 reads it as a transposed B operand (two 16-byte LDS reads
// per k16 half instead of eight 4-byte ones).  Built, parity-green (zero arg-max flips, the same conf error) and measured in round 5
// (profiles/r05c_ab_live_ut_xcd_direct.txt, settings interleaved in one process): mlp0 23.5 vs 23.8 us event-timed, mlp3 16.7 vs 16.9 -- but
// 0.693 vs 0.688 ms per frame and 1908 vs 1925 frames/s in flight: the transposed fold needs the per-point factor in 16 registers (one rank-1
// fp32 MFMA per head), the kernel holds 237-243 registers instead of 192 (no other frame's wave fits beside it) and stat_final merges twice the
// partials.  The product library runs the channel-major pair; the alternative schedules / tiles / ablations exist in that form only.
bool sp_ut_on(int prec) {
    if constexpr (!TUNING_BUILD) return false;
    if (prec < 3) return false;
    const int nst2 = tuning_knob("SP_NST2", -1);
    if (sp_sched() != 4 || !sp_bias_table() || (sp_direct_store() & 2) == 0 || tuning_knob("SP_ABL", 0) != 0 || (nst2 > 0 && (nst2 & 1)) || stat_fused())
        return false;
    return tuning_knob("SP_UT", 0) != 0;
}

// mlp0's tile choice (launch_mlp0_sp_m) and, from it, the XCD granule of the 64-column kernels beside it: with the 128-column mlp0 tile an XCD
// can own PAIRS of 64-column tiles in qkv_kv / mlp3 as well, so that a column range stays on the XCD (= the L2) that produced it across
// mlp3 -> qkv_kv -> mlp0 -> mlp3.  Measured (profiles/r05c_ab_live_ut_xcd_direct.txt, interleaved in one process): no effect (0.6927 vs 0.6928 ms per
// frame, 1912 vs 1908 frames/s in flight) -- the operands of these launches are not waiting for a remote L2.  Off; GATSSPG_SP_XCD_PAIR=1 in tuning builds.
static bool mlp0_sp_wide(const ColLayout& L) {
    const int wide_min = tuning_knob("SP_MLP0_WIDE_MIN", 48), wide_max = tuning_knob("SP_MLP0_WIDE_MAX", 64);
    return active_tiles(L) / 2 >= wide_min && active_tiles(L) / 2 <= wide_max;
}
static ColLayout sp_paired_layout(const ColLayout& L0, int prec) {
    ColLayout L = L0;
    L.xgs = (prec >= 3 && mlp0_sp_wide(L0) && tuning_knob("SP_XCD_PAIR", 0) != 0) ? 1 : 0;
    return L;
}

template <class T, int SCHED, int EPI>
static void launch_qkv_sp_v(const float* sc, const float* bqkv, const PlaneSet& p, const Workspace& w, hipStream_t s, ProfileHook* hk) {
    allow_big_lds_sp<qkv_kv_sp_kernel<T, SCHED, EPI>>();
    const ColLayout L = sp_paired_layout(w.L, T::MODE);
    GATSSPG_LAUNCH(hk, KID_QKV_KV, s, (qkv_kv_sp_kernel<T, SCHED, EPI>), dim3(xcd_grid_g(6, active_tiles(L), L.xgs)), dim3(T::THREADS), (size_t)T::RING_BYTES + 1024, s,
                   sc, bqkv, p.p0, p.p1, p.p2, w.Z, w.Q, w.kvpart, L);
}
template <int MODE>
static void launch_qkv_sp_t(const float* sc, const float* bqkv, const unsigned short* wb, const Workspace& w, hipStream_t s, ProfileHook* hk) {
    using T = QkvSpTile<MODE>;
    const PlaneSet p = planes(wb, MODE, AttnWB::QKV_HI, AttnWB::QKV_LO, AttnWB::QKV_LO2, AttnWB::QKV_H16, AttnWB::QKV_L16);
    if constexpr (MODE >= 3) {
        if constexpr (TUNING_BUILD) {
            const bool direct = sp_direct_store() & 1;
            switch (sp_sched()) {
                case 4:
                    if (!direct) return launch_qkv_sp_v<T, 4, 0>(sc, bqkv, p, w, s, hk);
                    if (!sp_bias_table()) return launch_qkv_sp_v<T, 4, 1>(sc, bqkv, p, w, s, hk);
                    break;
                case 3: return launch_qkv_sp_v<T, 3, 1>(sc, bqkv, p, w, s, hk);
                case 2: return direct ? launch_qkv_sp_v<T, 2, 1>(sc, bqkv, p, w, s, hk) : launch_qkv_sp_v<T, 2, 0>(sc, bqkv, p, w, s, hk);
                default: return launch_qkv_sp_v<T, 0, 0>(sc, bqkv, p, w, s, hk);
            }
        }
        launch_qkv_sp_v<T, SP_SCHED_DEFAULT, 3>(sc, bqkv, p, w, s, hk);   // direct Q stores, bias through its LDS table
    } else {
        launch_qkv_sp_v<T, 0, 0>(sc, bqkv, p, w, s, hk);   // bf16 modes (tuning builds: GATSSPG_SPLIT_LOOP_BF16X3 / _BF16X6)
    }
}
void launch_qkv_kv_sp(const float* sc, const float* bqkv, const unsigned short* wb, const Workspace& w, hipStream_t s, ProfileHook* hk) {
    switch (w.prec) {
#ifdef GATSSPG_TUNING
        case 1: launch_qkv_sp_t<1>(sc, bqkv, wb, w, s, hk); break;
        case 2: launch_qkv_sp_t<2>(sc, bqkv, wb, w, s, hk); break;
#endif
        case 3: launch_qkv_sp_t<3>(sc, bqkv, wb, w, s, hk); break;
        default: launch_qkv_sp_t<4>(sc, bqkv, wb, w, s, hk); break;
    }
}
// fp32 (MODE 0) on this loop: the operators are the fp32 matrices themselves.  Tuning builds only (GATSSPG_FP32_DMA; measured 8 % slower per frame
// than the register-staged fp32 loop of gemm_f32_mfma.h): the product library never calls these.
void launch_qkv_kv_dma(const float* Wqkv, const float* bqkv, const Workspace& w, hipStream_t s, ProfileHook* hk) {
#ifdef GATSSPG_TUNING
    using T = Fp32SpTile;
    allow_big_lds_sp<qkv_kv_sp_kernel<T, 2>>();
    GATSSPG_LAUNCH(hk, KID_QKV_KV, s, (qkv_kv_sp_kernel<T, 2>), dim3(xcd_grid(6, active_tiles(w.L))), dim3(T::THREADS), (size_t)T::RING_BYTES + 1024, s, Wqkv,
                   bqkv, reinterpret_cast<const unsigned short*>(Wqkv), nullptr, nullptr, w.Z, w.Q, w.kvpart, w.L);
#else
    (void)Wqkv; (void)bqkv; (void)w; (void)s; (void)hk;
#endif
}
void launch_mlp0_dma(const float* W0, const float* b0, const Workspace& w, hipStream_t s, ProfileHook* hk) {
#ifdef GATSSPG_TUNING
    using T = Fp32SpTile;
    allow_big_lds_sp<mlp0_sp_kernel<T, 0, 2>>();
    GATSSPG_LAUNCH(hk, KID_MLP0, s, (mlp0_sp_kernel<T, 0, 2>), dim3(xcd_grid(512 / T::BM, active_tiles(w.L))), dim3(T::THREADS),
                   (size_t)T::RING_BYTES + 2048, s, W0, b0, reinterpret_cast<const unsigned short*>(W0), nullptr, nullptr, w.Z, w.Q,
                   reinterpret_cast<const unsigned short*>(w.Mop), w.ksumT, w.zsc, w.U, w.statpart, w.stats, stat_fused() ? w.statcnt : nullptr, w.L, g_trace);
#else
    (void)W0; (void)b0; (void)w; (void)s; (void)hk;
#endif
}
void launch_mlp3_dma(const float* W3, const float* b3, const Workspace& w, hipStream_t s, ProfileHook* hk) {
#ifdef GATSSPG_TUNING
    using T = Fp32SpTile;
    allow_big_lds_sp<mlp3_sp_kernel<T, 2>>();
    GATSSPG_LAUNCH(hk, KID_MLP3, s, (mlp3_sp_kernel<T, 2>), dim3(xcd_grid(256 / T::BM, active_tiles(w.L))), dim3(T::THREADS), (size_t)T::RING_BYTES + 4096, s,
                   W3, b3, reinterpret_cast<const unsigned short*>(W3), nullptr, nullptr, w.U, w.stats, w.Z, w.L);
#else
    (void)W3; (void)b3; (void)w; (void)s; (void)hk;
#endif
}

template <class T, int ABL = 0, int SCHED = 0, int EPI = 0>
static void launch_mlp0_sp_t(const float* sc, const float* b0, const unsigned short* wb, const Workspace& w, hipStream_t s, ProfileHook* hk) {
    const PlaneSet p = planes(wb, T::MODE, AttnWB::W0_HI, AttnWB::W0_LO, AttnWB::W0_LO2, AttnWB::W0_H16, AttnWB::W0_L16);
    const int NT = active_tiles(w.L) / (T::BN / MLP0_BN);
    allow_big_lds_sp<mlp0_sp_kernel<T, ABL, SCHED, EPI>>();
    GATSSPG_LAUNCH(hk, KID_MLP0, s, (mlp0_sp_kernel<T, ABL, SCHED, EPI>), dim3(xcd_grid(512 / T::BM, NT)), dim3(T::THREADS), (size_t)T::RING_BYTES + 2048, s, sc, b0,
                   p.p0, p.p1, p.p2, w.Z, w.Q, w.Mpl, w.ksumT, w.zsc, w.U, w.statpart, w.stats, (!(EPI & 1) && stat_fused()) ? w.statcnt : nullptr, w.L, g_trace);
}
template <int MODE>
static void launch_mlp0_sp_m(const float* sc, const float* b0, const unsigned short* wb, const Workspace& w, hipStream_t s, ProfileHook* hk) {
    // the 128-column tile (8 waves, one workgroup per CU) moves two thirds of the operand bytes per product through L2: taken when
    // its 4 x tiles workgroups make ONE round of the 256 CUs and fill at least three quarters of it (the headline shape: 252); with
    // fewer the 64-column tile (4 waves, two workgroups per CU, twice as many) fills the chip better, with more than one round its
    // co-resident pairs overlap one workgroup's store tail with the other's loop (fp16x4, 8 frames per step: 173 vs 185 us per launch)
    const bool wide = mlp0_sp_wide(w.L);
    if constexpr (TUNING_BUILD) {
        if constexpr (MODE == 4) {   // timing-only ablations of the main loop (wrong results) and the alternative tiles
            switch (tuning_knob("SP_ABL", 0)) {
                case 1: return launch_mlp0_sp_t<Mlp0SpTileW<MODE>, 1>(sc, b0, wb, w, s, hk);     // no DMA after the prologue
                case 2: return launch_mlp0_sp_t<Mlp0SpTileW<MODE>, 2>(sc, b0, wb, w, s, hk);     // no MFMAs
                case 4: return launch_mlp0_sp_t<Mlp0SpTileW<MODE>, 4>(sc, b0, wb, w, s, hk);     // no split VALU
                case 8: return launch_mlp0_sp_t<Mlp0SpTileW<MODE>, 8>(sc, b0, wb, w, s, hk);     // no fragment reads
                case 14: return launch_mlp0_sp_t<Mlp0SpTileW<MODE>, 14>(sc, b0, wb, w, s, hk);   // DMA + barriers only
                case 13: return launch_mlp0_sp_t<Mlp0SpTileW<MODE>, 13>(sc, b0, wb, w, s, hk);   // MFMAs + barriers only
                case 15: return launch_mlp0_sp_t<Mlp0SpTileW<MODE>, 15>(sc, b0, wb, w, s, hk);   // barriers only (+ prologue, epilogue)
                case 9: return launch_mlp0_sp_t<Mlp0SpTileW<MODE>, 9>(sc, b0, wb, w, s, hk);     // no DMA, no reads: MFMAs + split
                case 31: return launch_mlp0_sp_t<Mlp0SpTileW<MODE>, 31>(sc, b0, wb, w, s, hk);   // skeleton without the tile's global stores
                case 63: return launch_mlp0_sp_t<Mlp0SpTileW<MODE>, 63>(sc, b0, wb, w, s, hk);   // skeleton without any epilogue
                case 16: return launch_mlp0_sp_t<Mlp0SpTileW<MODE>, 16>(sc, b0, wb, w, s, hk);   // full loop, no global stores of the tile
                case 48: return launch_mlp0_sp_t<Mlp0SpTileW<MODE>, 48>(sc, b0, wb, w, s, hk);   // full loop, no epilogue
                case 100: return launch_mlp0_sp_t<Mlp0SpTileT<MODE>, 0>(sc, b0, wb, w, s, hk);   // (not an ablation) the one-wave-per-SIMD tile
                case 101: return launch_mlp0_sp_t<Mlp0SpTileW<MODE>, 0, 1>(sc, b0, wb, w, s, hk);   // (not an ablation) the ping-pong schedule
                case 102: return launch_mlp0_sp_t<Mlp0SpTileN<MODE>, 0, 1>(sc, b0, wb, w, s, hk);   // ping-pong on the 4-wave tile (groups on different SIMDs)
                default: break;
            }
            const int nst2 = tuning_knob("SP_NST2", -1);
            if (nst2 > 0 && (nst2 & 1)) return launch_mlp0_sp_t<Mlp0SpTileN2<MODE>, 0, 2>(sc, b0, wb, w, s, hk);   // two-stage ring, three workgroups per CU
        }
        if constexpr (MODE >= 3) {
            switch (sp_sched()) {
                case 4:
                    if (sp_bias_table()) break;
                    return wide ? launch_mlp0_sp_t<Mlp0SpTileW<MODE>, 0, 4>(sc, b0, wb, w, s, hk) : launch_mlp0_sp_t<Mlp0SpTileN<MODE>, 0, 4>(sc, b0, wb, w, s, hk);
                case 3:
                    if (wide && (tuning_knob("SP_NST4", 0) & 1)) return launch_mlp0_sp_t<Mlp0SpTileW4<MODE>, 0, 3>(sc, b0, wb, w, s, hk);   // four-stage ring
                    return wide ? launch_mlp0_sp_t<Mlp0SpTileW<MODE>, 0, 3>(sc, b0, wb, w, s, hk) : launch_mlp0_sp_t<Mlp0SpTileN<MODE>, 0, 3>(sc, b0, wb, w, s, hk);
                case 2:
                    if ((sp_direct_store() & 4) && !stat_fused())   // tile stores in front of the statistics
                        return wide ? launch_mlp0_sp_t<Mlp0SpTileW<MODE>, 0, 2, 1>(sc, b0, wb, w, s, hk) : launch_mlp0_sp_t<Mlp0SpTileN<MODE>, 0, 2, 1>(sc, b0, wb, w, s, hk);
                    return wide ? launch_mlp0_sp_t<Mlp0SpTileW<MODE>, 0, 2>(sc, b0, wb, w, s, hk) : launch_mlp0_sp_t<Mlp0SpTileN<MODE>, 0, 2>(sc, b0, wb, w, s, hk);
                default:
                    return wide ? launch_mlp0_sp_t<Mlp0SpTileW<MODE>>(sc, b0, wb, w, s, hk) : launch_mlp0_sp_t<Mlp0SpTileN<MODE>>(sc, b0, wb, w, s, hk);
            }
        }
    }
    if constexpr (MODE >= 3) {
        if constexpr (TUNING_BUILD) {
            if (sp_ut_on(MODE)) {   // EPI 2 | 8: bias through its LDS table, transposed accumulators -> U^T + in-lane statistics
                if (wide) launch_mlp0_sp_t<Mlp0SpTileW<MODE>, 0, SP_SCHED_DEFAULT, 10>(sc, b0, wb, w, s, hk);
                else launch_mlp0_sp_t<Mlp0SpTileN<MODE>, 0, SP_SCHED_DEFAULT, 10>(sc, b0, wb, w, s, hk);
                return;
            }
        }
        if (wide) launch_mlp0_sp_t<Mlp0SpTileW<MODE>, 0, SP_SCHED_DEFAULT, 2>(sc, b0, wb, w, s, hk);   // (EPI 2: bias through its LDS table)
        else launch_mlp0_sp_t<Mlp0SpTileN<MODE>, 0, SP_SCHED_DEFAULT, 2>(sc, b0, wb, w, s, hk);
    } else {   // bf16 modes (tuning builds)
        if (wide) launch_mlp0_sp_t<Mlp0SpTileW<MODE>>(sc, b0, wb, w, s, hk);
        else launch_mlp0_sp_t<Mlp0SpTileN<MODE>>(sc, b0, wb, w, s, hk);
    }
}
void launch_mlp0_sp(const float* sc, const float* b0, const unsigned short* wb, const Workspace& w, hipStream_t s, ProfileHook* hk) {
    switch (w.prec) {
#ifdef GATSSPG_TUNING
        case 1: launch_mlp0_sp_m<1>(sc, b0, wb, w, s, hk); break;
        case 2: launch_mlp0_sp_m<2>(sc, b0, wb, w, s, hk); break;
#endif
        case 3: launch_mlp0_sp_m<3>(sc, b0, wb, w, s, hk); break;
        default: launch_mlp0_sp_m<4>(sc, b0, wb, w, s, hk); break;
    }
}

template <class T, int SCHED, int EPI>
static void launch_mlp3_sp_v(const float* sc, const float* b3, const PlaneSet& p, int NT, const Workspace& w, hipStream_t s, ProfileHook* hk) {
    allow_big_lds_sp<mlp3_sp_kernel<T, SCHED, EPI>>();
    const ColLayout L = sp_paired_layout(w.L, T::MODE);
    GATSSPG_LAUNCH(hk, KID_MLP3, s, (mlp3_sp_kernel<T, SCHED, EPI>), dim3(xcd_grid_g(256 / T::BM, NT, L.xgs)), dim3(T::THREADS), (size_t)T::RING_BYTES + 4096, s, sc, b3,
                   p.p0, p.p1, p.p2, w.U, w.stats, w.Z, L);
}
// the schedule / store variants of one mlp3 tile (tuning builds); false = take the default
template <class T>
static bool launch_mlp3_sp_alt(const float* sc, const float* b3, const PlaneSet& p, int NT, const Workspace& w, hipStream_t s, ProfileHook* hk) {
    const bool direct = sp_direct_store() & 2;
    switch (sp_sched()) {
        case 4: if (direct) return false; launch_mlp3_sp_v<T, 4, 0>(sc, b3, p, NT, w, s, hk); return true;
        case 3: launch_mlp3_sp_v<T, 3, 1>(sc, b3, p, NT, w, s, hk); return true;
        case 2: if (direct) launch_mlp3_sp_v<T, 2, 1>(sc, b3, p, NT, w, s, hk); else launch_mlp3_sp_v<T, 2, 0>(sc, b3, p, NT, w, s, hk); return true;
        default: launch_mlp3_sp_v<T, 0, 0>(sc, b3, p, NT, w, s, hk); return true;
    }
}
template <int MODE>
static void launch_mlp3_sp_t(const float* sc, const float* b3, const unsigned short* wb, const Workspace& w, hipStream_t s, ProfileHook* hk) {
    using T = Mlp3SpTile<MODE>;
    const PlaneSet p = planes(wb, MODE, AttnWB::W3_HI, AttnWB::W3_LO, AttnWB::W3_LO2, AttnWB::W3_H16, AttnWB::W3_L16);
    const int NT = active_tiles(w.L) / (T::BN / 64);
    if constexpr (MODE >= 3) {
        // more than one round of the three-stage ring's two workgroups per CU (batched frames, N_3D = 20000): the two-stage ring's three
        // per CU turn 1.46 rounds into one at 8 frames per step (fp16x4-b8: 0.565 vs 0.571 ms per frame, profiles/r04_ab_live_b8_tiles.txt)
        using T2 = Mlp3SpTile2<MODE>;
        const int nst2 = tuning_knob("SP_NST2", -1);
        const bool two_stage = nst2 >= 0 ? (nst2 & 2) != 0 : (256 / T2::BM) * NT > 512;
        if constexpr (TUNING_BUILD) {
            if (two_stage ? launch_mlp3_sp_alt<T2>(sc, b3, p, NT, w, s, hk) : launch_mlp3_sp_alt<T>(sc, b3, p, NT, w, s, hk)) return;
        }
        if constexpr (TUNING_BUILD) {
            if (sp_ut_on(MODE)) {   // EPI 1 | 8: direct stores, U read point-major
                if (two_stage) launch_mlp3_sp_v<T2, SP_SCHED_DEFAULT, 9>(sc, b3, p, NT, w, s, hk);
                else launch_mlp3_sp_v<T, SP_SCHED_DEFAULT, 9>(sc, b3, p, NT, w, s, hk);
                return;
            }
        }
        if (two_stage) launch_mlp3_sp_v<T2, SP_SCHED_DEFAULT, 1>(sc, b3, p, NT, w, s, hk);
        else launch_mlp3_sp_v<T, SP_SCHED_DEFAULT, 1>(sc, b3, p, NT, w, s, hk);
    } else {
        launch_mlp3_sp_v<T, 0, 0>(sc, b3, p, NT, w, s, hk);   // bf16 modes (tuning builds)
    }
}
void launch_mlp3_sp(const float* sc, const float* b3, const unsigned short* wb, const Workspace& w, hipStream_t s, ProfileHook* hk) {
    switch (w.prec) {
#ifdef GATSSPG_TUNING
        case 1: launch_mlp3_sp_t<1>(sc, b3, wb, w, s, hk); break;
        case 2: launch_mlp3_sp_t<2>(sc, b3, wb, w, s, hk); break;
#endif
        case 3: launch_mlp3_sp_t<3>(sc, b3, wb, w, s, hk); break;
        default: launch_mlp3_sp_t<4>(sc, b3, wb, w, s, hk); break;
    }
}

}  // namespace gatsspg
