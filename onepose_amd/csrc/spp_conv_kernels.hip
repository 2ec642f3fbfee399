// SuperPoint extractor, dense stages (reference: src/models/extractors/SuperPoint/superpoint.py:142-158,183-184).
//
//   conv1a            direct 1 -> 64 channel 3x3 (HBM-bound: one 67 MB plane set written)
//   conv3x3 / conv1x1 implicit GEMM on the fp32 MFMA main loop of gemm_f32_mfma.h: the activation planes are
//                     stored zero-padded and flattened (spp_common.h), so tap (dy, dx) is the same [Cin][N]
//                     matrix shifted by dy*Wp + dx columns and the convolution is ONE GEMM with
//                     K = taps * Cin whose B slabs are column-shifted views (4-byte aligned loads).
//   maxpool 2x2       streaming kernel between resolutions.
#include <stdlib.h>
#include <type_traits>

#include "gemm_f32_mfma.h"
#include "spp_common.h"
#include "../../include/superpoint.h"

namespace spp {

using gatsspg::BK;
using gatsspg::f32x16;
using gatsspg::GemmTile;
using gatsspg::mfma_row;

// PREC: 0 = fp32 MFMA main loop; 4 = four-term split-fp16 (gemm_mainloop_bf3, F16, NP = 4: two fp16 terms per operand, all four
// products on v_mfma_f32_32x32x16_f16 -- fp32-class results, a quarter of the matrix-pipe time; weights pre-split at pack time,
// activations split in the loop)
template <class T, int PREC = 0>
constexpr size_t smem_bytes() {
    size_t b = sizeof(float) * T::SMEM_FLOATS;
    if constexpr (PREC == 4) {
        if (gatsspg::Bf3Layout<T>::SMEM_BYTES > b) b = gatsspg::Bf3Layout<T>::SMEM_BYTES;
    }
    return b;
}

// =====================================================================================================
// conv1a + ReLU (:142): one thread per padded position, all 64 output channels
// =====================================================================================================
// zero the one-pixel pad ring of every channel plane of P (the patch-tiled convolutions write pixels only)
__device__ void zero_pad_ring(float* __restrict__ P, const FeatLayout& L, int C, int worker, int nworkers) {
    const int ring = 2 * L.Wp + 2 * L.H;                 // top row, bottom row, left + right columns
    const int total = C * L.b * ring;                    // < 2^31: C * b * ring <= 512 * b * 2056
    for (int i = worker; i < total; i += nworkers) {
        const int ci = i / ring, r = i - ci * ring;
        const int c = ci / L.b, im = ci - c * L.b;
        int q;
        if (r < L.Wp) q = r;                                                   // y = 0
        else if (r < 2 * L.Wp) q = (L.H + 1) * L.Wp + (r - L.Wp);              // y = H + 1
        else {
            const int k = r - 2 * L.Wp, y = 1 + (k >> 1);
            q = y * L.Wp + ((k & 1) ? L.Wp - 1 : 0);
        }
        P[(size_t)c * L.ldt + (size_t)im * L.ld + q] = 0.f;
    }
}

struct PadPlanes {
    static constexpr int N = 6;
    float* p[N];
    FeatLayout L[N];
    int C[N];
    int nblocks;      // workgroups of conv1a_kernel (after the image ones) that do this
};

__global__ __launch_bounds__(256) void conv1a_kernel(const float* __restrict__ img, const float* __restrict__ w9,
                                                     const float* __restrict__ bias, float* __restrict__ Y, FeatLayout L,
                                                     PadPlanes pp) {
    const int img_blocks = (L.ld + 255) / 256;
    if ((int)blockIdx.x >= img_blocks) {
        if (blockIdx.y == 0) {
            const int worker = (blockIdx.x - img_blocks) * 256 + threadIdx.x, nworkers = pp.nblocks * 256;
#pragma unroll
            for (int i = 0; i < PadPlanes::N; ++i) zero_pad_ring(pp.p[i], pp.L[i], pp.C[i], worker, nworkers);
        }
        return;
    }
    __shared__ float sw[64 * 9 + 64];
    for (int i = threadIdx.x; i < 64 * 9; i += 256) sw[i] = w9[i];
    if (threadIdx.x < 64) sw[576 + threadIdx.x] = bias[threadIdx.x];
    __syncthreads();
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int im = blockIdx.y;
    if (q >= L.ld) return;
    int y, x;
    const bool ok = feat_valid(L, q, y, x);
    float v[9];
    if (ok) {
        const float* src = img + (size_t)im * L.H * L.W;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = y - 1 + t / 3 - 1, xx = x - 1 + t % 3 - 1;   // pixel coordinates of the tap
            v[t] = (yy >= 0 && yy < L.H && xx >= 0 && xx < L.W) ? src[(size_t)yy * L.W + xx] : 0.f;
        }
    }
    float* dst = Y + (size_t)im * L.ld + q;
#pragma unroll 8
    for (int c = 0; c < 64; ++c) {
        float a = 0.f;
        if (ok) {
            a = sw[576 + c];
#pragma unroll
            for (int t = 0; t < 9; ++t) a = fmaf(sw[c * 9 + t], v[t], a);
            a = fmaxf(a, 0.f);
        }
        dst[(size_t)c * L.ldt] = a;
    }
}

// =====================================================================================================
// 3x3 / 1x1 convolution + bias (+ ReLU) as an implicit GEMM.  Wt [rows][TAPS*CIN] (k = tap*CIN + ci),
// X [CIN][ldt] -> Y [cout][ldt].
//   PATCH = false: the BN columns of a workgroup are BN consecutive positions of the flattened padded plane (pad
//                  positions included; they are written as zeros).  Used by the 1x1 convolutions.
//   PATCH = true : the BN columns are a 2-row x BN/2-pixel image patch (column map of the main loop): the three dy taps
//                  of the two rows touch 4 input rows instead of 6, i.e. a third fewer L1-missing bytes per MFMA, which
//                  is what the main loop is sensitive to.  Only pixels are written; the pad ring of the output plane is
//                  zeroed once per forward by conv1a_kernel's extra workgroups.
// =====================================================================================================
struct PatchCol {
    int half, Wp;     // half = pixels per patch row (BN / 2)
    __device__ __forceinline__ int operator()(int c) const { return c < half ? c : c - half + Wp; }
};

template <class T, int CIN, int TAPS, bool PATCH, int PREC = 0>
__global__ __launch_bounds__(T::THREADS) void conv_gemm_kernel(const float* __restrict__ Wt, const float* __restrict__ bias,
                                                               const unsigned short* __restrict__ Wp16, int rows,
                                                               const float* __restrict__ X, float* __restrict__ Y, FeatLayout L,
                                                               int cout, int relu, int tstride) {   // tstride > 0: Y is [column][tstride channels]
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if constexpr (PREC == 4) gatsspg::fp16_saturate_mode();
    constexpr int CPS = CIN / BK;            // K slabs per tap
    constexpr int KT = TAPS * CPS;
    constexpr int HALF = T::BN / 2;
    static_assert(KT % 2 == 0, "the main loop consumes slabs in pairs");
    static_assert(!PATCH || TAPS == 9, "patch tiling is for the 3x3 convolutions");
    const int SEG = (L.W + HALF - 1) / HALF, HP = (L.H + 1) / 2;   // H is odd at 1/8 resolution when H/8 is
    const int MT = (cout + T::BM - 1) / T::BM;
    const int NT = PATCH ? L.b * HP * SEG : L.ldt / T::BN;
    // XCD-aware tile map (workgroup g runs on XCD g % 8): every XCD owns one contiguous band of column tiles, i.e. a
    // band of image rows, and walks it top to bottom with the row tiles of a column tile back to back.  The dy = +-1 taps
    // of a tile read the rows of the tiles one image row above / below; with bands those are in the same XCD's L2
    // (a round-robin map puts them on other XCDs and every input row is fetched into three L2s).
    const int per = (NT + 7) / 8;
    const int slot = blockIdx.x >> 3;
    const int rt = slot % MT;
    const int ct = (blockIdx.x & 7) * per + slot / MT;
    if (slot / MT >= per || ct >= NT) return;
    const int ldt = L.ldt, Wp = L.Wp;
    int c0, im = 0, sx = 0, yp = 0;
    if constexpr (PATCH) {
        im = ct / (HP * SEG);
        const int r = ct - im * (HP * SEG);
        yp = r / SEG;
        sx = r - yp * SEG;
        c0 = im * L.ld + (1 + 2 * yp) * Wp + 1 + HALF * sx;
    } else {
        c0 = ct * T::BN;
    }
    const float* A = Wt + (size_t)rt * T::BM * (TAPS * CIN);
    f32x16 acc[T::TM][T::TN];
    gatsspg::zero_acc(acc);
    auto al = [&](int kt) { return A + kt * BK; };
    auto bl = [&](int kt) {
        const int tap = kt / CPS, cc = kt - tap * CPS;
        const int shift = TAPS == 9 ? (tap / 3 - 1) * Wp + (tap % 3 - 1) : 0;
        return X + ((ptrdiff_t)cc * BK * ldt + c0 + shift);
    };
    if constexpr (PREC == 4) {
        // fp16 planes of the weights, slab-major [K/32][rows][32]: hi plane, then lo plane
        const unsigned short* Ph = Wp16 + (size_t)rt * T::BM * BK;
        const unsigned short* Pl = Ph + (size_t)rows * (TAPS * CIN);
        auto ah = [&](int kt) { return Ph + (size_t)kt * rows * BK; };
        auto alo = [&](int kt) { return Pl + (size_t)kt * rows * BK; };
        unsigned short* sm16 = reinterpret_cast<unsigned short*>(smem);
        if constexpr (PATCH)
            gatsspg::gemm_mainloop_bf3<T, decltype(ah), decltype(alo), decltype(bl), gatsspg::NoHooks, true, 4, PatchCol>(
                acc, sm16, KT, ah, alo, BK, bl, ldt, nullptr, PatchCol{HALF, Wp});
        else
            gatsspg::gemm_mainloop_bf3<T, decltype(ah), decltype(alo), decltype(bl), gatsspg::NoHooks, true, 4>(acc, sm16, KT, ah, alo, BK, bl, ldt);
    } else {
        if constexpr (PATCH) gatsspg::gemm_mainloop<T, decltype(al), decltype(bl), 0, PatchCol>(acc, smem, KT, al, TAPS * CIN, bl, ldt, PatchCol{HALF, Wp});
        else gatsspg::gemm_mainloop<T>(acc, smem, KT, al, TAPS * CIN, bl, ldt);
    }

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / T::WN, wn = wave % T::WN, half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int tn = 0; tn < T::TN; ++tn) {
        const int tc = (wn * T::TN + tn) * 32 + l31;     // column inside the tile
        int col;
        bool ok, store;
        if constexpr (PATCH) {
            const int prow = tc / HALF, px = tc - prow * HALF;
            col = c0 + prow * Wp + px;
            ok = store = HALF * sx + px < L.W && 2 * yp + prow < L.H;
        } else {
            col = c0 + tc;
            int y, x;
            ok = feat_valid(L, col % L.ld, y, x);
            store = true;
        }
        if (tstride > 0) {
            // position-major output (the dense descriptors): a lane's registers 4 g .. 4 g + 3 are four consecutive channels
            // of its column -> one 16-byte store each (cout is a multiple of 4 rows wherever this is used)
#pragma unroll
            for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int row0 = rt * T::BM + (wm * T::TM + tm) * 32 + 8 * g + 4 * half;
                    if (row0 < cout && store) {
                        const float4 bs = *reinterpret_cast<const float4*>(bias + row0);
                        float4 v = {acc[tm][tn][4 * g] + bs.x, acc[tm][tn][4 * g + 1] + bs.y, acc[tm][tn][4 * g + 2] + bs.z, acc[tm][tn][4 * g + 3] + bs.w};
                        if (relu) v = {fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
                        if (!ok) v = {0.f, 0.f, 0.f, 0.f};
                        *reinterpret_cast<float4*>(Y + (size_t)col * tstride + row0) = v;
                    }
                }
            continue;
        }
#pragma unroll
        for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rt * T::BM + (wm * T::TM + tm) * 32 + mfma_row(r, half);
                if (row < cout && store) {
                    float v = acc[tm][tn][r] + bias[row];
                    if (relu) v = fmaxf(v, 0.f);
                    Y[(size_t)row * ldt + col] = ok ? v : 0.f;
                }
            }
    }
}

// =====================================================================================================
// 3x3 convolution + bias + ReLU + MaxPool2d(2, 2) (:143-151: conv1b, conv2b, conv3b feed only the pool).
// The 128 GEMM "columns" of a workgroup are a 2-row x 64-column image patch (the column map of the main loop sends
// column c to row c / 64, x = c % 64), so the 2x2 maxima are workgroup-local: the full-resolution activation is never
// written (conv1b: 67.7 MB less HBM traffic, one launch less per resolution).  Y2 is the half-resolution padded plane;
// only its pixels are written (its pad ring is zeroed once per forward by conv1a_kernel's extra workgroups).
// =====================================================================================================
template <class T, int CIN, int PREC = 0>
__global__ __launch_bounds__(T::THREADS) void conv_pool_kernel(const float* __restrict__ Wt, const float* __restrict__ bias,
                                                               const unsigned short* __restrict__ Wp16, int rows,
                                                               const float* __restrict__ X, float* __restrict__ Y2, FeatLayout L,
                                                               FeatLayout L2, int cout) {
    static_assert(T::BN == 128 && T::BM == 64, "2 x 64 pixel patch, 64 output channels per workgroup");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if constexpr (PREC == 4) gatsspg::fp16_saturate_mode();
    constexpr int CPS = CIN / BK, KT = 9 * CPS;
    static_assert(KT % 2 == 0, "the main loop consumes slabs in pairs");
    const int SEG = (L.W + 63) / 64, HP = L.H / 2;
    const int MT = cout / T::BM, NT = L.b * HP * SEG;
    const int per = (NT + 7) / 8;                      // XCD bands of consecutive patches (see conv_gemm_kernel)
    const int slot = blockIdx.x >> 3;
    const int rt = slot % MT;
    const int t = (blockIdx.x & 7) * per + slot / MT;
    if (slot / MT >= per || t >= NT) return;
    const int im = t / (HP * SEG), r = t - im * (HP * SEG);
    const int yp = r / SEG, sx = r - yp * SEG;         // pooled row, 64-pixel segment
    const int ldt = L.ldt, Wp = L.Wp;
    const int c0 = im * L.ld + (1 + 2 * yp) * Wp + 1 + 64 * sx;
    const float* A = Wt + (size_t)rt * T::BM * (9 * CIN);
    f32x16 acc[T::TM][T::TN];
    gatsspg::zero_acc(acc);
    auto al = [&](int kt) { return A + kt * BK; };
    auto bl = [&](int kt) {
        const int tap = kt / CPS, cc = kt - tap * CPS;
        return X + ((ptrdiff_t)cc * BK * ldt + c0 + (tap / 3 - 1) * Wp + (tap % 3 - 1));
    };
    if constexpr (PREC == 4) {
        const unsigned short* Ph = Wp16 + (size_t)rt * T::BM * BK;
        const unsigned short* Pl = Ph + (size_t)rows * (9 * CIN);
        auto ah = [&](int kt) { return Ph + (size_t)kt * rows * BK; };
        auto alo = [&](int kt) { return Pl + (size_t)kt * rows * BK; };
        gatsspg::gemm_mainloop_bf3<T, decltype(ah), decltype(alo), decltype(bl), gatsspg::NoHooks, true, 4, PatchCol>(
            acc, reinterpret_cast<unsigned short*>(smem), KT, ah, alo, BK, bl, ldt, nullptr, PatchCol{64, Wp});
    } else {
        gatsspg::gemm_mainloop<T, decltype(al), decltype(bl), 0, PatchCol>(acc, smem, KT, al, 9 * CIN, bl, ldt, PatchCol{64, Wp});
    }

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / T::WN, wn = wave % T::WN, half = lane >> 5, l31 = lane & 31;
    constexpr int TS = T::BN + 4;                      // [64][132] staging tile in the (now free) operand buffers
    static_assert(T::BM * TS <= T::SMEM_FLOATS, "staging tile must fit the operand buffers");
    __syncthreads();
#pragma unroll
    for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < T::TN; ++tn)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int row = (wm * T::TM + tm) * 32 + mfma_row(rr, half);
                const int col = (wn * T::TN + tn) * 32 + l31;
                smem[row * TS + col] = fmaxf(acc[tm][tn][rr] + bias[rt * T::BM + row], 0.f);
            }
    __syncthreads();
    const int W2 = L.W / 2;
    float* dst = Y2 + (size_t)im * L2.ld + (size_t)(yp + 1) * L2.Wp + 1 + 32 * sx;
    for (int p = tid; p < T::BM * 32; p += T::THREADS) {
        const int ch = p >> 5, px = p & 31;
        if (32 * sx + px < W2) {
            const float* s0 = smem + ch * TS + 2 * px;
            dst[(size_t)(rt * T::BM + ch) * L2.ldt + px] = fmaxf(fmaxf(s0[0], s0[1]), fmaxf(s0[64], s0[65]));
        }
    }
}

// =====================================================================================================
// conv1a + ReLU + conv1b + ReLU + MaxPool2d(2, 2) in ONE launch (split-fp16 arithmetic, SPP_FLAG_PREC_FP16X4; :142-145).
// conv1b's B operand for a 2-row x 64-pixel output patch is the 64-channel conv1a activation on 4 rows x 66 pixels.  Instead of
// reading it from HBM (conv1a_kernel writes 67.7 MB, the nine shifted views of conv1b re-read it: 590 MB of L2 -> L1 traffic,
// which is what bounds the split-fp16 conv1b), the workgroup RECOMPUTES it from the 6 x 68 image pixels behind it -- on the matrix
// pipe: [32 channels] x [16 k] . [16 k] x [32 positions] with k = the nine taps, the bias (against a column of ones) and zeros,
// weights and pixels as two fp16 terms each, all four products, like conv1b itself -- applies ReLU, splits it into fp16 hi / lo
// terms ONCE (the shifted-view form splits every element once per tap) and keeps the 32-channel slab resident in LDS in the
// B-fragment image ([position][32 k], 80-byte rows).  A tap (dy, dx) is then an offset of dy * 66 + dx positions into that image;
// only the weight planes (8 KB per tap, the same for every patch) stream through the A buffer, three taps (one stencil row) at a time.
//
// Schedule: persistent 512-thread workgroups, two per CU (78.9 KB of LDS each), each walking its XCD band of patches; a patch is
// 12 barrier intervals, alternately STAGING and MULTIPLYING: [block 0] [24 MFMAs per wave] [A slabs 1] [MFMAs] [A slabs 2] [MFMAs]
// [A slabs 3 + block 1] [MFMAs] [A slabs 4] [MFMAs] [A slabs 5, next pixels] [MFMAs] and, without a barrier, [A slabs 0 of the next
// patch, epilogue from registers].  The accumulator chain of a 32x32 tile is serial, so a workgroup alone cannot overlap its
// staging with its MFMAs; the CU does it with the other workgroup.  Measured with per-interval s_memtime stamps
// (tools/conv1ab_probe.hip, profiles/r03k_*): a patch takes ~31 k cycles per workgroup = 18.4 k cycles of MFMA per SIMD for the two
// workgroups' patches (59 % busy, shader clock 1.9 GHz under this load); multiply intervals 1750 + ~650 cycles at their barrier,
// plain staging ~800, the block builds 2700-4300 (wave 0 builds two tiles), the epilogue ~2500.  Tried on the way, each measured
// on the GPU: one tap per barrier, double-buffered (83 us); three fragment sets in flight (spills: a scratch reload on the way into
// the epilogue or a block build is a memory round trip on the critical path); a start delay for the second wave of workgroups, to
// put the two residents of a CU out of phase (`phase_delay`, no gain at any value: kept as a tuning knob at 0); ONE 1024-thread
// workgroup whose halves run one barrier interval apart (89 us: every long staging interval stalls the other half at the shared
// barrier); conv1a on the vector ALU (432 FMAs per thread and patch: as many VALU cycles as the MFMAs take, 82 us).
// The epilogue needs no LDS and no barrier: a wave's 32 columns are 16 pixels of BOTH patch rows, so the 2x2 maximum is two lane
// swizzles (the pixel pair: lane ^ 4, the rows: lane ^ 16), and the pooled values go to HBM from registers.
// =====================================================================================================
constexpr int C1_PHASE_DELAY = 0;                     // x 64 cycles of start delay for the second wave of workgroups (measured: no gain)
constexpr int C1_BW = 66, C1_COLS = 4 * C1_BW;        // resident block: 4 rows x 66 positions
constexpr int C1_KS = 40;                             // 16-bit elements per LDS row (32 + 8 pad: conflict-free 16-byte fragment reads)
constexpr int C1_A_PLANE = 64 * C1_KS, C1_A_STAGE = 2 * C1_A_PLANE;   // hi + lo planes of one tap's 64 x 32 weight slab
constexpr int C1_B_PLANE = C1_COLS * C1_KS;
constexpr int C1_HALF_U16 = 3 * C1_A_STAGE + 2 * C1_B_PLANE + 2 * 6 * 68;   // A buffer (3 taps), block hi / lo, fp32 pixel patch
constexpr int C1_SHARED_BYTES = 2 * 2 * 64 * 16 + 128 * 4;                  // conv1a's A fragments [2 slabs][hi | lo][64 lanes], conv1b's bias
constexpr size_t C1_SMEM_BYTES = 2 * (size_t)C1_HALF_U16 + C1_SHARED_BYTES;
static_assert(2 * C1_SMEM_BYTES <= 160 * 1024 && (2 * C1_HALF_U16) % 16 == 0, "two workgroups per CU");

// C1A: the first layer (input = the image, conv1a recomputed into the resident block); otherwise the block is LOADED from the
// 64-channel padded plane X of the same resolution (its zero pad ring is the halo) and split once -- conv2a, conv2b, conv3a: the
// generic loop splits every activation once per tap and row tile (9 - 18 times).  POOL: 2x2 maximum into the half-resolution plane
// (layout LY), otherwise bias + ReLU into the full-resolution plane.  rows = output channels (64 per workgroup item, rows / 64 items per patch).
template <bool C1A, bool POOL>
__global__ __launch_bounds__(512, 4) void conv1ab_pool_f16_kernel(const float* __restrict__ img, const float* __restrict__ w1a,
                                                               const float* __restrict__ b1a, const unsigned short* __restrict__ Wp16,
                                                               const float* __restrict__ bias, float* __restrict__ Y2, FeatLayout L,
                                                               FeatLayout L2, int rows, PadPlanes pp, int phase_delay, int abl   // abl: timing ablations (tuning builds), 0 in the product
#ifdef C1_PROBE
                                                               ,
                                                               unsigned long long* __restrict__ probe   // tools/conv1ab_probe.hip
#endif
) {
#ifdef C1_PROBE
    int probe_n = 0;
#define C1_STAMP()                                                                                                   \
    do {                                                                                                             \
        if (blockIdx.x == 8 && (threadIdx.x & 511) == 0 && probe_n < 64)                                             \
            probe[(threadIdx.x >> 9) * 64 + probe_n++] = __builtin_readcyclecounter();                               \
    } while (0)
#else
#define C1_STAMP() ((void)0)
#endif
    using gatsspg::u32x4;
    using gatsspg::bf16x8;
    using gatsspg::f16x8;
#ifdef C1_PROBE
    if (blockIdx.x == 8 && threadIdx.x == 0) {
        probe[128] = __builtin_readcyclecounter();
        probe[130] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    extern __shared__ __attribute__((aligned(16))) float smem_all[];
    gatsspg::fp16_saturate_mode();
    unsigned short* const sm16 = reinterpret_cast<unsigned short*>(smem_all);
    unsigned short* const Abuf = sm16;                 // [3 taps][hi | lo][64][40]
    unsigned short* const Bhi = sm16 + 3 * C1_A_STAGE;
    unsigned short* const Blo = Bhi + C1_B_PLANE;
    float* const patch = reinterpret_cast<float*>(Blo + C1_B_PLANE);   // [6][68] image pixels, zeros outside the image
    gatsspg::u32x4* const wfrag = reinterpret_cast<gatsspg::u32x4*>(sm16 + C1_HALF_U16);
    float* const bias_s = reinterpret_cast<float*>(wfrag + 2 * 2 * 64);

    const int SEG = (L.W + 63) / 64, HP = L.H / 2;
    const int MT = rows >> 6;                          // 64-channel row tiles: items (patch, row tile), the row tiles of a patch back to back
    const int NT = L.b * HP * SEG * MT;
    const int per = (NT + 7) / 8;                      // XCD bands of consecutive patches (see conv_gemm_kernel)
    const int nslots = gridDim.x >> 3;                 // workgroups per XCD; each walks its band with that stride
    const int xcd = blockIdx.x & 7;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3, half = lane >> 5, l31 = lane & 31;

    // ---- weight planes of conv1b: slab-major [K/32][64][32], k = tap * 64 + ci -> slab of (tap, channel slab cc) = tap * 2 + cc;
    //      the K loop runs cc-major (the resident block changes once per channel slab)
    constexpr int K1B = 9 * 64;
    const int a_plane = __builtin_amdgcn_readfirstlane(tid >> 8);          // waves 0-3: hi plane, 4-7: lo plane
    const int a_r = (tid & 255) >> 2, a_c8 = tid & 3;
    const unsigned short* a_src = Wp16 + (size_t)a_plane * rows * K1B + (size_t)a_r * 32 + a_c8 * 8;
    const int a_soff = a_plane * C1_A_PLANE + a_r * C1_KS + a_c8 * 8;
    // group grp = cc * 3 + (dy + 1): the three taps dx = -1, 0, 1 of one stencil row; rt = row tile of the item
    auto gload_a = [&](int grp, int rt, u32x4 (&ra)[3]) {
        const int cc = grp / 3, dyi = grp - cc * 3;
#pragma unroll
        for (int d = 0; d < 3; ++d)
            ra[d] = *reinterpret_cast<const u32x4*>(a_src + ((size_t)((dyi * 3 + d) * 2 + cc) * rows + rt * 64) * 32);
    };
    auto swrite_a = [&](const u32x4 (&ra)[3]) {
#pragma unroll
        for (int d = 0; d < 3; ++d) *reinterpret_cast<u32x4*>(Abuf + d * C1_A_STAGE + a_soff) = ra[d];
    };

    // ---- image patch of tile t: rows 2 yp - 2 .. 2 yp + 3, columns 64 sx - 2 .. 64 sx + 65 (one pixel per thread, tid < 408)
    const int p_ri = tid / 68, p_xi = tid - p_ri * 68;
    auto gload_patch = [&](int t, bool valid) -> float {
        const int im = t / (HP * SEG), r0 = t - im * (HP * SEG);
        const int yp = r0 / SEG, sx = r0 - yp * SEG;
        const int Y = 2 * yp - 2 + p_ri, X = 64 * sx - 2 + p_xi;
        return (valid && tid < 6 * 68 && Y >= 0 && Y < L.H && X >= 0 && X < L.W) ? img[((size_t)im * L.H + Y) * L.W + X] : 0.f;
    };

    // ---- conv1a on the matrix pipe.  A fragments per lane: row = channel l31 of the slab, k = 8 half .. + 7
    auto load_w1a = [&](int cc, u32x4& whi, u32x4& wlo) {
        const int ch = cc * 32 + l31;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = half ? 0.f : w1a[ch * 9 + i];
        if (half) {
            v[0] = w1a[ch * 9 + 8];
            v[1] = b1a[ch];
        }
        unsigned hi[4], lo[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) gatsspg::fp16_split2(v[2 * i], v[2 * i + 1], hi[i], lo[i]);
        whi = (u32x4){hi[0], hi[1], hi[2], hi[3]};
        wlo = (u32x4){lo[0], lo[1], lo[2], lo[3]};
    };
    auto h8u = [](u32x4 v) { return __builtin_bit_cast(f16x8, v); };
    // N 32-position tiles of the resident block at once (their dependency chains -- LDS, split, 4 MFMAs, split, LDS -- interleave):
    // positions j = 32 tt + l31 -> block row j / 66 (image row 2 yp - 1 + r), x = j % 66
    auto block_tiles = [&](auto ntag, const int (&tts)[2], int cc, int yp, int sx) {
        constexpr int N = decltype(ntag)::value;
        const u32x4 whi = wfrag[(cc * 2 + 0) * 64 + lane], wlo = wfrag[(cc * 2 + 1) * 64 + lane];
        u32x4 phi[N], plo[N];
        int jj[N];
#pragma unroll
        for (int n = 0; n < N; ++n) {
            int j = 32 * tts[n] + l31;
            asm volatile("" : "+v"(j));                        // recompute the addresses here instead of keeping them alive across the K loop
            jj[n] = j;
            const int r = j / C1_BW, x = j - r * C1_BW;
            const int Y = 2 * yp - 1 + r, X = 64 * sx - 1 + x;
            const bool inside = j < C1_COLS && Y >= 0 && Y < L.H && X >= 0 && X < L.W;
            const float* pp = patch + min(r, 3) * 68 + x;      // the 3 x 3 window's top-left pixel
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = pp[half ? (2 * 68 + 2) : ((i / 3) * 68 + i % 3)];
            if (half) {
                v[1] = 1.f;
#pragma unroll
                for (int i = 2; i < 8; ++i) v[i] = 0.f;
            }
            unsigned hi[4], lo[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                gatsspg::fp16_split2(v[2 * i], v[2 * i + 1], hi[i], lo[i]);
                hi[i] = inside ? hi[i] : 0u;                   // conv1a_kernel: zeros at the pad positions of its plane (bias column included)
                lo[i] = inside ? lo[i] : 0u;
            }
            phi[n] = (u32x4){hi[0], hi[1], hi[2], hi[3]};
            plo[n] = (u32x4){lo[0], lo[1], lo[2], lo[3]};
        }
        f32x16 c[N];
#pragma unroll
        for (int n = 0; n < N; ++n)
#pragma unroll
            for (int q = 0; q < 16; ++q) c[n][q] = 0.f;
#pragma unroll
        for (int n = 0; n < N; ++n) c[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8u(wlo), h8u(plo[n]), c[n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < N; ++n) c[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8u(wlo), h8u(phi[n]), c[n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < N; ++n) c[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8u(whi), h8u(plo[n]), c[n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < N; ++n) c[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8u(whi), h8u(phi[n]), c[n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < N; ++n) {
            if (jj[n] < C1_COLS) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {                  // channels 8 g + 4 half .. + 3 of the slab = c[4 g .. 4 g + 3]
                    unsigned h0, l0, h1, l1;
                    gatsspg::fp16_split2(fmaxf(c[n][4 * g], 0.f), fmaxf(c[n][4 * g + 1], 0.f), h0, l0);
                    gatsspg::fp16_split2(fmaxf(c[n][4 * g + 2], 0.f), fmaxf(c[n][4 * g + 3], 0.f), h1, l1);
                    const int o = jj[n] * C1_KS + 8 * g + 4 * half;
                    *reinterpret_cast<uint2*>(Bhi + o) = make_uint2(h0, h1);
                    *reinterpret_cast<uint2*>(Blo + o) = make_uint2(l0, l1);
                }
            }
        }
    };
    // the loaded form: position j of the block is X[ch][im][(2 yp + r) * Wp + 64 sx + x] (padded coordinates: the plane's zero ring is
    // the halo); a lane owns 16 channels of its position: 16 loads (each coalesced over the wave's positions), 8 splits, 2 + 2 16-byte stores
    // two phases, so that the loads of channel slab 1 can be in flight while slab 0 is multiplied
    auto load_issue = [&](int tt, int cc, int im, int yp, int sx, float (&v)[16]) {
        int j = 32 * tt + l31;
        asm volatile("" : "+v"(j));
        j = min(j, C1_COLS - 1);                           // lanes past the block re-read its last position (not stored)
        const int r = j / C1_BW, x = j - r * C1_BW;
        const float* src = img + (size_t)im * L.ld + (size_t)(2 * yp + r) * L.Wp + 64 * sx + x + (size_t)(cc * 32 + 16 * half) * L.ldt;
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = src[(size_t)i * L.ldt];
    };
    auto load_commit = [&](int tt, const float (&v)[16]) {
        int j = 32 * tt + l31;
        asm volatile("" : "+v"(j));
        if (j >= C1_COLS) return;
        unsigned hi[8], lo[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) gatsspg::fp16_split2(v[2 * i], v[2 * i + 1], hi[i], lo[i]);
        const int o = j * C1_KS + 16 * half;
        *reinterpret_cast<u32x4*>(Bhi + o) = (u32x4){hi[0], hi[1], hi[2], hi[3]};
        *reinterpret_cast<u32x4*>(Bhi + o + 8) = (u32x4){hi[4], hi[5], hi[6], hi[7]};
        *reinterpret_cast<u32x4*>(Blo + o) = (u32x4){lo[0], lo[1], lo[2], lo[3]};
        *reinterpret_cast<u32x4*>(Blo + o + 8) = (u32x4){lo[4], lo[5], lo[6], lo[7]};
    };
    // the eight positions past the eighth tile (block row 3, x = 58 .. 65) x 32 channels: one value per thread of waves 0-3
    auto tail_issue = [&](int cc, int im, int yp, int sx) -> float {
        if (tid >= 256) return 0.f;
        return img[(size_t)im * L.ld + (size_t)(2 * yp + 3) * L.Wp + 64 * sx + 58 + (tid & 7) + (size_t)(cc * 32 + (tid >> 3)) * L.ldt];
    };
    auto tail_commit = [&](float v) {
        if (tid >= 256) return;
        unsigned h, l;
        gatsspg::fp16_split2(v, 0.f, h, l);
        const int o = (256 + (tid & 7)) * C1_KS + (tid >> 3);
        Bhi[o] = (unsigned short)h;
        Blo[o] = (unsigned short)l;
    };
    // 264 positions = 8 tiles of 32 + 8: wave w builds tile w, wave 0 also the tail
    auto make_block = [&](int cc, int im, int yp, int sx) {
        C1_STAMP();
        if constexpr (!C1A) {
            float v[16];
            load_issue(wave, cc, im, yp, sx, v);
            const float tv = tail_issue(cc, im, yp, sx);
            load_commit(wave, v);
            tail_commit(tv);
            C1_STAMP();
            return;
        }
        if (wave == 0) {
            const int tts[2] = {0, 8};
            block_tiles(std::integral_constant<int, 2>{}, tts, cc, yp, sx);
        } else {
            const int tts[2] = {wave, wave};
            block_tiles(std::integral_constant<int, 1>{}, tts, cc, yp, sx);
        }
        C1_STAMP();
    };

    const int arow = (wm * 32 + l31) * C1_KS;
    // a wave's 32 columns: pixels 16 wn .. 16 wn + 15 of patch row l31 / 16 (the 2x2 pooling window stays inside the wave).  Lanes
    // 4-11 of a 16-lane row take the EVEN pixels, lanes 0-3 and 12-15 the odd ones: a 16-byte LDS read is served in the lane groups
    // {0-3, 12-15, 20-27} and {4-11, 16-19, 28-31}, i.e. eight lanes of patch row 0 and eight of row 1, 66 = 2 (mod 16) positions
    // apart; with odd pixels in one row and even ones in the other the sixteen 80-byte rows of a group sit on sixteen different bank
    // quads (pixels in lane order: two groups of eight collide on two of them -- 8.4 M conflict cycles of 23 M LDS cycles, PMC)
    const int m16 = l31 & 15;
    const int px16 = (m16 >= 4 && m16 < 12) ? 2 * (m16 - 4) : (m16 < 4 ? 2 * m16 + 1 : 2 * (m16 - 12) + 9);
    const int jbase = ((l31 >> 4) + 1) * C1_BW + 16 * wn + px16 + 1;         // the column's position in the block for the centre tap
    f32x16 acc;
    // fragment sets (one 16-deep k step of a tap: A hi / lo, B hi / lo) are requested one to two steps ahead of the four MFMAs that
    // consume them (two sets: a third one spills -- and a scratch reload on the way into the epilogue costs a memory round trip)
    struct Frags {
        bf16x8 ah, al, bh, bl;
    };
    auto lfrag = [&](Frags& f, int dyi, int hs) {          // hs = 2 * (dx + 1) + k step
        const int d = hs >> 1, ko = 16 * (hs & 1) + 8 * half;
        const unsigned short* A = Abuf + d * C1_A_STAGE + arow + ko;
        const int jb = (jbase + (dyi - 1) * C1_BW + (d - 1)) * C1_KS + ko;
        f.ah = *reinterpret_cast<const bf16x8*>(A);
        f.al = *reinterpret_cast<const bf16x8*>(A + C1_A_PLANE);
        f.bh = *reinterpret_cast<const bf16x8*>(Bhi + jb);
        f.bl = *reinterpret_cast<const bf16x8*>(Blo + jb);
    };
    auto mm = [&](const Frags& f) {
        auto h8 = [](bf16x8 v) { return __builtin_bit_cast(f16x8, v); };
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(f.al), h8(f.bl), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(f.al), h8(f.bh), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(f.ah), h8(f.bl), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(f.ah), h8(f.bh), acc, 0, 0, 0);
    };
    auto compute3 = [&](int dyi) {
        Frags f0, f1;
        lfrag(f0, dyi, 0);
        lfrag(f1, dyi, 1);
#pragma unroll
        for (int hs = 0; hs < 6; hs += 2) {
            __builtin_amdgcn_sched_barrier(0);
            mm(f0);
            __builtin_amdgcn_sched_barrier(0);
            if (hs + 2 < 6) lfrag(f0, dyi, hs + 2);
            __builtin_amdgcn_sched_barrier(0);
            mm(f1);
            __builtin_amdgcn_sched_barrier(0);
            if (hs + 3 < 6) lfrag(f1, dyi, hs + 3);
        }
    };

    // ---- epilogue of a finished patch, from registers: bias + ReLU, 2x2 max (pixel pairs = neighbouring lanes, rows = lanes 16 apart)
    auto epilogue = [&](int item) {
        if (abl & 8) {
            if (acc[0] == 12345.f) Y2[0] = acc[1];
            return;
        }
        const int t = item / MT, rt = item - t * MT;
        const int im = t / (HP * SEG), r0 = t - im * (HP * SEG);
        const int yp = r0 / SEG, sx = r0 - yp * SEG;
        if constexpr (!POOL) {
            // bias + ReLU into the full-resolution plane: lane -> (patch row l31 / 16, pixel 16 wn + px16), a register per channel
            int col = (2 * yp + (l31 >> 4) + 1) * L2.Wp + 64 * sx + 16 * wn + px16 + 1;
            asm volatile("" : "+v"(col));
            const bool inimg = 64 * sx + 16 * wn + px16 < L.W;
            float* dstc = Y2 + (size_t)im * L2.ld + col;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                int row0 = rt * 64 + wm * 32 + 8 * g + 4 * half;
                asm volatile("" : "+v"(row0));
                const float4 bs = *reinterpret_cast<const float4*>(bias_s + row0);
                const float bb[4] = {bs.x, bs.y, bs.z, bs.w};
                float* drow = dstc + (size_t)row0 * L2.ldt;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (inimg) drow[(size_t)q * L2.ldt] = fmaxf(acc[4 * g + q] + bb[q], 0.f);
            }
            return;
        }
        int px2 = 8 * wn + ((l31 & 15) - 4);               // pooled pixel of an even-pixel lane (4-11) within the 32-pixel pooled segment
        asm volatile("" : "+v"(px2));                      // (addresses recomputed here, not kept alive across the K loop)
        float* dst = Y2 + (size_t)im * L2.ld + (size_t)(yp + 1) * L2.Wp + 1 + 32 * sx + px2;
        const bool writer = l31 >= 4 && l31 < 12 && 32 * sx + px2 < L.W / 2;
        float hmax[16];
        int other[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bs = *reinterpret_cast<const float4*>(bias_s + rt * 64 + wm * 32 + 8 * g + 4 * half);   // rows mfma_row(4 g .. 4 g + 3, half)
            const float bb[4] = {bs.x, bs.y, bs.z, bs.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float v = fmaxf(acc[4 * g + q] + bb[q], 0.f);
                hmax[4 * g + q] = v;
            }
        }
        // pixel 2 j sits in lane 4 + j, pixel 2 j + 1 in lane j (j < 4) or 8 + j: lane ^ 4; the other patch row: lane ^ 16
#pragma unroll
        for (int r = 0; r < 16; ++r) other[r] = __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, hmax[r]), 0x101F);
#pragma unroll
        for (int r = 0; r < 16; ++r) hmax[r] = fmaxf(hmax[r], __builtin_bit_cast(float, other[r]));
#pragma unroll
        for (int r = 0; r < 16; ++r) other[r] = __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, hmax[r]), 0x401F);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            int row0 = rt * 64 + wm * 32 + 8 * g + 4 * half;
            asm volatile("" : "+v"(row0));
            float* drow = dst + (size_t)row0 * L2.ldt;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (writer) drow[(size_t)q * L2.ldt] = fmaxf(hmax[4 * g + q], __builtin_bit_cast(float, other[4 * g + q]));
        }
    };

    // the pad rings of the patch-tiled planes (conv1a_kernel carries this job in its spare workgroups when it runs): none of them is
    // read here, and this kernel's own output plane gets pixels only.  Done after the patches (at the start it delays every
    // workgroup's first patch: +4.6 us measured, against 5.8 us for a launch of its own).
    auto zero_rings = [&]() {
        if constexpr (C1A) {
#pragma unroll
            for (int i = 0; i < PadPlanes::N; ++i) zero_pad_ring(pp.p[i], pp.L[i], pp.C[i], blockIdx.x * 512 + threadIdx.x, gridDim.x * 512);
        }
    };
    int slot = blockIdx.x >> 3;
    if (slot >= per || xcd * per + slot >= NT) {
        zero_rings();
        return;
    }
    u32x4 ra[3];
    int item = xcd * per + slot;                           // (patch, row tile)
    gload_a(0, item % MT, ra);
    if constexpr (C1A) {
        if (tid < 6 * 68) patch[tid] = gload_patch(item, true);
        if (wave < 2) {                                    // conv1a's A fragments of both channel slabs
            u32x4 whi, wlo;
            load_w1a(wave, whi, wlo);
            wfrag[(wave * 2 + 0) * 64 + lane] = whi;
            wfrag[(wave * 2 + 1) * 64 + lane] = wlo;
        }
    }
    if (wave >= 2 && wave < 2 + (rows >> 6)) bias_s[(wave - 2) * 64 + lane] = bias[(wave - 2) * 64 + lane];
    if (2 * slot >= nslots)                                // the workgroups that land beside the first wave of the grid (tuning knob, 0)
        for (int i = 0; i < phase_delay; ++i) __builtin_amdgcn_s_sleep(1);
    swrite_a(ra);                                          // group 0
    __syncthreads();
    // The A slab registers are written to the buffer FIRST in every staging interval and re-requested LAST, so that they are not
    // live across the block builds and the epilogue (the kernel sits at the 128-register limit of 4 waves per SIMD).
    for (;;) {
        const int t = item / MT, rt = item - t * MT;
        const int im = t / (HP * SEG), r0 = t - im * (HP * SEG);
        const int yp = r0 / SEG, sx = r0 - yp * SEG;       // pooled row, 64-pixel segment
        const int nslot = slot + nslots;
        const bool more = nslot < per && xcd * per + nslot < NT;
        const int nitem = xcd * per + nslot;
        float pix = 0.f;
        if constexpr (C1A) pix = gload_patch(nitem, more); // the next patch's pixel: in flight until the staging interval after group 4
        // interval 0: resident block of channel slab 0 (the A slabs of group 0 are in the buffer)
        if (!(abl & 1)) make_block(0, im, yp, sx);
        float nv[16], nvt = 0.f;                           // loaded form: slab 1 of this wave's tile (+ tail value), requested three groups ahead
        if constexpr (!C1A) {
            load_issue(wave, 1, im, yp, sx, nv);
            nvt = tail_issue(1, im, yp, sx);
        }
        gload_a(1, rt, ra);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        C1_STAMP();
        __syncthreads();
        C1_STAMP();
        // intervals 1..11: group grp = 3 taps x 8 MFMAs from the A buffer | the next group's slabs (in registers since the previous
        // staging interval) are written, the one after is requested (after group 5 comes group 0 of the next item)
        for (int grp = 0; grp < 6; ++grp) {
            if (!(abl & 2)) compute3(grp % 3);
            C1_STAMP();
            __syncthreads();
            C1_STAMP();
            if (grp == 5) break;
            swrite_a(ra);
            asm volatile("" ::: "memory");
            if (grp == 2 && !(abl & 1)) {
                if constexpr (C1A) make_block(1, im, yp, sx);
                else {
                    load_commit(wave, nv);
                    tail_commit(nvt);
                }
            }
            if constexpr (C1A) {
                if (grp == 4 && tid < 6 * 68) patch[tid] = pix;    // block 1 is built: the next patch's pixels
            }
            if (grp == 4) gload_a(0, more ? nitem % MT : 0, ra);
            else gload_a(grp + 2, rt, ra);
            C1_STAMP();
            __syncthreads();
            C1_STAMP();
        }
        if (more) swrite_a(ra);                            // group 0 of the next item
        epilogue(item);                                    // no LDS, no barrier: the next item's interval 0 follows in the same interval
        C1_STAMP();
        if (!more) break;
        slot = nslot;
        item = nitem;
    }
    zero_rings();
#ifdef C1_PROBE
    if (blockIdx.x == 8 && threadIdx.x == 0) {
        probe[129] = __builtin_readcyclecounter();
        probe[131] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// =====================================================================================================
// MaxPool2d(2, 2) (:145,148,151) between padded planes; pad positions of the output are zeros
// =====================================================================================================
__global__ __launch_bounds__(256) void pool_kernel(const float* __restrict__ X, FeatLayout Li, float* __restrict__ Y,
                                                   FeatLayout Lo) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y, im = blockIdx.z;
    if (q >= Lo.ld) return;
    int y, x;
    float v = 0.f;
    if (feat_valid(Lo, q, y, x)) {
        const float* s = X + (size_t)c * Li.ldt + (size_t)im * Li.ld + (size_t)(2 * y - 1) * Li.Wp + (2 * x - 1);
        v = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[Li.Wp], s[Li.Wp + 1]));
    }
    Y[(size_t)c * Lo.ldt + (size_t)im * Lo.ld + q] = v;
}

// dense descriptors out of the padded position-major plane ([position][256]): [b][256][Hc][Wc].  A workgroup transposes
// 64 cells x 64 channels through LDS (reads: 256 B per cell, writes: 256 B per channel).
__global__ __launch_bounds__(256) void export_dense_kernel(const float* __restrict__ X, FeatLayout L, float* __restrict__ out) {
    __shared__ float tile[64][65];
    const int i0 = blockIdx.x * 64, c0 = blockIdx.y * 64, im = blockIdx.z;
    const int HW = L.H * L.W;
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
        const int ci = e >> 6, ch = e & 63, i = i0 + ci;
        if (i < HW) {
            const int y = i / L.W, x = i - y * L.W;
            tile[ci][ch] = X[((size_t)im * L.ld + (size_t)(y + 1) * L.Wp + x + 1) * DD + c0 + ch];
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
        const int ch = e >> 6, ci = e & 63, i = i0 + ci;
        if (i < HW) out[((size_t)im * DD + c0 + ch) * HW + i] = tile[ci][ch];
    }
}

// =====================================================================================================
// weight packing
// =====================================================================================================
// src [cout][cin][k][k] -> dst rows [row0, row0 + cout) of [rows][taps*cin]; rows beyond are zeroed by the caller
__global__ void pack_conv_kernel(const float* __restrict__ src, const float* __restrict__ bsrc, float* __restrict__ dst,
                                 float* __restrict__ bdst, int cout, int cin, int taps, int row0) {
    const int K = taps * cin;
    const size_t n = (size_t)cout * K;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int o = (int)(i / K), k = (int)(i % K);
        const int tap = k / cin, ci = k % cin;
        dst[(size_t)(row0 + o) * K + k] = src[((size_t)o * cin + ci) * taps + tap];
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cout; i += gridDim.x * blockDim.x) bdst[row0 + i] = bsrc[i];
}

// fp16 hi / lo planes of one GEMM convolution's packed weights (slab-major, spp_common.h)
__global__ __launch_bounds__(256) void split_conv_weights_kernel(const float* __restrict__ w, unsigned short* __restrict__ planes, int rows,
                                                                 int K) {
    gatsspg::fp16_saturate_mode();
    const size_t n = (size_t)rows * K;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t m = i / K, k = i % K;
        unsigned hi, lo;
        gatsspg::fp16_split2(w[i], 0.f, hi, lo);
        const size_t d = ((k >> 5) * rows + m) * 32 + (k & 31);
        planes[d] = (unsigned short)(hi & 0xFFFFu);
        planes[n + d] = (unsigned short)(lo & 0xFFFFu);
    }
}

void launch_pack_weights(const void* raw_host, float* packed, hipStream_t s) {
    const spp_raw_weights& r = *static_cast<const spp_raw_weights*>(raw_host);
    (void)hipMemsetAsync(packed, 0, sizeof(float) * PW_TOTAL, s);
    // conv1a: [64][1][3][3] is already [64][9]
    (void)hipMemcpyAsync(packed + PW_C1A_W, r.weight[0], sizeof(float) * 64 * 9, hipMemcpyDeviceToDevice, s);
    (void)hipMemcpyAsync(packed + PW_C1A_B, r.bias[0], sizeof(float) * 64, hipMemcpyDeviceToDevice, s);
    auto pack = [&](int gi, int layer, int row0) {
        const ConvSpec c = kConv[gi];
        const int cout = layer == 9 ? 65 : (gi == 7 ? 256 : c.cout);
        hipLaunchKernelGGL(pack_conv_kernel, dim3(256), dim3(256), 0, s, r.weight[layer], r.bias[layer], packed + conv_w_off(gi),
                           packed + conv_b_off(gi), cout, c.cin, c.taps, row0);
    };
    for (int gi = 0; gi < 7; ++gi) pack(gi, gi + 1, 0);   // conv1b .. conv4b
    pack(7, 8, 0);      // convPa
    pack(7, 10, 256);   // convDa
    pack(8, 9, 0);      // convPb
    pack(9, 11, 0);     // convDb
    unsigned short* planes = reinterpret_cast<unsigned short*>(packed + PW_TOTAL);
    for (int gi = 0; gi < NGEMM; ++gi)
        hipLaunchKernelGGL(split_conv_weights_kernel, dim3(128), dim3(256), 0, s, packed + conv_w_off(gi), planes + conv_wp_off(gi),
                           kConv[gi].rows, (int)conv_k(gi));
}

// =====================================================================================================
// launchers
// =====================================================================================================
// Tuning knobs: the product library never reads the environment; a -DSPP_TUNING build (build_ext --tuning) does.
static const char* tuning_env(const char* name) {
#ifdef SPP_TUNING
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

using Tile64x128 = GemmTile<64, 128, 1, 4, false, true>;
using Tile64x64 = GemmTile<64, 64, 2, 2, false, true>;
using Tile128x64 = GemmTile<128, 64, 2, 2, false, true>;
using Tile64x128w8 = GemmTile<64, 128, 2, 4, false, true>;    // 8 waves, one 32x32 MFMA tile each
using Tile128x64w8 = GemmTile<128, 64, 4, 2, false, true>;

static bool patch_tiling() {
    static const bool on = !(tuning_env("SPP_PATCH") && atoi(tuning_env("SPP_PATCH")) == 0);   // SPP_PATCH=0: flat tiles (A/B timing)
    return on;
}

template <class T, int CIN, int TAPS, int PREC>
static void launch_conv_t(int gi, int kid, const float* packed, const float* X, float* Y, const FeatLayout& L, int relu,
                          hipStream_t s, ProfileHook* hk) {
    const int cout = kConv[gi].cout;
    const int MT = (cout + T::BM - 1) / T::BM;
    const unsigned short* planes = reinterpret_cast<const unsigned short*>(packed + PW_TOTAL) + conv_wp_off(gi);
    if constexpr (TAPS == 9) {
        if (patch_tiling()) {
            auto kern = conv_gemm_kernel<T, CIN, TAPS, true, PREC>;
            const int NT = L.b * ((L.H + 1) / 2) * ((L.W + T::BN / 2 - 1) / (T::BN / 2));
            SPP_LAUNCH(hk, kid, s, kern, dim3(gatsspg::xcd_grid(MT, NT)), dim3(T::THREADS), (smem_bytes<T, PREC>()), s,
                       packed + conv_w_off(gi), packed + conv_b_off(gi), planes, kConv[gi].rows, X, Y, L, cout, relu, 0);
            return;
        }
    }
    auto kern = conv_gemm_kernel<T, CIN, TAPS, false, PREC>;
    const int NT = L.ldt / T::BN;
    static_assert(conv_b_off(NGEMM - 1) % 4 == 0, "the position-major epilogue reads the bias as float4");
    const int tstride = gi == NGEMM - 1 ? DD : 0;      // convDb: the dense descriptors are kept position-major (DescView, spp_common.h)
    SPP_LAUNCH(hk, kid, s, kern, dim3(gatsspg::xcd_grid(MT, NT)), dim3(T::THREADS), (smem_bytes<T, PREC>()), s, packed + conv_w_off(gi),
               packed + conv_b_off(gi), planes, kConv[gi].rows, X, Y, L, cout, relu, tstride);
}

// Tile shape per GEMM convolution: 0 = 64x128, 1 = 64x64, 2 = 128x64 (rows x columns), 3 = 64x128 on 8 waves,
// 4 = 128x64 on 8 waves.  SPP_TILES="a,b,..." (ten
// comma-separated ids, forward order conv1b..convDb) overrides the defaults for tuning; every choice is equally correct.
static const int* conv_tiles() {
    static int tiles[NGEMM] = {0, 0, 0, 1, 1, 1, 1, 1, 1, 1};
    static bool init = false;
    if (!init) {
        init = true;
        if (const char* e = tuning_env("SPP_TILES")) {
            int i = 0;
            for (const char* p = e; *p && i < NGEMM; ++p)
                if (*p >= '0' && *p <= '4') tiles[i++] = *p - '0';
        }
        for (int i = 0; i < NGEMM; ++i)
            if ((tiles[i] == 2 || tiles[i] == 4) && kConv[i].rows % 128) tiles[i] = 1;
    }
    return tiles;
}

template <int CIN, int TAPS>
static void launch_conv(int gi, int kid, const float* packed, const float* X, float* Y, const FeatLayout& L, int relu,
                        hipStream_t s, ProfileHook* hk, int prec) {
    if (prec == 4) {   // split-fp16 loop: a thread owns 4 or 8 k of one column -> the 64x128 tile runs on 8 waves
        switch (conv_tiles()[gi]) {
            case 0: case 3: launch_conv_t<Tile64x128w8, CIN, TAPS, 4>(gi, kid, packed, X, Y, L, relu, s, hk); break;
            case 2: launch_conv_t<Tile128x64, CIN, TAPS, 4>(gi, kid, packed, X, Y, L, relu, s, hk); break;
            case 4: launch_conv_t<Tile128x64w8, CIN, TAPS, 4>(gi, kid, packed, X, Y, L, relu, s, hk); break;
            default: launch_conv_t<Tile64x64, CIN, TAPS, 4>(gi, kid, packed, X, Y, L, relu, s, hk); break;
        }
        return;
    }
    switch (conv_tiles()[gi]) {
        case 0: launch_conv_t<Tile64x128, CIN, TAPS, 0>(gi, kid, packed, X, Y, L, relu, s, hk); break;
        case 2: launch_conv_t<Tile128x64, CIN, TAPS, 0>(gi, kid, packed, X, Y, L, relu, s, hk); break;
        case 3: launch_conv_t<Tile64x128w8, CIN, TAPS, 0>(gi, kid, packed, X, Y, L, relu, s, hk); break;
        case 4: launch_conv_t<Tile128x64w8, CIN, TAPS, 0>(gi, kid, packed, X, Y, L, relu, s, hk); break;
        default: launch_conv_t<Tile64x64, CIN, TAPS, 0>(gi, kid, packed, X, Y, L, relu, s, hk); break;
    }
}

static void launch_pool(const float* X, const FeatLayout& Li, float* Y, const FeatLayout& Lo, int C, hipStream_t s,
                        ProfileHook* hk) {
    SPP_LAUNCH(hk, KID_POOL, s, pool_kernel, dim3((Lo.ld + 255) / 256, C, Lo.b), dim3(256), 0, s, X, Li, Y, Lo);
}

// conv + ReLU + 2x2 pool in one launch (SPP_FUSE_POOL=0 falls back to the two-kernel form for A/B timing)
template <int CIN>
static void launch_conv_pool(int gi, int kid, const float* packed, const float* X, float* tmp, float* Y2, const FeatLayout& L,
                             const FeatLayout& L2, hipStream_t s, ProfileHook* hk, int prec) {
    static const bool fuse = !(tuning_env("SPP_FUSE_POOL") && atoi(tuning_env("SPP_FUSE_POOL")) == 0);
    const int cout = kConv[gi].cout;
    if (!fuse) {
        launch_conv<CIN, 9>(gi, kid, packed, X, tmp, L, 1, s, hk, prec);
        launch_pool(tmp, L, Y2, L2, cout, s, hk);
        return;
    }
    static const bool w8 = tuning_env("SPP_POOL_TILE") && atoi(tuning_env("SPP_POOL_TILE")) == 1;   // 1: the 64x128 patch on 8 waves
    const int NT = L.b * (L.H / 2) * ((L.W + 63) / 64);
    const unsigned short* planes = reinterpret_cast<const unsigned short*>(packed + PW_TOTAL) + conv_wp_off(gi);
    auto go = [&](auto kern, int threads, size_t lds) {
        SPP_LAUNCH(hk, kid, s, kern, dim3(gatsspg::xcd_grid(cout / 64, NT)), dim3(threads), lds, s, packed + conv_w_off(gi),
                   packed + conv_b_off(gi), planes, kConv[gi].rows, X, Y2, L, L2, cout);
    };
    if (prec == 4) go(conv_pool_kernel<Tile64x128w8, CIN, 4>, Tile64x128w8::THREADS, smem_bytes<Tile64x128w8, 4>());
    else if (w8) go(conv_pool_kernel<Tile64x128w8, CIN, 0>, Tile64x128w8::THREADS, smem_bytes<Tile64x128w8>());
    else go(conv_pool_kernel<Tile64x128, CIN, 0>, Tile64x128::THREADS, smem_bytes<Tile64x128>());
}

void launch_dense(const float* packed, const float* image, const Workspace& w, hipStream_t s, ProfileHook* hk) {
    PadPlanes pp{{w.a2, w.b2, w.a3, w.b3, w.a4, w.b4}, {w.L2, w.L2, w.L3, w.L3, w.L4, w.L4}, {64, 64, 64, 128, 128, 128}, 192};
    const int pr = w.prec;
    static const int c1slots = tuning_env("SPP_C1_SLOTS") ? atoi(tuning_env("SPP_C1_SLOTS")) : 64;
    static const int c1delay = tuning_env("SPP_C1_DELAY") ? atoi(tuning_env("SPP_C1_DELAY")) : C1_PHASE_DELAY;
    static const int c1abl = tuning_env("SPP_C1_ABL") ? atoi(tuning_env("SPP_C1_ABL")) : 0;
    static const bool fuse1 = !(tuning_env("SPP_FUSE_CONV1") && atoi(tuning_env("SPP_FUSE_CONV1")) == 0);   // 0: separate conv1a / conv1b (A/B timing)
    static const bool resident = !(tuning_env("SPP_RESIDENT") && atoi(tuning_env("SPP_RESIDENT")) == 0);   // 0: generic loop for conv2a / conv2b / conv3a
    // the resident-block kernel (split-fp16 only): gi = GEMM convolution, X = input (the image for the first layer), Y = output plane
    auto resident_conv = [&](auto c1a, auto pool, int gi, int kid, const float* X, float* Y, const FeatLayout& L, const FeatLayout& LY) {
        auto kern = conv1ab_pool_f16_kernel<decltype(c1a)::value, decltype(pool)::value>;
        static bool lds_ok[64] = {};                   // per instantiation of this lambda
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= 64 || !lds_ok[dev]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C1_SMEM_BYTES);
            if (dev >= 0 && dev < 64) lds_ok[dev] = true;
        }
        const int rows = kConv[gi].rows;
        const int NT = L.b * (L.H / 2) * ((L.W + 63) / 64) * (rows / 64);
        const int per = (NT + 7) / 8;                  // items per XCD band; 64 resident workgroups per XCD (2 per CU) walk it
        SPP_LAUNCH(hk, kid, s, kern, dim3(8 * std::min(per, c1slots)), dim3(512), C1_SMEM_BYTES, s, X, packed + PW_C1A_W, packed + PW_C1A_B,
                   reinterpret_cast<const unsigned short*>(packed + PW_TOTAL) + conv_wp_off(gi), packed + conv_b_off(gi), Y, L, LY, rows, pp,
                   c1delay, c1abl
#ifdef C1_PROBE
                   ,
                   (unsigned long long*)nullptr
#endif
        );
    };
    using std::true_type;
    using std::false_type;
    auto fits = [](const FeatLayout& L) { return (L.H & 1) == 0 && L.W % 64 == 0; };
    if (pr == 4 && fuse1 && (w.L1.H & 1) == 0) {
        // split-fp16: conv1a is recomputed inside conv1b's workgroups (its 64-channel full-resolution plane never exists)
        resident_conv(true_type{}, true_type{}, 0, KID_CONV1B, image, w.a2, w.L1, w.L2);
    } else {
        SPP_LAUNCH(hk, KID_CONV1A, s, conv1a_kernel, dim3((w.L1.ld + 255) / 256 + pp.nblocks, w.L1.b), dim3(256), 0, s, image,
                   packed + PW_C1A_W, packed + PW_C1A_B, w.a1, w.L1, pp);
        launch_conv_pool<64>(0, KID_CONV1B, packed, w.a1, w.b1, w.a2, w.L1, w.L2, s, hk, pr);     // conv1b + pool
    }
    if (pr == 4 && resident && fits(w.L2)) {
        resident_conv(false_type{}, false_type{}, 1, KID_CONV2, w.a2, w.b2, w.L2, w.L2);           // conv2a
        resident_conv(false_type{}, true_type{}, 2, KID_CONV2, w.b2, w.a3, w.L2, w.L3);            // conv2b + pool
    } else {
        launch_conv<64, 9>(1, KID_CONV2, packed, w.a2, w.b2, w.L2, 1, s, hk, pr);
        launch_conv_pool<64>(2, KID_CONV2, packed, w.b2, w.a2, w.a3, w.L2, w.L3, s, hk, pr);      // conv2b + pool (a2 is free: scratch)
    }
    if (pr == 4 && resident && fits(w.L3)) resident_conv(false_type{}, false_type{}, 3, KID_CONV3A, w.a3, w.b3, w.L3, w.L3);   // conv3a
    else launch_conv<64, 9>(3, KID_CONV3A, packed, w.a3, w.b3, w.L3, 1, s, hk, pr);
    launch_conv_pool<128>(4, KID_CONV3B, packed, w.b3, w.c3, w.a4, w.L3, w.L4, s, hk, pr);    // conv3b + pool
    launch_conv<128, 9>(5, KID_CONV4, packed, w.a4, w.b4, w.L4, 1, s, hk, pr);
    launch_conv<128, 9>(6, KID_CONV4, packed, w.b4, w.a4, w.L4, 1, s, hk, pr);
    launch_conv<128, 9>(7, KID_HEADS, packed, w.a4, w.hd, w.L4, 1, s, hk, pr);                                  // relu(convPa), relu(convDa)
    launch_conv<256, 1>(8, KID_CONVPB, packed, w.hd, w.lg, w.L4, 0, s, hk, pr);                                 // logits
    launch_conv<256, 1>(9, KID_CONVDB, packed, w.hd + (size_t)256 * w.L4.ldt, w.dd, w.L4, 0, s, hk, pr);       // descriptors
}

void launch_export_dense(const Workspace& w, float* dense_desc, hipStream_t s) {
    hipLaunchKernelGGL(export_dense_kernel, dim3((w.L4.H * w.L4.W + 63) / 64, DD / 64, w.L4.b), dim3(256), 0, s, w.dd, w.L4,
                       dense_desc);
}

}  // namespace spp
