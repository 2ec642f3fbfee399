// fp32 MFMA GEMM main loop for gfx950 (v_mfma_f32_32x32x2_f32: exact f32 FMA chain at the
// 157 TFLOP/s matrix rate).  C[BM x BN] += A[BM x K] * B[K x BN], one 256-thread workgroup
// (4 wave64s arranged WM x WN), K consumed in BK=32 slabs, LDS double-buffered with register
// staging: the global loads of slab t+1 are issued before the MFMAs of slab t and written to the
// other LDS buffer after them -- one barrier per slab (cdna guide T14 "issue early / write late").
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

template <int BM_, int BN_, int WM_, int WN_, bool AKM_>
struct GemmTile {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_;
    static constexpr bool AKM = AKM_;
    static constexpr int TM = BM / WM / 32;   // 32x32 MFMA tiles per wave along M
    static constexpr int TN = BN / WN / 32;   // ... along N
    static constexpr int A_STRIDE = AKM ? BM : (BK + 4);
    static constexpr int A_FLOATS = AKM ? BK * BM : BM * (BK + 4);
    static constexpr int B_FLOATS = BK * BN;
    static constexpr int STAGE_FLOATS = A_FLOATS + B_FLOATS;
    static constexpr int SMEM_FLOATS = 2 * STAGE_FLOATS;
    static constexpr int A_VEC = BM * BK / 4 / 256;   // float4 per thread per slab
    static constexpr int B_VEC = BK * BN / 4 / 256;
    static_assert(WM * WN == 4, "256-thread workgroup = 4 waves");
    static_assert(TM >= 1 && TN >= 1, "wave tile must hold at least one 32x32 MFMA tile");
    static_assert(A_VEC >= 1 && B_VEC >= 1, "tile too small for 256 threads");
};

// row (within a 32x32 MFMA tile) held by accumulator register r of a lane in half `half`
__device__ __forceinline__ int mfma_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// aload(kt, r, c): float4 of the A slab kt.  row-major A: rows r of the tile, k offset c (0,4,..,28).
//                  KM A: slab row k = r (0..31), tile column c (multiple of 4).
// bload(kt, k, c): float4 of B slab kt, slab row k (0..31), tile column c (multiple of 4).
// baux(kt, k):     per-row float2 fetched together with the B slab (e.g. InstanceNorm mean / rstd);
// bxform(v, aux):  applied to the B registers when they are written to LDS (after the MFMAs), so
//                  the raw load has a whole slab of MFMA time to land.
struct NoAux {
    __device__ __forceinline__ float2 operator()(int, int) const { return make_float2(0.f, 0.f); }
};
struct NoXform {
    __device__ __forceinline__ void operator()(vf4&, float2) const {}
};

// ABLATE (profiling only, wrong results): 1 = no global loads in the steady-state loop,
//                                         2 = no global loads and no LDS writes (pure LDS-read + MFMA loop)
//                                         3 = steady-state loop skipped (fixed cost: prologue + 1 slab + epilogue)
template <class T, class ALoad, class BLoad, class BAux = NoAux, class BXform = NoXform, int ABLATE = 0>
__device__ __forceinline__ void gemm_mainloop(f32x16 (&acc)[T::TM][T::TN], float* smem, int KT, ALoad aload,
                                              BLoad bload, BAux baux = BAux(), BXform bxform = BXform()) {
    constexpr int BM = T::BM, BN = T::BN, TM = T::TM, TN = T::TN;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / T::WN, wn = wave % T::WN;
    const int half = lane >> 5, l31 = lane & 31;

    vf4 ra[T::A_VEC], rb[T::B_VEC];
    float2 rx[T::B_VEC];

    auto gload = [&](int kt) {
#pragma unroll
        for (int p = 0; p < T::A_VEC; ++p) {
            const int idx = p * 256 + tid;
            if constexpr (T::AKM) {
                ra[p] = aload(kt, idx / (BM / 4), (idx % (BM / 4)) * 4);
            } else {
                ra[p] = aload(kt, idx / (BK / 4), (idx % (BK / 4)) * 4);
            }
        }
#pragma unroll
        for (int p = 0; p < T::B_VEC; ++p) {
            const int idx = p * 256 + tid;
            rb[p] = bload(kt, idx / (BN / 4), (idx % (BN / 4)) * 4);
            rx[p] = baux(kt, idx / (BN / 4));
        }
    };
    auto swrite = [&](float* stage) {
        float* As = stage;
        float* Bs = stage + T::A_FLOATS;
#pragma unroll
        for (int p = 0; p < T::A_VEC; ++p) {
            const int idx = p * 256 + tid;
            if constexpr (T::AKM) {
                *reinterpret_cast<vf4*>(As + (idx / (BM / 4)) * BM + (idx % (BM / 4)) * 4) = ra[p];
            } else {
                *reinterpret_cast<vf4*>(As + (idx / (BK / 4)) * T::A_STRIDE + (idx % (BK / 4)) * 4) = ra[p];
            }
        }
#pragma unroll
        for (int p = 0; p < T::B_VEC; ++p) {
            const int idx = p * 256 + tid;
            vf4 v = rb[p];
            bxform(v, rx[p]);
            *reinterpret_cast<vf4*>(Bs + (idx / (BN / 4)) * BN + (idx % (BN / 4)) * 4) = v;
        }
    };
    auto compute = [&](const float* stage) {
        const float* As = stage;
        const float* Bs = stage + T::A_FLOATS;
        float a[TM][16];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            if constexpr (T::AKM) {
#pragma unroll
                for (int s = 0; s < 16; ++s)
                    a[tm][s] = As[(half * 16 + s) * BM + wm * TM * 32 + tm * 32 + l31];
            } else {
                const vf4* ap =
                    reinterpret_cast<const vf4*>(As + (wm * TM * 32 + tm * 32 + l31) * T::A_STRIDE + half * 16);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const vf4 x = ap[v];
                    a[tm][4 * v + 0] = x[0]; a[tm][4 * v + 1] = x[1]; a[tm][4 * v + 2] = x[2]; a[tm][4 * v + 3] = x[3];
                }
            }
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            float bv[TN];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) bv[tn] = Bs[(half * 16 + s) * BN + wn * TN * 32 + tn * 32 + l31];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][s], bv[tn], acc[tm][tn], 0, 0, 0);
        }
    };

    gload(0);
    swrite(smem);
    __syncthreads();
    // Steady state: branch-free body (the last slab is peeled) so that hipcc cannot sink the global
    // loads into a conditional block next to their vmcnt wait; sched_barrier pins the issue order
    // loads -> MFMAs -> LDS writes, i.e. the loads have the whole slab of MFMA time to land.
    for (int kt = 0; kt + 1 < (ABLATE == 3 ? 1 : KT); ++kt) {
        float* cur = smem + (kt & 1) * T::STAGE_FLOATS;
        float* nxt = smem + ((kt + 1) & 1) * T::STAGE_FLOATS;
        if constexpr (ABLATE == 0) gload(kt + 1);
        asm volatile("" ::: "memory");  // SelectionDAG-level pin (sched_barrier alone is not a memory fence)
        __builtin_amdgcn_sched_barrier(0);
        compute(cur);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
        if constexpr (ABLATE <= 1) swrite(nxt);
        __syncthreads();
    }
    compute(smem + ((KT - 1) & 1) * T::STAGE_FLOATS);
    __syncthreads();  // callers re-use the LDS for their epilogue
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

__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x : expm1f(x); }

}  // namespace gatsspg
