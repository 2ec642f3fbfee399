// Micro-benchmark (gfx950): do v_mfma_f32_32x32x16_f16 and ordinary VALU instructions overlap (a) inside one wave, (b) between the two
// waves of a SIMD?  Each wave runs ITERS x { NM MFMAs, NV VALU FMAs } in one of three orders; cycles per iteration from s_memtime.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mb tools/microbench/mfma_valu_overlap.hip && /tmp/mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define SB() __builtin_amdgcn_sched_barrier(0)

template <int ORDER, int NM, int NV, int LDSR, int RUN = 1, int NACC = 4, int DMA = 0>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters, const char* src) {
    __shared__ float lds[8192];   // 32 KiB
    __shared__ __attribute__((aligned(16))) char ring[3][8192];   // DMA target (pieces wrap inside 8 KiB: timing only)
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 0.001f + j); b[j] = (_Float16)(j * 0.5f); }
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = threadIdx.x + j;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i;
    __syncthreads();
    const float c = out[0];
    float4 l4 = make_float4(0, 0, 0, 0), pend0 = l4, pend1 = l4;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if constexpr (ORDER == 0) {   // interleaved: NV / NM VALU behind each MFMA
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                acc[(m / RUN) % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[(m / RUN) % NACC], 0, 0, 0);
                SB();
#pragma unroll
                for (int q = 0; q < NV / (NM ? NM : 1); ++q) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[q & 7]) : "v"(c));
                if constexpr (DMA > 0) {   // DMA LDS-DMA pieces (1 KiB each) per iteration and wave, at evenly spaced MFMA slots; sources rotate over 8 MiB (L2)
                    if (m % (NM / DMA) == 0) {
                        const int piece = (m / (NM / DMA)) * (blockDim.x >> 6) + (threadIdx.x >> 6);
                        const char* g = src + (((size_t)blockIdx.x * 131 + it) & 255) * 32768 + piece * 1024 + (threadIdx.x & 63) * 16;
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                         (__attribute__((address_space(3))) void*)(ring[it % 3] + (piece & 7) * 1024), 16, 0, 0);
                    }
                }
                if constexpr (LDSR) {   // LDSR 16-byte LDS reads per NM MFMAs, spread evenly; each value is consumed one slot later
                    if (((m + 1) * LDSR) / NM > (m * LDSR) / NM) {
                        l4 += pend0;
                        pend0 = *reinterpret_cast<const float4*>(&lds[(threadIdx.x * 4 + m * 256 + it * 64) & 8188]);
                    }
                    if (((m + 1) * LDSR) / NM > (m * LDSR) / NM + 1) {
                        l4 += pend1;
                        pend1 = *reinterpret_cast<const float4*>(&lds[(threadIdx.x * 4 + m * 256 + it * 64 + 128) & 8188]);
                    }
                }
                SB();
            }
        } else {   // clustered: all MFMAs, then all VALU
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                acc[(m / RUN) % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[(m / RUN) % NACC], 0, 0, 0);
                SB();
            }
#pragma unroll
            for (int q = 0; q < NV; ++q) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[q & 7]) : "v"(c));
            SB();
        }
        if constexpr (DMA > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA) : "memory");   // the previous iteration's pieces have landed
        if constexpr (ORDER == 2 || LDSR > 1) __syncthreads();   // clustered + a workgroup barrier per iteration (keeps the waves in phase)
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    l4 += pend0; l4 += pend1;
    float s = l4.x + l4.y + l4.z + l4.w;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int j = 0; j < 8; ++j) s += v[j];
    if (s == 12345.678f) out[1] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int ORDER, int NM, int NV, int LDSR = 0, int RUN = 1, int NACC = 4, int DMA = 0>
static void run(const char* what, int threads) {
    float* out; unsigned long long* cyc;
    static char* src = nullptr;
    if (!src) { hipMalloc(&src, 256 * 32768 + 65536); hipMemset(src, 0, 256 * 32768 + 65536); }
    hipMalloc(&out, 64); hipMemset(out, 0, 64); hipMalloc(&cyc, 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<ORDER, NM, NV, LDSR, RUN, NACC, DMA><<<256, threads>>>(out, cyc, iters, src);
    hipEventRecord(e0);
    k<ORDER, NM, NV, LDSR, RUN, NACC, DMA><<<256, threads>>>(out, cyc, iters, src);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-58s waves/SIMD %d  NM %2d NV %3d : %7.1f s_memtime ticks/iter  %8.3f ns/iter\n", what, threads / 256, NM, NV, (double)c / iters, ms * 1e6 / iters);
    hipFree(out); hipFree(cyc);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    for (int threads : {256, 512}) {
        run<1, 16, 0>("MFMAs only", threads);
        run<1, 0, 80>("VALU only", threads);
        run<1, 0, 160>("VALU only", threads);
        run<0, 16, 80>("interleaved (5 VALU behind each MFMA)", threads);
        run<1, 16, 80>("clustered (16 MFMAs, then 80 VALU)", threads);
        run<2, 16, 80>("clustered + workgroup barrier per iteration", threads);
        run<0, 16, 160>("interleaved (10 VALU behind each MFMA)", threads);
        run<1, 16, 160>("clustered (16 MFMAs, then 160 VALU)", threads);
        run<2, 16, 160>("clustered + workgroup barrier per iteration", threads);
        run<0, 16, 80, 8>("interleaved + barrier +  8 ds_read_b128 per 16 MFMAs", threads);
        run<0, 16, 80, 12>("interleaved + barrier + 12 ds_read_b128 per 16 MFMAs", threads);
        run<0, 16, 80, 16>("interleaved + barrier + 16 ds_read_b128 per 16 MFMAs", threads);
        run<0, 16, 80, 24>("interleaved + barrier + 24 ds_read_b128 per 16 MFMAs", threads);
        run<0, 16, 80, 12, 1, 4, 4>("interleaved + barrier + 12 reads + 4 DMA pieces per wave", threads);
        run<0, 16, 80, 12, 1, 4, 8>("interleaved + barrier + 12 reads + 8 DMA pieces per wave", threads);
        run<0, 16, 80, 0, 1, 4, 4>("interleaved + 4 DMA pieces per wave (no reads, no barrier)", threads);
        run<1, 16, 0, 0, 1, 1>("MFMAs only, ONE accumulator (dependent chain)", threads);
        run<1, 16, 0, 0, 1, 2>("MFMAs only, two accumulators alternating", threads);
        run<1, 16, 0, 0, 4, 2>("MFMAs only, two accumulators, runs of 4 on each", threads);
        run<0, 16, 80, 0, 4, 2>("interleaved 5 VALU, two accumulators, runs of 4", threads);
        run<0, 16, 40, 0, 4, 2>("interleaved 2.5 VALU (40), two accumulators, runs of 4", threads);
    }
    return 0;
}
