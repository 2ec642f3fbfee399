// Cost of a software grid barrier on MI355X (8 XCDs, non-coherent L2s): G persistent workgroups of 512 threads, each
// round every thread writes `bytes_per_round / (G * 512)` bytes of fresh data (dirty L2 lines the release fence has to
// write back), then all workgroups meet at the barrier and read a neighbour's data (acquire side).  Prints us / round.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gbp tools/grid_barrier_probe.hip && /tmp/gbp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > (1u << 22)) { ok = false; break; }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    return ok;
}

__global__ __launch_bounds__(512) void probe(float* buf, size_t floats_per_wg, unsigned* counter, int rounds, int* err, float* sink) {
    const int G = gridDim.x, w = blockIdx.x;
    float acc = 0.f;
    for (int r = 0; r < rounds; ++r) {
        float* mine = buf + (size_t)w * floats_per_wg;
        for (size_t i = threadIdx.x; i < floats_per_wg; i += 512) mine[i] = (float)(r + i);
        if (!grid_barrier(counter, (unsigned)G * (r + 1))) { *err = 1; return; }
        const float* other = buf + (size_t)((w + G / 2 + 1) % G) * floats_per_wg;   // a workgroup on another XCD
        for (size_t i = threadIdx.x; i < floats_per_wg; i += 512) acc += other[i];
        if (!grid_barrier(counter + 1, (unsigned)G * (r + 1))) { *err = 1; return; }   // readers done before the next overwrite
    }
    if (acc == 12345.678f) *sink = acc;
}

__global__ void empty_kernel(float* p) { if (p == nullptr && threadIdx.x == 9999) *p = 0; }

int main() {
    float *buf, *sink;
    unsigned* counter;
    int* err;
    hipMalloc(&buf, 256 << 20); hipMalloc(&sink, 4); hipMalloc(&counter, 64); hipMalloc(&err, 4);
    hipMemset(err, 0, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int rounds = 50;
    for (int G : {256, 504, 512, 768}) {
        for (size_t mb : {0, 1, 16, 64}) {
            size_t fpw = mb ? (mb << 20) / 4 / G : 0;
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                hipMemset(counter, 0, 64);
                hipDeviceSynchronize();
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(probe, dim3(G), dim3(512), 0, 0, buf, fpw, counter, rounds, err, sink);
                hipEventRecord(e1, 0);
                hipDeviceSynchronize();
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            int h = 0; hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost);
            printf("G=%d  %zu MB written+read per round: %.2f us per round (2 barriers)%s\n", G, mb, best * 1e3f / rounds, h ? "  [BARRIER TIMEOUT]" : "");
            if (h) { hipMemset(err, 0, 4); }
        }
    }
    // for comparison: dependent tiny launches
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(empty_kernel, dim3(512), dim3(512), 0, 0, buf);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("200 dependent empty launches of 512x512: %.2f us per launch\n", ms * 1e3f / 200);
    return 0;
}
