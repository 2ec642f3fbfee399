// XCD-hierarchical software grid barrier on MI355X (the form MI355X_MICROARCH.md prices as "barrier-xcd"), measured next to the flat
// counter of tools/grid_barrier_probe.hip and a dependent empty launch.  Workgroup b arrives on the counter of group b % 8 (the XCD
// round-robin dispatch puts it on; correctness does not depend on that), the last arriver of a group is its leader: release
// fence -> top counter -> spin until all 8 leaders arrived -> acquire fence -> publish the group's generation; the other
// workgroups spin on their group's generation with relaxed loads + s_sleep, then one acquire fence.
//   strict = 1: EVERY workgroup issues the release fence before it arrives (placement-independent visibility of its own stores);
//   strict = 0: only the leader does (valid when a group really shares one L2 -- what the guide's 4.1 / 5.9 us figure is).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/xbp tools/xcd_barrier_probe.hip && /tmp/xbp
#include <hip/hip_runtime.h>
#include <stdio.h>

struct Bar {
    unsigned cnt[8][32];   // per-group arrival counters, one 128-byte line each
    unsigned gen[8][32];   // per-group generation
    unsigned top[32];
    int err;
};

__device__ __forceinline__ unsigned ld_rlx(unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void xcd_barrier(Bar* b, unsigned epoch, int strict) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const int g = blockIdx.x & 7;
        const unsigned ng = (gridDim.x + 7 - g) / 8;   // workgroups with blockIdx % 8 == g
        if (strict) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned old = __hip_atomic_fetch_add(&b->cnt[g][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        if (old == epoch * ng - 1) {   // leader of the group
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(&b->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (ld_rlx(&b->top[0]) < epoch * 8) {
                if (++spins > (1u << 22)) { b->err = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(&b->gen[g][0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (ld_rlx(&b->gen[g][0]) < epoch) {
                if (++spins > (1u << 22)) { b->err = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(512) void probe(float* buf, size_t floats_per_wg, Bar* bar, int rounds, int strict, float* sink) {
    const int G = gridDim.x, w = blockIdx.x;
    float acc = 0.f;
    for (int r = 0; r < rounds; ++r) {
        float* mine = buf + (size_t)w * floats_per_wg;
        for (size_t i = threadIdx.x; i < floats_per_wg; i += 512) mine[i] = (float)(r + i);
        xcd_barrier(bar, 2 * r + 1, strict);
        const float* other = buf + (size_t)((w + G / 2 + 1) % G) * floats_per_wg;   // a workgroup of another group
        for (size_t i = threadIdx.x; i < floats_per_wg; i += 512) acc += other[i];
        xcd_barrier(bar, 2 * r + 2, strict);   // readers done before the next overwrite
    }
    if (acc == 12345.678f) *sink = acc;
}

__global__ void empty_kernel(float* p) { if (p == nullptr && threadIdx.x == 9999) *p = 0; }

int main() {
    float *buf, *sink;
    Bar* bar;
    hipMalloc(&buf, 256 << 20); hipMalloc(&sink, 4); hipMalloc(&bar, sizeof(Bar));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int rounds = 50;
    for (int strict : {0, 1})
        for (int G : {256, 504, 512}) {
            for (size_t mb : {0, 1, 16}) {
                size_t fpw = mb ? (mb << 20) / 4 / G : 0;
                float best = 1e9f;
                int bad = 0;
                for (int rep = 0; rep < 3; ++rep) {
                    hipMemset(bar, 0, sizeof(Bar));
                    hipDeviceSynchronize();
                    hipEventRecord(e0, 0);
                    hipLaunchKernelGGL(probe, dim3(G), dim3(512), 0, 0, buf, fpw, bar, rounds, strict, sink);
                    hipEventRecord(e1, 0);
                    hipDeviceSynchronize();
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                    Bar h; hipMemcpy(&h, bar, sizeof(Bar), hipMemcpyDeviceToHost);
                    bad |= h.err;
                }
                printf("xcd-hierarchical barrier, release by %s: G=%d  %zu MB written+read per round: %.2f us per round (2 barriers) -> %.2f us per barrier%s\n",
                       strict ? "every workgroup" : "the group leaders only", G, mb, best * 1e3f / rounds, best * 1e3f / rounds / 2,
                       bad ? "  [TIMEOUT]" : "");
            }
        }
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(empty_kernel, dim3(512), dim3(512), 0, 0, buf);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("200 dependent empty launches of 512x512: %.2f us per launch\n", ms * 1e3f / 200);
    return 0;
}
