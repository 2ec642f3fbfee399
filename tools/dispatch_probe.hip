// Dispatcher probe: how does the hardware place G workgroups of a kernel shaped like the GEMMs
// (256 threads, LDS bytes given) onto CUs, and when does each start/end?
//   hipcc --offload-arch=gfx950 -O2 tools/dispatch_probe.hip -o /tmp/probe && /tmp/probe 504 53248 20
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>
#include <algorithm>

struct Rec { unsigned hwid, xcc; unsigned long long t0, t1; };

__global__ __launch_bounds__(256) void probe(Rec* out, unsigned long long spin_ticks) {
    extern __shared__ float smem[];
    const unsigned long long t0 = wall_clock64();
    smem[threadIdx.x] = (float)t0;
    __syncthreads();
    while (wall_clock64() - t0 < spin_ticks) { __builtin_amdgcn_s_sleep(2); }
    if (threadIdx.x == 0) {
        Rec r;
        r.hwid = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID
        r.xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);   // HW_REG_XCC_ID
        r.t0 = t0; r.t1 = wall_clock64();
        out[blockIdx.x] = r;
    }
}

int main(int argc, char** argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 504;
    const int lds = argc > 2 ? atoi(argv[2]) : 53248;
    const int spin_us = argc > 3 ? atoi(argv[3]) : 20;
    Rec* d; hipMalloc(&d, sizeof(Rec) * G);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
    int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);  // kHz
    const unsigned long long ticks = (unsigned long long)spin_us * rate / 1000;
    for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(probe, dim3(G), dim3(256), lds, 0, d, ticks); hipDeviceSynchronize(); }
    std::vector<Rec> h(G); hipMemcpy(h.data(), d, sizeof(Rec) * G, hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull, tmax = 0;
    for (auto& r : h) { tmin = std::min(tmin, r.t0); tmax = std::max(tmax, r.t1); }
    std::map<unsigned, int> per_cu; std::map<unsigned, int> per_xcc;
    for (auto& r : h) {
        const unsigned cu = (r.hwid >> 8) & 0xf, sh = (r.hwid >> 12) & 1, se = (r.hwid >> 13) & 0x7, xcc = r.xcc & 0xf;
        per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu]++; per_xcc[xcc]++;
    }
    std::map<int, int> hist; for (auto& kv : per_cu) hist[kv.second]++;
    printf("G=%d lds=%d spin=%dus clock=%dkHz  span=%.2f us  distinct CUs=%zu\n", G, lds, spin_us, rate, (tmax - tmin) * 1e3 / rate, per_cu.size());
    printf("blocks per CU histogram:"); for (auto& kv : hist) printf("  %d blocks: %d CUs;", kv.first, kv.second); printf("\n");
    printf("blocks per XCC:"); for (auto& kv : per_xcc) printf(" x%u=%d", kv.first, kv.second); printf("\n");
    // start-time distribution
    std::vector<double> st; for (auto& r : h) st.push_back((r.t0 - tmin) * 1e3 / rate); std::sort(st.begin(), st.end());
    printf("start time us: p0=%.2f p50=%.2f p90=%.2f p99=%.2f max=%.2f\n", st[0], st[G / 2], st[G * 9 / 10], st[G * 99 / 100], st[G - 1]);
    int late = 0; for (double s : st) if (s > spin_us * 0.5) late++;
    printf("blocks that started after half the spin time (i.e. waited for a slot): %d\n", late);
    printf("first 16 blocks (block: xcc se sh cu start_us):"); for (int b = 0; b < 16; ++b) printf(" [%d: %u %u %u %u %.2f]", b, h[b].xcc & 0xf, (h[b].hwid >> 13) & 7, (h[b].hwid >> 12) & 1, (h[b].hwid >> 8) & 0xf, (h[b].t0 - tmin) * 1e3 / rate); printf("\n");
    return 0;
}
