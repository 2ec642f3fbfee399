// Per-interval timestamps of conv1ab_pool_f16_kernel (workgroup 8 = XCD 0, slot 1; both halves): s_memtime before and after every
// barrier of the first patch pairs.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DC1_PROBE tools/conv1ab_probe.hip -o tools/bin/conv1ab_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "../onepose_amd/csrc/spp_conv_kernels.hip"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int abl = argc > 1 ? atoi(argv[1]) : 0, delay = argc > 2 ? atoi(argv[2]) : spp::C1_PHASE_DELAY;
    const int H = 512, W = 512;
    spp::FeatLayout L1 = spp::make_feat_layout(1, H, W), L2 = spp::make_feat_layout(1, H / 2, W / 2);
    std::vector<float> img(H * W), w1a(64 * 9), b1a(64), bias(64);
    unsigned r = 12345;
    auto rnd = [&]() { r = r * 1664525u + 1013904223u; return (float)(r >> 8) / 16777216.f; };
    for (auto& v : img) v = rnd();
    for (auto& v : w1a) v = rnd() - 0.5f;
    for (auto& v : b1a) v = 0.1f * (rnd() - 0.5f);
    for (auto& v : bias) v = 0.1f * (rnd() - 0.5f);
    std::vector<unsigned short> wp(2 * 64 * 576);
    for (auto& v : wp) v = 0x2800 + (unsigned short)(rnd() * 1024);      // small positive fp16 values
    float *dimg, *dw, *db, *dbias, *dy;
    unsigned short* dwp;
    unsigned long long* dprobe;
    const size_t ybytes = sizeof(float) * ((size_t)64 * L2.ldt + 2 * spp::feat_guard(L2));
    CK(hipMalloc(&dimg, img.size() * 4)); CK(hipMalloc(&dw, w1a.size() * 4)); CK(hipMalloc(&db, 256)); CK(hipMalloc(&dbias, 256));
    CK(hipMalloc(&dwp, wp.size() * 2)); CK(hipMalloc(&dy, ybytes)); CK(hipMalloc(&dprobe, 136 * 8));
    CK(hipMemcpy(dimg, img.data(), img.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, w1a.data(), w1a.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, b1a.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dbias, bias.data(), 256, hipMemcpyHostToDevice));
    CK(hipMemcpy(dwp, wp.data(), wp.size() * 2, hipMemcpyHostToDevice)); CK(hipMemset(dy, 0, ybytes)); CK(hipMemset(dprobe, 0, 136 * 8));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(spp::conv1ab_pool_f16_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)spp::C1_SMEM_BYTES));
    const int NT = (H / 2) * ((W + 63) / 64), per = (NT + 7) / 8;
    spp::PadPlanes pp = {};                       // no planes: C = 0
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float* yout = dy + spp::feat_guard(L2);
    for (int it = 0; it < 5; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((spp::conv1ab_pool_f16_kernel<true, true>), dim3(8 * std::min(per, 64)), dim3(512), spp::C1_SMEM_BYTES, 0, dimg, dw, db, dwp, dbias,
                           yout, L1, L2, 64, pp, delay, abl, dprobe);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("launch %d: %.1f us\n", it, ms * 1e3f);
    }
    std::vector<unsigned long long> pr(136);
    CK(hipMemcpy(pr.data(), dprobe, 136 * 8, hipMemcpyDeviceToHost));
    printf("workgroup 8 lifetime: %llu shader cycles, %.2f us of the 100 MHz real-time counter -> %.0f MHz\n", pr[129] - pr[128], (pr[131] - pr[130]) * 0.01,
           (double)(pr[129] - pr[128]) / ((pr[131] - pr[130]) * 0.01));
    for (int h = 0; h < 2; ++h) {
        printf("workgroup %s: stamp, cycles since the first stamp of workgroup 8, delta\n", h ? "8 + 8 * slots / 2 (delayed start)" : "8");
        const unsigned long long* q = pr.data() + h * 64;
        for (int i = 0; i < 64 && q[i]; ++i) printf("  s%02d  t=%8llu  d=%6lld\n", i, q[i] - pr[0], i ? (long long)(q[i] - q[i - 1]) : 0LL);
    }
    return 0;
}
