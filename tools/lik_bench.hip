// Stand-alone timing harness for k_irt_lik_r (headline shape) with ablation variants.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I vipsy_amd/csrc -o tools/lik_bench tools/lik_bench.hip
#include "k_irt_lik_r.hip"
#include <cstdio>
#include <vector>

template <int ABL>
static float run(const LikRDims& dm, const uint8_t* y, const float* x, const float* a, const float* b, float* gxp,
                 float* llp, float* slabs, int reps) {
    const size_t lds = likr_lds_bytes(dm.XS);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_irt_lik_r<0, 13, 1, ABL>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const dim3 grid(dm.groups * dm.n_pr);
    for (int w = 0; w < 2; ++w)
        hipLaunchKernelGGL((k_irt_lik_r<0, 13, 1, ABL>), grid, dim3(LR_THREADS), lds, 0, dm, y, nullptr, x, a, b, nullptr,
                           nullptr, gxp, llp, slabs);
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL((k_irt_lik_r<0, 13, 1, ABL>), grid, dim3(LR_THREADS), lds, 0, dm, y, nullptr, x, a, b, nullptr,
                           nullptr, gxp, llp, slabs);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) printf("error: %s\n", hipGetErrorString(err));
    return ms / reps;
}

int main() {
    const int D = 100, J = 500;
    const int64_t nb = 1000000;
    LikRDims dm;
    dm.D = D; dm.J = J; dm.K8 = 104; dm.XS = 108; dm.model = 2; dm.fast = 1; dm.groups = 4; dm.n_pr = 64;
    dm.Dc = 1.f; dm.scale = 1.f; dm.nb = nb; dm.slab_len = (int64_t)D * J + 3 * J; dm.gxt = 1;
    uint8_t* y; float *x, *a, *b, *gxp, *llp, *slabs;
    hipMalloc(&y, nb * J); hipMalloc(&x, nb * D * 4); hipMalloc(&a, D * J * 4); hipMalloc(&b, J * 4);
    hipMalloc(&gxp, (size_t)dm.groups * nb * D * 4); hipMalloc(&llp, (size_t)dm.groups * nb * 4);
    hipMalloc(&slabs, (size_t)dm.n_pr * dm.slab_len * 4);
    std::vector<uint8_t> hy(nb * J);
    std::vector<float> hx(nb * D), ha(D * J), hb(J);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0f / 16777216.0f); };
    for (auto& v : hy) v = rnd() < 0.5f ? 1 : 0;
    for (auto& v : hx) v = rnd() - 0.5f;
    for (auto& v : ha) v = 0.2f * (rnd() - 0.5f);
    for (auto& v : hb) v = rnd() - 0.5f;
    hipMemcpy(y, hy.data(), hy.size(), hipMemcpyHostToDevice);
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(a, ha.data(), ha.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
#define RUN(ABL, what) printf("ABL=%3d  %-46s %.3f ms\n", ABL, what, run<ABL>(dm, y, x, a, b, gxp, llp, slabs, 5))
    RUN(0, "full");
    RUN(1, "no cell math");
    RUN(2, "no R/LP LDS writes");
    RUN(3, "no cell math, no R/LP writes");
    RUN(4, "no gx store");
    RUN(8, "no staging of next tile");
    RUN(16, "no barriers");
    RUN(32, "no ll reduce");
    RUN(44, "no gx store, staging, ll");
    RUN(63, "MFMA + LDS operand reads only");
    RUN(64, "DMA issued, never waited for");
    RUN(128, "no x DMA");
    RUN(256, "no y DMA");
    RUN(384, "no DMA, but wait+loop structure");
    return 0;
}
