// Standalone timing harness for the NormEncoder forward kernels (k_norm_enc_fwd_b, k_norm_enc_fwd_h<NP, PF>) at 1M x 500:
// synthetic operands, the library's own kernel sources.  Checks the variants against each other (h, loc, raw within 1e-5).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/nenc_bench tools/nenc_bench.hip
#include "../vipsy_amd/csrc/vx_common.h"
#include "../vipsy_amd/csrc/k_util.hip"
#include "../vipsy_amd/csrc/k_mvn_enc.hip"
#include "../vipsy_amd/csrc/k_mvn_packed.hip"
#include "../vipsy_amd/csrc/k_irt_lik.hip"
#include "../vipsy_amd/csrc/k_irt_lik_r.hip"
#include "../vipsy_amd/csrc/k_irt_lik_b.hip"
#include "../vipsy_amd/csrc/k_irt_lik_h.hip"
#include "../vipsy_amd/csrc/k_mvn_bwd_t.hip"
#include "../vipsy_amd/csrc/k_mvn_bwd_b.hip"
#include "../vipsy_amd/csrc/k_mvn_fwd_b.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 1; } } while (0)

__global__ void k_fill(float* p, int64_t n, float amp, uint32_t seed) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = amp * ((float)(x & 0xFFFF) / 32768.0f - 1.0f);
    }
}
__global__ void k_fill_y(uint8_t* p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u + 77u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (x % 10u) ? (uint8_t)255 : (uint8_t)((x >> 7) & 1);      // 90 % missing
    }
}
__global__ void k_maxdiff(const float* a, const float* b, int64_t n, float* out) {
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(a[i] - b[i]));
    atomicMax((unsigned int*)out, __builtin_bit_cast(unsigned int, m));
}

int main(int argc, char** argv) {
    const int H = 64, J = argc > 2 ? atoi(argv[2]) : 500;
    const int64_t nb = argc > 1 ? atoll(argv[1]) : 1000000;
    float *W1, *b1, *W21, *b21, *W22, *b22, *h0, *h1, *loc0, *loc1, *raw0, *raw1, *packws, *md;
    uint8_t* y;
    CK(hipMalloc(&W1, H * J * 4)); CK(hipMalloc(&b1, H * 4)); CK(hipMalloc(&W21, 64 * 4)); CK(hipMalloc(&b21, 4));
    CK(hipMalloc(&W22, 64 * 4)); CK(hipMalloc(&b22, 4)); CK(hipMalloc(&y, nb * J + 4096));
    CK(hipMalloc(&h0, nb * 64 * 4)); CK(hipMalloc(&h1, nb * 64 * 4)); CK(hipMalloc(&loc0, nb * 4)); CK(hipMalloc(&loc1, nb * 4));
    CK(hipMalloc(&raw0, nb * 4)); CK(hipMalloc(&raw1, nb * 4)); CK(hipMalloc(&packws, nh_pack_floats(J) * 4)); CK(hipMalloc(&md, 16));
    k_fill<<<256, 256>>>(W1, H * J, 0.045f, 1); k_fill<<<1, 64>>>(b1, H, 0.045f, 2);
    k_fill<<<1, 64>>>(W21, 64, 0.125f, 3); k_fill<<<1, 1>>>(b21, 1, 0.125f, 4);
    k_fill<<<1, 64>>>(W22, 64, 0.125f, 5); k_fill<<<1, 1>>>(b22, 1, 0.125f, 6);
    k_fill_y<<<4096, 256>>>(y, nb * J);
    EncDims dm; dm.D = 1; dm.J = J; dm.H = 64; dm.Hp = 64; dm.DS = 3; dm.T = 1; dm.nb = nb;
    uint8_t* w1img = (uint8_t*)packws;
    float* sc = packws + fb_w1img_floats(J);
    float* part = sc + 16;
    const int n_ks = (J + 15) / 16;
    hipLaunchKernelGGL(k_norm_pack_max, dim3(NH_MAX_BLOCKS), dim3(256), 0, 0, J, W1, part);
    hipLaunchKernelGGL(k_norm_pack_w1, dim3(n_ks), dim3(256), 0, 0, J, W1, (const float*)part, sc, w1img);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int N = 20;
    auto timeit = [&](const char* name, auto launch) -> int {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipDeviceSynchronize());
        hipEventRecord(e0);
        for (int i = 0; i < N; ++i) launch();
        hipEventRecord(e1); CK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1);
        CK(hipMemset(md, 0, 16));
        k_maxdiff<<<1024, 256>>>(h0, h1, nb * 64, md); k_maxdiff<<<256, 256>>>(loc0, loc1, nb, md + 1); k_maxdiff<<<256, 256>>>(raw0, raw1, nb, md + 2);
        float d[3]; CK(hipMemcpy(d, md, 12, hipMemcpyDeviceToHost));
        printf("%-44s %8.1f us a launch;  against k_norm_enc_fwd_b: h %.2e loc %.2e raw %.2e\n", name, 1e3 * ms / N, d[0], d[1], d[2]);
        return 0;
    };
    {
        const size_t lds = nb_lds_bytes(J);
        CK(hipFuncSetAttribute((const void*)k_norm_enc_fwd_b, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_norm_enc_fwd_b, dim3((unsigned)((nb + 127) / 128)), dim3(NB_THREADS), lds, 0, dm, y, (const int64_t*)nullptr, W1, b1, W21, b21, W22, b22, h0, loc0, raw0);
        CK(hipDeviceSynchronize());
        if (timeit("k_norm_enc_fwd_b (bf16x3, split in the loop)", [&]() {
            hipLaunchKernelGGL(k_norm_enc_fwd_b, dim3((unsigned)((nb + 127) / 128)), dim3(NB_THREADS), lds, 0, dm, y, (const int64_t*)nullptr, W1, b1, W21, b21, W22, b22, h1, loc1, raw1); })) return 1;
    }
#define RUN_H(NP, PF)                                                                                                             \
    {                                                                                                                             \
        const size_t lds = nh_lds_bytes<NP>(J);                                                                                   \
        CK(hipFuncSetAttribute((const void*)k_norm_enc_fwd_h<NP, PF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));     \
        if (timeit("k_norm_enc_fwd_h<" #NP ", " #PF ">", [&]() {                                                                   \
            hipLaunchKernelGGL((k_norm_enc_fwd_h<NP, PF>), dim3((unsigned)((nb + 127) / 128)), dim3(64 * (4 / NP)), lds, 0, dm, y, \
                               (const int64_t*)nullptr, (const uint8_t*)w1img, (const float*)sc, b1, W21, b21, W22, b22, h1, loc1, raw1); })) return 1; \
    }
    RUN_H(1, 3) RUN_H(1, 7) RUN_H(2, 3) RUN_H(2, 11)
    return 0;
}
