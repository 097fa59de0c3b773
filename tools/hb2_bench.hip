// Standalone timing harness for k_mvn_enc_bwd_h_b2 (both wave shapes), synthetic operands at the headline shape, the library's
// own kernel source.  Timing only: the check lines compare the two shapes' outputs with each other, not with the oracle.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I vipsy_amd/csrc -o tools/hb2_bench tools/hb2_bench.hip
#include "../vipsy_amd/csrc/vx_common.h"
#include "../vipsy_amd/csrc/k_mvn_enc.hip"
#include "../vipsy_amd/csrc/k_pack.hip"
#include "../vipsy_amd/csrc/k_mvn_bwd_t.hip"
#include "../vipsy_amd/csrc/k_mvn_bwd_hb.hip"
#include "../vipsy_amd/csrc/k_mvn_bwd_hb2.hip"
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

template <int NSET>
static int run(const EncDims& dm, const uint8_t* img, const float* sc, const float* h, const float* eps, const float* gxT,
               const float* gdT, const float* hT, float* out, uint32_t* maxw, int64_t nb, std::vector<float>& host) {
    const size_t lds = hb2_lds_bytes(dm.D);
    CK(hipFuncSetAttribute((const void*)k_mvn_enc_bwd_h_b2<7, NSET>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_mvn_enc_bwd_h_b2<7, NSET>), dim3((unsigned)((nb + 255) / 256)), dim3(64 * HB2_WAVES_OF(NSET)), lds, 0, dm,
                           img, sc, h, eps, gxT, gdT, (float*)nullptr, hT, out, maxw);
        hipEventRecord(e1); CK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("k_mvn_enc_bwd_h_b2<7,%d> nb=%lld lds %zu: %.3f ms\n", NSET, (long long)nb, lds, ms);
    }
    host.resize(64 * 1024);
    for (int r = 0; r < 64; ++r) CK(hipMemcpy(host.data() + r * 1024, out + (int64_t)r * nb + (nb - 1024), 4096, hipMemcpyDeviceToHost));
    return 0;
}

__global__ void k_tr64(const float* h, float* hT, int64_t nb) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nb * 64; i += (int64_t)gridDim.x * blockDim.x)
        hT[(i & 63) * nb + (i >> 6)] = h[i];
}

int main(int argc, char** argv) {
    const int D = argc > 2 ? atoi(argv[2]) : 100;
    const int64_t nb = argc > 1 ? atoll(argv[1]) : 1000000;
    const int T = D * (D + 1) / 2;
    float *W21, *W22, *sc, *h, *eps, *gxT, *gdT, *hT, *out; uint8_t* img; uint32_t* maxw;
    CK(hipMalloc(&W21, D * 64 * 4)); CK(hipMalloc(&W22, (size_t)T * 64 * 4)); CK(hipMalloc(&sc, 64)); CK(hipMalloc(&maxw, 16));
    CK(hipMalloc(&img, hb_img_floats(D) * 4));
    CK(hipMalloc(&h, nb * 64 * 4)); CK(hipMalloc(&hT, nb * 64 * 4)); CK(hipMalloc(&out, nb * 64 * 4));
    CK(hipMalloc(&eps, nb * D * 4)); CK(hipMalloc(&gxT, nb * D * 4)); CK(hipMalloc(&gdT, nb * D * 4));
    k_fill<<<1024, 256>>>(W22, (int64_t)T * 64, 0.1f, 1); k_fill<<<64, 256>>>(W21, D * 64, 0.1f, 2);
    k_fill<<<4096, 256>>>(eps, nb * D, 3.0f, 3); k_fill<<<4096, 256>>>(gdT, nb * D, 2.0f, 4); k_fill<<<4096, 256>>>(gxT, nb * D, 2.0f, 5);
    k_fill<<<4096, 256>>>(h, nb * 64, 1.5f, 6);
    k_tr64<<<4096, 256>>>(h, hT, nb);                              // hT[hh][p] = h[p][hh]: the two layouts of the same values
    float hsc[16] = {0}; hsc[2] = 65536.f; hsc[3] = 1024.f;
    CK(hipMemcpy(sc, hsc, 64, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_pack_heads_hb, dim3(hb_units(D)), dim3(64), 0, 0, D, W21, W22, (const float*)sc, img, maxw);
    EncDims dm; dm.D = D; dm.J = 500; dm.H = 64; dm.Hp = 64; dm.DS = enc_ds(D); dm.T = T; dm.nb = nb;
    if (argc > 3) {                                              // small batch: the SPLIT form of k_mvn_bwd_hb.hip
        const size_t ldsh = hb_lds_bytes(D);
        CK(hipFuncSetAttribute((const void*)k_mvn_enc_bwd_h_b<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsh));
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_mvn_enc_bwd_h_b<true>, dim3((unsigned)((nb + 31) / 32)), dim3(HB_THREADS), ldsh, 0, dm, (const uint8_t*)img,
                               (const float*)sc, (const float*)h, (const float*)eps, (const float*)gxT, (const float*)gdT, (float*)nullptr,
                               (const float*)hT, out, maxw, (int64_t)0);
            hipEventRecord(e1); CK(hipEventSynchronize(e1));
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("k_mvn_enc_bwd_h_b<true> nb=%lld: %.1f us\n", (long long)nb, 1000.f * ms);
        }
        std::vector<float> o(64 * nb);
        CK(hipMemcpy(o.data(), out, 64 * nb * 4, hipMemcpyDeviceToHost));
        double cs = 0; for (size_t i = 0; i < o.size(); ++i) cs += (double)o[i] * (double)((i * 2654435761u) % 1000 + 1);
        printf("checksum %.17g\n", cs);
        return 0;
    }
    std::vector<float> a, b;
    if (run<2>(dm, img, sc, h, eps, gxT, gdT, hT, out, maxw, nb, a)) return 1;
    CK(hipMemset(out, 0, nb * 64 * 4));
    if (run<1>(dm, img, sc, h, eps, gxT, gdT, hT, out, maxw, nb, b)) return 1;
    double md = 0, mx = 0;
    for (size_t i = 0; i < a.size(); ++i) { md = fmax(md, fabs((double)a[i] - b[i])); mx = fmax(mx, fabs((double)a[i])); }
    printf("check: max |NSET2 - NSET1| = %g, max |out| = %g, samples %g %g %g\n", md, mx, a[0], a[1], a[5000]);
    return 0;
}
