// Standalone timing harness for k_mvn_enc_bwd_w_b (compiles in ~20 s instead of the library's 2.5 min): synthetic operands
// at the headline shape, the library's own kernel source.  -D(timing only: the check line is a spot value, not a parity test).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I vipsy_amd/csrc -o tools/bwb_bench tools/bwb_bench.hip
#include "../vipsy_amd/csrc/vx_common.h"
#include "../vipsy_amd/csrc/k_mvn_enc.hip"
#include "../vipsy_amd/csrc/k_pack.hip"
#include "../vipsy_amd/csrc/k_mvn_bwd_t.hip"
#ifdef BWB_KERNEL_FILE                                  // another generation of the kernel (e.g. a saved copy of round 4's file)
#include BWB_KERNEL_FILE
#else
#include "../vipsy_amd/csrc/k_mvn_bwd_b.hip"
#endif
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 1; } } while (0)

__global__ void k_fill(float* p, int64_t n, float amp, uint32_t seed) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = amp * ((float)(x & 0xFFFF) / 32768.0f - 1.0f);
    }
}

__global__ void k_csum(const uint32_t* p, int64_t n, unsigned long long* out) {
    unsigned long long a = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        a += (unsigned long long)p[i] * (unsigned long long)(((uint32_t)i * 2654435761u) | 1u);
    atomicAdd(out, a);
}

int main(int argc, char** argv) {
    const int D = 100, H = 64;
    const int64_t nb = argc > 1 ? atoll(argv[1]) : 1000000;
    const int Rp = pk_rows(D);
    float *W21, *b21, *W22, *b22, *Wp, *bp, *WpT, *epsT, *gdT, *gxT, *hT, *slabs, *sc;
    uint32_t *gtab, *maxw; uint16_t* hs;
    const int T = D * (D + 1) / 2;
    CK(hipMalloc(&W21, D * 64 * 4)); CK(hipMalloc(&b21, D * 4)); CK(hipMalloc(&W22, (size_t)T * 64 * 4)); CK(hipMalloc(&b22, T * 4));
    CK(hipMalloc(&Wp, (size_t)Rp * 64 * 4)); CK(hipMalloc(&bp, Rp * 4)); CK(hipMalloc(&WpT, (size_t)Rp * 64 * 4)); CK(hipMalloc(&gtab, (Rp / 8 + 8) * 4));
    CK(hipMalloc(&epsT, nb * D * 4)); CK(hipMalloc(&gdT, nb * D * 4)); CK(hipMalloc(&gxT, nb * D * 4)); CK(hipMalloc(&hT, nb * 64 * 4));
    CK(hipMalloc(&hs, nb * 64 * 2 * 2)); CK(hipMalloc(&sc, 64)); CK(hipMalloc(&maxw, 16));
    k_fill<<<1024, 256>>>(W22, (int64_t)T * 64, 0.1f, 1); k_fill<<<64, 256>>>(W21, D * 64, 0.1f, 2);
    k_fill<<<4096, 256>>>(epsT, nb * D, 3.0f, 3); k_fill<<<4096, 256>>>(gdT, nb * D, 2.0f, 4); k_fill<<<4096, 256>>>(gxT, nb * D, 2.0f, 5);
    k_fill<<<4096, 256>>>(hT, nb * 64, 1.5f, 6);
    hipLaunchKernelGGL(k_pack_heads, dim3(Rp), dim3(64), 0, 0, D, 64, W21, b21, W22, b22, Wp, bp, gtab, WpT);
    float hsc[16] = {0}; hsc[3] = 1024.f;
    CK(hipMemcpy(sc, hsc, 64, hipMemcpyHostToDevice));
    float fm[4] = {2.0f, 2.0f, 3.0f, 0.f}; CK(hipMemcpy(maxw, fm, 16, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_split2_f16, dim3(2048), dim3(256), 0, 0, hT, nb * 64, sc + 3, hs);
    EncDims dm; dm.D = D; dm.J = 500; dm.H = 64; dm.Hp = 64; dm.DS = enc_ds(D); dm.T = T; dm.nb = nb;
    const int n_rowslabs = (Rp + BT_ROWS - 1) / BT_ROWS;
    int n_prw = argc > 2 ? atoi(argv[2]) : 256 / n_rowslabs;            // (fewer person slabs: the kernel on part of the chip)
    CK(hipMalloc(&slabs, (size_t)n_prw * Rp * 65 * 4));
#ifdef BWB_KERNEL_FILE
    const size_t lds = bb_lds_bytes(D);
    auto kern = k_mvn_enc_bwd_w_b;
#else
#ifndef BWB_NBUF
#define BWB_NBUF 3
#endif
    const size_t lds = BWB_NBUF == 3 ? bb_lds_bytes_n(D) : bb_lds_bytes(D);
    auto kern = k_mvn_enc_bwd_w_b<BWB_NBUF>;
#endif
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(n_rowslabs, n_prw), dim3(BWB_THREADS), lds, 0, dm, hs, epsT, gdT, gxT, gtab,
                           (const float*)sc, (const uint32_t*)maxw, slabs, (int64_t)Rp * 65);
        hipEventRecord(e1); CK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("k_mvn_enc_bwd_w_b nb=%lld grid %dx%d: %.3f ms\n", (long long)nb, n_rowslabs, n_prw, ms);
    }
    std::vector<float> out(8);
    CK(hipMemcpy(out.data(), slabs + 640, 32, hipMemcpyDeviceToHost));
    printf("check %g %g %g %g\n", out[0], out[1], out[2], out[3]);
    unsigned long long* cs; CK(hipMalloc(&cs, 8)); CK(hipMemset(cs, 0, 8));
    hipLaunchKernelGGL(k_csum, dim3(1024), dim3(256), 0, 0, (const uint32_t*)slabs, (int64_t)n_prw * Rp * 65, cs);
    unsigned long long cv = 0; CK(hipMemcpy(&cv, cs, 8, hipMemcpyDeviceToHost));
    printf("csum slabs %llx\n", cv);
    return 0;
}
