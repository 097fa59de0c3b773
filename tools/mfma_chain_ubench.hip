// Micro-benchmark: does a DEPENDENT chain of v_mfma_f32_32x32x16_bf16 (each accumulating into the previous result) issue
// back to back, or does it pay a bubble per MFMA that independent accumulator chains avoid?
// hipcc --offload-arch=gfx950 -O3 -o mfma_chain_ubench tools/mfma_chain_ubench.hip && ./mfma_chain_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(seed + threadIdx.x * 0.001f + j); b[j] = (__bf16)(seed * 0.5f + j); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 24; ++u) acc[u % NACC] = MF(a, b, acc[u % NACC]);
    }
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(const char* name, int blocks, float* out) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC>), dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)iters * 24 * (blocks / 256.0);       // MFMAs a SIMD executes (blocks = CUs x waves/SIMD)
    printf("%-40s : %7.3f ms  -> %.1f ns per MFMA per SIMD (%.1f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / per_simd,
           ms * 1e6 / per_simd * 2.4);
}

int main() {
    float* out; hipMalloc(&out, 1024 * 256 * 4);
    run<1>("1 chain, 1 wave/SIMD", 256, out);
    run<2>("2 chains, 1 wave/SIMD", 256, out);
    run<3>("3 chains, 1 wave/SIMD", 256, out);
    run<4>("4 chains, 1 wave/SIMD", 256, out);
    run<1>("1 chain, 2 waves/SIMD", 512, out);
    run<2>("2 chains, 2 waves/SIMD", 512, out);
    return 0;
}
