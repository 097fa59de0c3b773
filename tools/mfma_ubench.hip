// Micro-benchmark: what fp32 MFMA (v_mfma_f32_32x32x2_f32) issue patterns sustain on MI355X.
// hipcc --offload-arch=gfx950 -O3 -o mfma_ubench tools/mfma_ubench.hip && ./mfma_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)

template <int NACC, int LDSOPS>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = seed + i * 1e-6f;
    __syncthreads();
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    float a = seed + threadIdx.x, b = seed * 2 + threadIdx.x;
    const float* lp = lds + (threadIdx.x & 63);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            float av = a, bv = b;
            if (LDSOPS >= 1) av = lp[(u * 67 + it) & 4031];
            if (LDSOPS >= 2) bv = lp[(u * 131 + it * 3) & 4031];
            acc[u % NACC] = MF(av, bv, acc[u % NACC]);
        }
    }
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int LDSOPS>
void run(const char* name, int blocks, int threads, float* out) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, LDSOPS>), dim3(blocks), dim3(threads), 0, 0, out, 10, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, LDSOPS>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)blocks * (threads / 64) * iters * 16;
    const double tflops = mfmas * 2 * 32 * 32 * 2 / (ms * 1e-3) / 1e12;
    printf("%-44s blocks %4d thr %3d : %7.3f ms  %6.1f TFLOP/s  (%.1f%% of 157.3)\n", name, blocks, threads, ms, tflops,
           100 * tflops / 157.3);
}

int main() {
    float* out; hipMalloc(&out, 4096 * 256 * sizeof(float));
    run<1, 0>("1 chain, regs, 1 wave/SIMD", 256, 256, out);
    run<2, 0>("2 chains, regs, 1 wave/SIMD", 256, 256, out);
    run<4, 0>("4 chains, regs, 1 wave/SIMD", 256, 256, out);
    run<1, 0>("1 chain, regs, 2 waves/SIMD", 512, 256, out);
    run<1, 0>("1 chain, regs, 4 waves/SIMD", 1024, 256, out);
    run<1, 1>("1 chain, 1 LDS read/MFMA, 1 wave/SIMD", 256, 256, out);
    run<1, 2>("1 chain, 2 LDS reads/MFMA, 1 wave/SIMD", 256, 256, out);
    run<4, 2>("4 chains, 2 LDS reads/MFMA, 1 wave/SIMD", 256, 256, out);
    run<1, 2>("1 chain, 2 LDS reads/MFMA, 2 waves/SIMD", 512, 256, out);
    run<4, 2>("4 chains, 2 LDS reads/MFMA, 2 waves/SIMD", 512, 256, out);
    run<1, 2>("1 chain, 2 LDS reads/MFMA, 1.5 waves/SIMD(128thr)", 768, 128, out);
    return 0;
}
