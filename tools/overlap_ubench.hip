// Does VALU work issued by the SAME wavefront overlap with its in-flight fp32 MFMAs on gfx950?
//   hipcc --offload-arch=gfx950 -O3 -o tools/overlap_ubench tools/overlap_ubench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int NV>   // MODE 1: MFMA only, 2: VALU only, 3: both interleaved; NV fmas per MFMA
__global__ __launch_bounds__(256) void k(const float* __restrict__ in, float* __restrict__ out, int n) {
    f32x16 acc = {0};
    float a[4], v[16];
    for (int i = 0; i < 4; ++i) a[i] = in[threadIdx.x + 64 * i];
    for (int i = 0; i < 16; ++i) v[i] = in[1024 + threadIdx.x + 64 * i];
    const float c1 = in[0], c2 = in[1];
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (MODE & 1) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m & 3], a[(m + 1) & 3], acc, 0, 0, 0);
            if (MODE & 2) {
#pragma unroll
                for (int j = 0; j < NV; ++j) v[(m * NV + j) & 15] = fmaf(v[(m * NV + j) & 15], c1, c2);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i] + v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int NV>
float run(const float* in, float* out, int waves_per_simd) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 20000;
    const dim3 grid(256 * waves_per_simd);
    hipLaunchKernelGGL((k<MODE, NV>), grid, dim3(256), 0, 0, in, out, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NV>), grid, dim3(256), 0, 0, in, out, n);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3f * 2.4e9f / (n * 8.0f);      // cycles per (MFMA + NV VALU) step at 2.4 GHz
}

int main() {
    float *in, *out;
    hipMalloc(&in, 1 << 20); hipMalloc(&out, 4 << 20);
    hipMemset(in, 0, 1 << 20);
    for (int w = 1; w <= 2; ++w) {
        printf("waves/SIMD=%d  cycles per step: MFMA only %.1f\n", w, run<1, 8>(in, out, w));
#define ROW(NV) printf("  NV=%2d  VALU only %.1f   both %.1f\n", NV, run<2, NV>(in, out, w), run<3, NV>(in, out, w))
        ROW(4); ROW(8); ROW(12); ROW(16); ROW(24);
    }
    return 0;
}
