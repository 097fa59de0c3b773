// fp32 product by three-way bf16 splitting on the bf16 MFMA (6 cross products, fp32 accumulate) against the exact
// fp32 MFMA: accuracy against an fp64 reference and cycles per 32 x 32 x 64 tile.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I vipsy_amd/csrc -o tools/bf16x3_ubench tools/bf16x3_ubench.hip
#include "vx_common.h"
#include <cstdio>
#include <cmath>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float v, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)v;
    const float r1 = v - (float)h;
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);
}

// MODE 0: fp32 MFMA (32 steps of K = 2);  MODE 1: bf16x3, 6 products x 4 chunks of K = 16
template <int MODE>
__global__ __launch_bounds__(64) void k(const float* __restrict__ A /*[32][64]*/, const float* __restrict__ B /*[64][32]*/,
                                        float* __restrict__ C /*[32][32]*/, int reps, float* __restrict__ sink) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    f32x16 acc = zero16();
    if (MODE == 0) {
        float a[32], b[32];
        for (int s = 0; s < 32; ++s) { a[s] = A[r * 64 + 2 * s + h]; b[s] = B[(2 * s + h) * 32 + r]; }
        for (int it = 0; it < reps; ++it) {
            if (it) acc = zero16();
#pragma unroll
            for (int s = 0; s < 32; ++s) acc = mfma32(a[s], b[s], acc);
            if (it + 1 < reps) sink[lane] += acc[0];
        }
    } else {
        bf16x8 ah[4], am[4], al[4], bh[4], bm[4], bl[4];
        for (int c = 0; c < 4; ++c)
            for (int j = 0; j < 8; ++j) {
                __bf16 x0, x1, x2;
                split3(A[r * 64 + 16 * c + 8 * h + j], x0, x1, x2);
                ah[c][j] = x0; am[c][j] = x1; al[c][j] = x2;
                split3(B[(16 * c + 8 * h + j) * 32 + r], x0, x1, x2);
                bh[c][j] = x0; bm[c][j] = x1; bl[c][j] = x2;
            }
        for (int it = 0; it < reps; ++it) {
            if (it) acc = zero16();
#pragma unroll
            for (int c = 0; c < 4; ++c) {                               // small terms first
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[c], bh[c], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[c], bl[c], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[c], bm[c], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[c], bh[c], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[c], bm[c], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[c], bh[c], acc, 0, 0, 0);
            }
            if (it + 1 < reps) sink[lane] += acc[0];
        }
    }
    for (int q = 0; q < 16; ++q) C[crow32(q, h) * 32 + r] = acc[q];
}

int main() {
    std::vector<float> hA(32 * 64), hB(64 * 32), hC(32 * 32);
    uint32_t s = 7;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 4.0f; };
    for (auto& v : hA) v = rnd();
    for (auto& v : hB) v = rnd() * 0.3f;
    float *A, *B, *C, *sink;
    hipMalloc(&A, hA.size() * 4); hipMalloc(&B, hB.size() * 4); hipMalloc(&C, hC.size() * 4); hipMalloc(&sink, 1024);
    hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, A, B, C, 1, sink);
        else hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, A, B, C, 1, sink);
        hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
        double maxerr = 0, maxabs = 0, sumabs = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double ref = 0, sa = 0;
                for (int kk = 0; kk < 64; ++kk) { ref += (double)hA[i * 64 + kk] * hB[kk * 32 + j]; sa += fabs((double)hA[i * 64 + kk] * hB[kk * 32 + j]); }
                maxerr = fmax(maxerr, fabs(hC[i * 32 + j] - ref) / sa);
                maxabs = fmax(maxabs, fabs(ref)); sumabs = fmax(sumabs, sa);
            }
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        const int reps = 20000;
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1024), dim3(64), 0, 0, A, B, C, reps, sink);
        else hipLaunchKernelGGL(k<1>, dim3(1024), dim3(64), 0, 0, A, B, C, reps, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: max |err| / sum|a b| = %.3g   cycles per 32x32x64 tile (2.4 GHz eq., 1 wave/SIMD) = %.0f\n",
               mode ? "bf16x3 (24 bf16 MFMAs)" : "fp32   (32 f32 MFMAs) ", maxerr, ms * 1e-3 * 2.4e9 / reps);
    }
    return 0;
}
