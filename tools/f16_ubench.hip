// fp16 MFMA on gfx950: (1) are subnormal fp16 inputs honoured or flushed?  (2) issue rate and clock against the bf16 form on
// random data.  hipcc --offload-arch=gfx950 -O3 -o tools/f16_ubench tools/f16_ubench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void k_sub(const float* av, const float* bv, float* out) {
    h8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = (_Float16)av[i]; B[i] = (_Float16)bv[i]; }
    f16v c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}

template <int F16>
__global__ __launch_bounds__(256) void k_rate(const float* src, float* out, int iters) {
    const int l = threadIdx.x;
    h8 A[4], B[4]; b8 Ab[4], Bb[4];
    for (int s = 0; s < 4; ++s)
        for (int i = 0; i < 8; ++i) {
            const float x = src[(l * 4 + s) * 8 + i], y = src[4096 + (l * 4 + s) * 8 + i];
            A[s][i] = (_Float16)x; B[s][i] = (_Float16)y; Ab[s][i] = (__bf16)x; Bb[s][i] = (__bf16)y;
        }
    f16v c0 = {0}, c1 = {0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (F16) { c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[s], B[s], c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[s], B[(s + 1) & 3], c1, 0, 0, 0); }
            else { c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ab[s], Bb[s], c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ab[s], Bb[(s + 1) & 3], c1, 0, 0, 0); }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
    out[blockIdx.x * 256 + l] = s;
}

int main() {
    float *a, *b, *o;
    hipMalloc(&a, 1 << 20); hipMalloc(&b, 1 << 20); hipMalloc(&o, 1 << 22);
    // (1) subnormals: A = 2^-20 (fp16 subnormal), B = 2^10 -> 16 products of 2^-10 = 2^-6 if honoured, 0 if flushed
    float ha[8], hb[8], r;
    for (int i = 0; i < 8; ++i) { ha[i] = ldexpf(1.f, -20); hb[i] = 1024.f; }
    hipMemcpy(a, ha, 32, hipMemcpyHostToDevice); hipMemcpy(b, hb, 32, hipMemcpyHostToDevice);
    k_sub<<<1, 64>>>(a, b, o); hipMemcpy(&r, o, 4, hipMemcpyDeviceToHost);
    printf("subnormal A (2^-20) x 2^10, K=16: got %g, honoured would be %g\n", r, 16 * ldexpf(1.f, -10));
    for (int i = 0; i < 8; ++i) { ha[i] = ldexpf(1.f, -24); hb[i] = 1.f; }
    hipMemcpy(a, ha, 32, hipMemcpyHostToDevice); hipMemcpy(b, hb, 32, hipMemcpyHostToDevice);
    k_sub<<<1, 64>>>(a, b, o); hipMemcpy(&r, o, 4, hipMemcpyDeviceToHost);
    printf("smallest subnormal 2^-24 x 1, K=16: got %g (honoured %g)\n", r, 16 * ldexpf(1.f, -24));
    // (2) rate
    float* hs = (float*)malloc(8192 * 4);
    for (int i = 0; i < 8192; ++i) hs[i] = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(a, hs, 8192 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 256;
    for (int rep = 0; rep < 2; ++rep)
        for (int f = 0; f < 2; ++f) {
            hipEventRecord(e0);
            if (f) k_rate<1><<<blocks, 256>>>(a, o, iters); else k_rate<0><<<blocks, 256>>>(a, o, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double mf = (double)blocks * 4 * iters * 8;
            printf("%s: %.3f ms, %.1f ns per MFMA per SIMD, %.0f TFLOP/s\n", f ? "f16 " : "bf16", ms, ms * 1e6 / (iters * 8.0), mf * 32768 / (ms * 1e-3) / 1e12);
        }
    return 0;
}
