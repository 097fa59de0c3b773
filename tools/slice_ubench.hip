// What does one vector instruction cost in the shadow of a bf16 MFMA (v_mfma_f32_32x32x16_bf16, 8 passes) on gfx950?
// One wave per SIMD; per step: one MFMA (two alternating accumulators) + NV instructions of one kind.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/slice_ubench tools/slice_ubench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// KIND 0: none, 1: v_cvt_pk_bf16_f32, 2: v_pk_add_f32, 3: v_pk_mul_f32, 4: v_lshlrev_b32, 5: v_and_b32, 6: v_add_f32,
//      7: v_perm_b32
template <int KIND, int NV, bool MF>
__global__ __launch_bounds__(256) void k(const float* __restrict__ in, float* __restrict__ out, int n) {
    f32x16 acc0 = {0}, acc1 = {0};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)in[threadIdx.x + i]; b[i] = (__bf16)in[64 + threadIdx.x + i]; }
    f32x2 v[8];
    uint32_t w[8];
    for (int i = 0; i < 8; ++i) { v[i] = f32x2{in[128 + i], in[256 + i]}; w[i] = threadIdx.x + i; }
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (MF) {
                if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int q = (m * NV + j) & 7;
                if (KIND == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[q]) : "v"(v[q].x), "v"(v[q].y));
                if (KIND == 2) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(v[q]) : "v"(v[q]), "v"(v[(q + 3) & 7]));
                if (KIND == 3) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(v[q]) : "v"(v[q]), "v"(v[(q + 3) & 7]));
                if (KIND == 4) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(w[q]) : "v"(w[(q + 3) & 7]));
                if (KIND == 5) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(w[q]) : "v"(w[(q + 3) & 7]));
                if (KIND == 6) asm volatile("v_add_f32 %0, %1, %2" : "=v"(v[q].x) : "v"(v[q].y), "v"(v[(q + 3) & 7].x));
                if (KIND == 7) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(w[q]) : "v"(w[(q + 1) & 7]), "v"(w[(q + 3) & 7]), "v"(w[(q + 5) & 7]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y + (float)w[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND, int NV, bool MF>
float run(const float* in, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 20000;
    hipLaunchKernelGGL((k<KIND, NV, MF>), dim3(256), dim3(256), 0, 0, in, out, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, NV, MF>), dim3(256), dim3(256), 0, 0, in, out, n);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3f * 2.4e9f / (n * 8.0f);      // cycles per step at 2.4 GHz
}

int main() {
    float *in, *out;
    hipMalloc(&in, 1 << 20); hipMalloc(&out, 4 << 20);
    hipMemset(in, 0, 1 << 20);
    printf("MFMA only: %.1f cycles per step (at 2.4 GHz)\n", run<0, 0, true>(in, out));
    const char* names[] = {"", "v_cvt_pk_bf16_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_lshlrev_b32", "v_and_b32", "v_add_f32", "v_perm_b32"};
#define ROW(K) printf("%-18s  alone x6 %.1f  |  with MFMA: x2 %.1f  x4 %.1f  x6 %.1f  x8 %.1f\n", names[K], run<K, 6, false>(in, out), \
                      run<K, 2, true>(in, out), run<K, 4, true>(in, out), run<K, 6, true>(in, out), run<K, 8, true>(in, out))
    ROW(1); ROW(2); ROW(3); ROW(4); ROW(5); ROW(6); ROW(7);
    return 0;
}
