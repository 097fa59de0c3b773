// Micro-benchmark: what the fp16 matrix pipe SUSTAINS on an MI355X at the board's power cap (docs/HARDWARE.md rule 41).
// Every case runs ~0.4 s (long enough for the power management to settle); the rate is the whole run's, and the clock it
// implies is rate / (the pipe's flops per cycle) for the cases that keep the pipe full.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/power_ubench tools/power_ubench.hip && tools/power_ubench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// MODE 0: dense MFMAs, eight rotating operand pairs of random fp16 values, four accumulator chains
// MODE 1: the same instruction stream on all-zero operands
// MODE 2: dense MFMAs with NV independent fp32 FMAs of a wave between two of them (the shipped kernels: 4-8 a MFMA)
// MODE 3: dense MFMAs, each followed by one 16-byte LDS read per lane
template <int MODE, int NV>
__global__ __launch_bounds__(256) void k_pw(float* out, int iters, uint32_t seed) {
    __shared__ uint32_t lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (i * 2654435761u + seed) | 0x04000400u;
    __syncthreads();
    h8 a[8], b[8];
    for (int u = 0; u < 8; ++u)
        for (int e = 0; e < 8; ++e) {
            uint32_t x = (threadIdx.x * 8 + e + 64 * u) * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
            const float fa = (float)(x & 0xFFFF) / 32768.0f - 1.0f, fb = (float)(x >> 16) / 32768.0f - 1.0f;
            a[u][e] = MODE == 1 ? (_Float16)0.f : (_Float16)fa;
            b[u][e] = MODE == 1 ? (_Float16)0.f : (_Float16)fb;
        }
    f32x16 acc[4];
    for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = 1.0f + 1e-3f * (float)(threadIdx.x + j);
    uint4 lv = make_uint4(0, 0, 0, 0);
    const uint4* lp = (const uint4*)lds + (threadIdx.x & 63);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u], b[u], acc[u & 3], 0, 0, 0);
            if (MODE == 2) {
#pragma unroll
                for (int j = 0; j < NV; ++j) v[j & 7] = __builtin_fmaf(v[j & 7], 0.999f, 1e-4f);
            }
            if (MODE == 3) {
                const uint4 t = lp[((u * 67 + it) & 31) * 64];
                lv.x ^= t.x; lv.y ^= t.y; lv.z ^= t.z; lv.w ^= t.w;
            }
        }
    }
    float s = 0.f;
    for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * 256 + threadIdx.x] = s + (float)(lv.x ^ lv.y ^ lv.z ^ lv.w);
}

template <int MODE, int NV>
static void run(const char* name, int blocks, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_pw<MODE, NV>), dim3(blocks), dim3(256), 0, 0, out, 2000, 1u);
    hipDeviceSynchronize();
    const int iters = 2500000 / (1 + (MODE == 2 ? NV / 4 : 0)) / (blocks > 256 ? blocks / 256 : 1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_pw<MODE, NV>), dim3(blocks), dim3(256), 0, 0, out, iters, 7u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)blocks * 4 * iters * 8;
    const double tflops = mfmas * 2 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
    // one MFMA per 32 cycles and SIMD when the pipe never waits: the clock a full pipe implies (only meaningful for 256 blocks
    // of the dense cases; with more blocks a CU the SIMD's waves share the pipe and the figure is the same)
    const double ghz = mfmas / 1024.0 * 32.0 / (ms * 1e-3) / 1e9;
    printf("%-52s blocks %4d: %7.1f ms  %7.1f TFLOP/s fp16 (%.3f of 2517)  pipe-full clock %.2f GHz\n", name, blocks, ms, tflops,
           tflops / 2517.0, ghz);
}

int main() {
    float* out; hipMalloc(&out, 2048 * 256 * sizeof(float));
    run<0, 0>("dense MFMA, random fp16 operands, 1 wave/SIMD", 256, out);
    run<0, 0>("dense MFMA, random fp16 operands, 2 waves/SIMD", 512, out);
    run<1, 0>("dense MFMA, zero operands, 1 wave/SIMD", 256, out);
    run<2, 4>("MFMA + 4 fp32 FMA each, 2 waves/SIMD", 512, out);
    run<2, 8>("MFMA + 8 fp32 FMA each, 2 waves/SIMD", 512, out);
    run<3, 0>("MFMA + one 16-byte LDS read each, 2 waves/SIMD", 512, out);
    run<0, 0>("dense MFMA, random operands, 64 CUs only", 64, out);
    return 0;
}
