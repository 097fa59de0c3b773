// Issue cost of gfx950 global->LDS DMA (global_load_lds_dwordx4) next to an fp32-MFMA stream.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I vipsy_amd/csrc -o tools/dma_ubench tools/dma_ubench.hip
#include "vx_common.h"
#include <cstdio>

// per iteration: 64 MFMAs and NDMA DMA instructions; MODE bit0: change M0 for every DMA; bit1: only 25 lanes active;
// bit2: 4-byte-aligned (not 16) global addresses; bit3: plain global_load_dwordx4 into registers instead of DMA
template <int NDMA, int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ in, float* __restrict__ out, int n, int64_t stride) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x16 acc = zero16();
    float a0 = in[lane], a1 = in[64 + lane];
    float4 sink = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* base = in + (int64_t)(blockIdx.x * 4 + wave) * stride + ((MODE & 4) ? 1 : 0);
    for (int it = 0; it < n; ++it) {
        const float* src = base + (int64_t)(it & 63) * 4096 + 4 * lane;
#pragma unroll
        for (int d = 0; d < NDMA; ++d) {
            if (!(MODE & 2) || lane < 25) {
                if (MODE & 8) {
                    const float4 v = *(const float4*)(src + d * 256);
                    sink.x += v.x; sink.y += v.y; sink.z += v.z; sink.w += v.w;
                } else {
                    dma16(src + d * 256, lds_addr_uniform(lds + wave * 4096 + ((MODE & 1) ? d * 256 : 0)));
                }
            }
        }
#pragma unroll
        for (int m = 0; m < 64; ++m) acc = mfma32(a0, a1, acc);
        if ((it & 7) == 7) vx_wait_vmem();
    }
    vx_wait_vmem();
    __syncthreads();
    float s = lds[threadIdx.x] + sink.x + sink.y + sink.z + sink.w;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NDMA, int MODE>
float run(const float* in, float* out, int64_t stride) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 2000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<NDMA, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL((k<NDMA, MODE>), dim3(256), dim3(256), 65536, 0, in, out, 10, stride);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NDMA, MODE>), dim3(256), dim3(256), 65536, 0, in, out, n, stride);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3f * 2.4e9f / n;      // cycles per iteration at 2.4 GHz
}

int main() {
    float *in, *out;
    const int64_t stride = 64 * 4096 + 64;
    hipMalloc(&in, (size_t)1024 * stride * 4 + 65536); hipMalloc(&out, 1 << 20);
    hipMemset(in, 0, (size_t)1024 * stride * 4 + 65536);
    const float base = run<0, 0>(in, out, stride);
    printf("64 MFMAs alone: %.0f cycles/iter\n", base);
#define ROW(N) printf("NDMA=%2d: same-M0 %.0f | new-M0 %.0f | 25 lanes %.0f | 4B-aligned %.0f | plain loads %.0f   (extra cycles per DMA over MFMA-only)\n", N, \
    (run<N, 0>(in, out, stride) - base) / N, (run<N, 1>(in, out, stride) - base) / N, (run<N, 3>(in, out, stride) - base) / N, \
    (run<N, 5>(in, out, stride) - base) / N, (run<N, 8>(in, out, stride) - base) / N)
    ROW(2); ROW(8); ROW(16);
    return 0;
}
