// What does one "step" of k_mvn_enc_bwd_w_t cost?  8 MFMAs alternating two accumulators, A operands produced by 4
// multiplies of values just read from LDS (two ds_read_b128), reads of the next step issued before the MFMAs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I vipsy_amd/csrc -o tools/step_ubench tools/step_ubench.hip
#include "vx_common.h"
#include <cstdio>

template <int MODE>   // bit0: LDS reads, bit1: multiplies, bit2: single accumulator chain instead of two
__global__ __launch_bounds__(256) void k(const float* __restrict__ in, float* __restrict__ out, int n) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = in[i & 1023];
    __syncthreads();
    f32x16 acc0 = zero16(), acc1 = zero16();
    f32x4 g = *(const f32x4*)(lds + 4 * lane), e = *(const f32x4*)(lds + 1024 + 4 * lane);
    const f32x4 h0 = *(const f32x4*)(lds + 2048 + 4 * lane), h1 = *(const f32x4*)(lds + 3072 + 4 * lane);
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            f32x4 gn = g, en = e;
            if (MODE & 1) {
                gn = *(const f32x4*)(lds + 4 * lane + 256 * ((s + it) & 7));
                en = *(const f32x4*)(lds + 4096 + 4 * lane + 256 * ((s + it) & 7));
            }
            f32x4 v = g;
            if (MODE & 2) v = g * e;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc0 = mfma32(v[i], h0[i], acc0);
                if (MODE & 4) acc0 = mfma32(v[i], h1[i], acc0);
                else acc1 = mfma32(v[i], h1[i], acc1);
            }
            g = gn; e = en;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + g[0] + e[0];
}

template <int MODE>
float run(const float* in, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 4000;
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 32768, 0, in, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 32768, 0, in, out, n);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3f * 2.4e9f / (n * 8.0f);      // cycles per step (8 MFMAs) at 2.4 GHz
}

int main() {
    float *in, *out;
    hipMalloc(&in, 1 << 16); hipMalloc(&out, 1 << 20);
    float h[1024];
    for (int i = 0; i < 1024; ++i) h[i] = 0.001f * (i % 97) - 0.04f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    printf("cycles per 8-MFMA step (2.4 GHz equiv.; ideal 512)\n");
    printf("  two accumulators:   bare %.0f | +reads %.0f | +mul %.0f | +reads+mul %.0f\n", run<0>(in, out), run<1>(in, out), run<2>(in, out), run<3>(in, out));
    printf("  one accumulator:    bare %.0f | +reads %.0f | +mul %.0f | +reads+mul %.0f\n", run<4>(in, out), run<5>(in, out), run<6>(in, out), run<7>(in, out));
    return 0;
}
