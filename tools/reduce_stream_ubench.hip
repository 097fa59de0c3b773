// Would k_lik_reduce_parts be faster as a pure streaming kernel?  Today it turns person-major x through LDS (the prior term) and
// reads its eight dimension-major streams in 256-byte pieces of 100 rows (64 persons a block).  This harness times the same
// arithmetic with x ALSO dimension-major: seven input streams, two output streams, 16 bytes a lane, whole rows contiguous.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/reduce_stream_ubench tools/reduce_stream_ubench.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL>
__global__ __launch_bounds__(256) void k_stream(const f32x4* __restrict__ p0, const f32x4* __restrict__ p1, const f32x4* __restrict__ p2,
                                                const f32x4* __restrict__ p3, const f32x4* __restrict__ xT, const f32x4* __restrict__ epsT,
                                                const f32x4* __restrict__ ldT, float scale, int64_t n4, f32x4* __restrict__ gxT,
                                                f32x4* __restrict__ gdT, uint32_t* __restrict__ opmax) {
    float mg = 0.f, md = 0.f, me = 0.f;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride * UNROLL) {
        f32x4 a[UNROLL], x[UNROLL], e[UNROLL], l[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int64_t j = i + u * stride;
            if (j < n4) {
                a[u] = __builtin_nontemporal_load(p0 + j) + __builtin_nontemporal_load(p1 + j) + __builtin_nontemporal_load(p2 + j) +
                       __builtin_nontemporal_load(p3 + j);
                x[u] = __builtin_nontemporal_load(xT + j); e[u] = epsT[j]; l[u] = __builtin_nontemporal_load(ldT + j);
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int64_t j = i + u * stride;
            if (j < n4) {
                f32x4 g, d;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    g[c] = __builtin_fmaf(-scale, x[u][c], a[u][c]);
                    d[c] = __builtin_fmaf(g[c] * e[u][c], l[u][c], scale);
                    mg = fmaxf(mg, fabsf(g[c])); md = fmaxf(md, fabsf(d[c])); me = fmaxf(me, fabsf(e[u][c]));
                }
                gxT[j] = g; gdT[j] = d;
            }
        }
    }
    for (int o = 32; o; o >>= 1) { mg = fmaxf(mg, __shfl_xor(mg, o)); md = fmaxf(md, __shfl_xor(md, o)); me = fmaxf(me, __shfl_xor(me, o)); }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(opmax, __float_as_uint(mg)); atomicMax(opmax + 1, __float_as_uint(md)); atomicMax(opmax + 2, __float_as_uint(me));
    }
}

__global__ void k_fill(float* p, int64_t n, uint32_t seed) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (float)(x & 0xFFFF) / 32768.0f - 1.0f;
    }
}

template <int UNROLL>
static void run(int blocks, float** b, int64_t n, uint32_t* opmax) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_stream<UNROLL>), dim3(blocks), dim3(256), 0, 0, (const f32x4*)b[0], (const f32x4*)b[1], (const f32x4*)b[2],
                           (const f32x4*)b[3], (const f32x4*)b[4], (const f32x4*)b[5], (const f32x4*)b[6], 1.0f, n / 4, (f32x4*)b[7],
                           (f32x4*)b[8], opmax);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    printf("unroll %d blocks %5d: %.3f ms  %.2f TB/s (9 streams x %.2f GB)\n", UNROLL, blocks, best, 9.0 * n * 4 / (best * 1e-3) / 1e12, n * 4 / 1e9);
}

int main(int argc, char** argv) {
    const int64_t nb = argc > 1 ? atoll(argv[1]) : 1000000, D = 100, n = nb * D;
    float* b[9];
    for (int i = 0; i < 9; ++i) { hipMalloc(&b[i], n * 4); if (i < 7) k_fill<<<4096, 256>>>(b[i], n, 17u * i + 1); }
    uint32_t* opmax; hipMalloc(&opmax, 16); hipMemset(opmax, 0, 16);
    hipDeviceSynchronize();
    for (int blocks : {1024, 2048, 4096, 8192}) { run<1>(blocks, b, n, opmax); run<2>(blocks, b, n, opmax); run<4>(blocks, b, n, opmax); }
    return 0;
}
