#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void k(const float* __restrict__ in, const unsigned char* __restrict__ yb, float* __restrict__ out, unsigned* __restrict__ yout) {
    __shared__ __attribute__((aligned(16))) float buf[64 * 108];
    __shared__ __attribute__((aligned(16))) unsigned ybuf[64 * 32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * 108; i += 256) buf[i] = -1.f;
    __syncthreads();
    for (int r = wave; r < 64; r += 4) {
        if (lane < 25)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(in + (size_t)r * 100 + 4 * lane),
                                             (__attribute__((address_space(3))) void*)(buf + r * 108), 16, 0, 0);
    }
    // y: 64 rows of 500 bytes, take bytes 128..255 of each row: one wave instruction = 8 rows x 8 lanes x 16 B
    for (int r8 = wave; r8 < 8; r8 += 4) {
        const int row = 8 * r8 + (lane >> 3);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(yb + (size_t)row * 500 + 128 + 16 * (lane & 7)),
                                         (__attribute__((address_space(3))) void*)(ybuf + r8 * 8 * 32), 16, 0, 0);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 108; i += 256) out[i] = buf[i];
    for (int i = threadIdx.x; i < 64 * 32; i += 256) yout[i] = ybuf[i];
}
int main() {
    std::vector<float> h(64 * 100 + 64);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)i;
    std::vector<unsigned char> hy(64 * 500 + 64);
    for (size_t i = 0; i < hy.size(); ++i) hy[i] = (unsigned char)(i * 7 + (i >> 8));
    float *d, *o; unsigned char* dy; unsigned* oy;
    hipMalloc(&d, h.size() * 4); hipMalloc(&o, 64 * 108 * 4); hipMalloc(&dy, hy.size()); hipMalloc(&oy, 64 * 32 * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dy, hy.data(), hy.size(), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, dy, o, oy);
    std::vector<float> r(64 * 108); std::vector<unsigned> ry(64 * 32);
    hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(ry.data(), oy, ry.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int p = 0; p < 64; ++p) for (int c = 0; c < 108; ++c) {
        float e = c < 100 ? (float)(p * 100 + c) : -1.f;
        if (r[p * 108 + c] != e) { if (bad < 5) printf("x mismatch p=%d c=%d got %f exp %f\n", p, c, r[p*108+c], e); ++bad; }
    }
    for (int p = 0; p < 64; ++p) for (int b = 0; b < 128; ++b) {
        unsigned char got = ((unsigned char*)ry.data())[p * 128 + b], e = hy[p * 500 + 128 + b];
        if (got != e) { if (bad < 10) printf("y mismatch p=%d b=%d got %u exp %u\n", p, b, got, e); ++bad; }
    }
    printf("bad=%d\n", bad);
    return 0;
}
