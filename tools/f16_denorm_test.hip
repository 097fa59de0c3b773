// Does v_mfma_f32_32x32x16_f16 keep fp16 subnormal inputs?  A = 2^-20 (subnormal), B = 2^10: expect 16 * 2^-10 per element.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float av, float bv) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)av; b[i] = (_Float16)bv; }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    out[threadIdx.x] = c[0];
}
int main() {
    float* d; hipMalloc(&d, 256);
    float tests[][2] = {{9.5367431640625e-07f, 1024.f}, {5.9604644775390625e-08f, 1024.f}, {6.103515625e-05f, 1024.f}, {1.f, 5.9604644775390625e-08f}};
    for (auto& t : tests) {
        k<<<1, 64>>>(d, t[0], t[1]);
        float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("a=%g b=%g -> %g (exact %g)\n", t[0], t[1], h, 16.0 * t[0] * t[1]);
    }
    return 0;
}
