// fast_log (vx_common.h: ln x = ln 2 * v_log_f32 x, the 3PL / 4PL cell's log of the clamped probability) against
// double-precision log for probabilities close to 1, where log2 is small and a fixed absolute error would be a large relative one.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/fast_log_check tools/fast_log_check.hip && tools/fast_log_check
#include "../vipsy_amd/csrc/vx_common.h"
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float* x, float* y, float* y2, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { y[i] = fast_log(x[i]); y2[i] = __logf(x[i]); }
}
int main() {
    std::vector<float> xs;
    for (int e = 3; e <= 7; ++e)                                    // 1 - d, d from 1e-3 down to the clamp's 1.19e-7
        for (int m = 0; m < 4000; ++m) xs.push_back(1.0f - (float)((1.0 + m * 2.25e-3) * std::pow(10.0, -e)));
    for (int m = 0; m < 4000; ++m) xs.push_back(1.1920929e-07f * (1.0f + m * 1e-3f));          // and near the lower clamp
    for (int m = 1; m < 4000; ++m) xs.push_back(m * 2.5e-4f);                                   // the bulk of (0, 1)
    const int n = (int)xs.size();
    float *dx, *dy, *dz;
    hipMalloc(&dx, n * 4); hipMalloc(&dy, n * 4); hipMalloc(&dz, n * 4);
    hipMemcpy(dx, xs.data(), n * 4, hipMemcpyHostToDevice);
    k<<<(n + 255) / 256, 256>>>(dx, dy, dz, n);
    std::vector<float> y(n), z(n);
    hipMemcpy(y.data(), dy, n * 4, hipMemcpyDeviceToHost); hipMemcpy(z.data(), dz, n * 4, hipMemcpyDeviceToHost);
    double ma = 0, mr = 0, ma_near = 0, mr_near = 0, za = 0, zr_near = 0;
    for (int i = 0; i < n; ++i) {
        const double t = std::log((double)xs[i]), ea = std::fabs(y[i] - t), er = ea / std::fabs(t);
        const double ez = std::fabs(z[i] - t);
        ma = std::fmax(ma, ea); mr = std::fmax(mr, er); za = std::fmax(za, ez);
        if (xs[i] > 0.999f) { ma_near = std::fmax(ma_near, ea); mr_near = std::fmax(mr_near, er); zr_near = std::fmax(zr_near, ez / std::fabs(t)); }
    }
    printf("fast_log over %d points of (0, 1): max abs error %.3e, max rel error %.3e\n", n, ma, mr);
    printf("  for 1 - 1e-3 <= x <= 1 - 1.2e-7: max abs error %.3e, max rel error %.3e (library __logf there: max rel %.3e; abs over all %.3e)\n",
           ma_near, mr_near, zr_near, za);
    return 0;
}
