// Standalone timing harness for k_fc1_bwd_c<F16, TILED>: the responses item-major ([J + 1][stride]: 64-byte pieces of rows a
// whole batch apart) against tile-major ([chunk of 64 persons][J][64 bytes]: 32 KB contiguous a workgroup and chunk), 1M x 500.
// The slabs of the two are compared bit for bit.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/fc1c_bench tools/fc1c_bench.hip
#include "../vipsy_amd/csrc/vx_common.h"
#include "../vipsy_amd/csrc/k_util.hip"
#include "../vipsy_amd/csrc/k_mvn_enc.hip"
#include "../vipsy_amd/csrc/k_mvn_packed.hip"
#include "../vipsy_amd/csrc/k_irt_lik.hip"
#include "../vipsy_amd/csrc/k_irt_lik_r.hip"
#include "../vipsy_amd/csrc/k_irt_lik_b.hip"
#include "../vipsy_amd/csrc/k_irt_lik_h.hip"
#include "../vipsy_amd/csrc/k_mvn_bwd_t.hip"
#include "../vipsy_amd/csrc/k_mvn_bwd_b.hip"
#include "../vipsy_amd/csrc/k_mvn_fwd_b.hip"
#include "../vipsy_amd/csrc/k_fc1_bwd_c.hip"
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 1; } } while (0)

__global__ void k_fill(float* p, int64_t n, float amp, uint32_t seed) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = amp * ((float)(x & 0xFFFF) / 32768.0f - 1.0f);
    }
}
// yT[j][i] item-major and its tile-major copy, the same bytes
__global__ void k_fill_y(uint8_t* yT, uint8_t* yt, int J, int64_t nb, int64_t stride) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < (int64_t)J * stride; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = e / stride, i = e - j * stride;
        uint32_t x = (uint32_t)e * 2654435761u + 77u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        const uint8_t v = i < nb ? ((x % 10u) ? (uint8_t)255 : (uint8_t)((x >> 7) & 1)) : (uint8_t)254;
        yT[e] = v;
        yt[((i >> 6) * J + j) * 64 + (i & 63)] = v;
    }
}
__global__ void k_csum(const uint32_t* p, int64_t n, unsigned long long* out) {
    unsigned long long a = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        a += (unsigned long long)p[i] * (unsigned long long)(((uint32_t)i * 2654435761u) | 1u);
    atomicAdd(out, a);
}

int main(int argc, char** argv) {
    const int J = argc > 2 ? atoi(argv[2]) : 500, H = 64;
    const int64_t nb = argc > 1 ? atoll(argv[1]) : 1000000;
    const int64_t stride = (nb + 63) / 64 * 64, n_ch = stride / 64;
    const int n_prf = 256;                                              // (vx_abi.hip::encb_plan: one workgroup a CU at J <= 511)
    const int64_t lenf = (int64_t)H * J + H;
    uint8_t *yT, *yt; float *g, *slabs; uint32_t* maxw; unsigned long long* cs;
    CK(hipMalloc(&yT, (size_t)(J + 1) * stride)); CK(hipMalloc(&yt, (size_t)n_ch * J * 64));
    CK(hipMalloc(&g, (size_t)64 * nb * 4)); CK(hipMalloc(&slabs, (size_t)n_prf * lenf * 4)); CK(hipMalloc(&maxw, 16)); CK(hipMalloc(&cs, 8));
    k_fill<<<2048, 256>>>(g, 64 * nb, 0.8f, 5);
    k_fill_y<<<4096, 256>>>(yT, yt, J, nb, stride);
    const float mx = 0.8f;
    uint32_t hm[4] = {0, 0, 0, __builtin_bit_cast(uint32_t, mx)};
    CK(hipMemcpy(maxw, hm, 16, hipMemcpyHostToDevice));
    EncDims dm; dm.D = 1; dm.J = J; dm.H = 64; dm.Hp = 64; dm.DS = 3; dm.T = 1; dm.nb = nb;
    const dim3 grid((unsigned)((J + 1 + 511) / 512), (unsigned)n_prf);
    CK(hipFuncSetAttribute((const void*)k_fc1_bwd_c<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)f1c_lds_bytes()));
    CK(hipFuncSetAttribute((const void*)k_fc1_bwd_c<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)f1c_lds_bytes()));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto csum = [&]() -> unsigned long long {
        hipMemset(cs, 0, 8);
        hipLaunchKernelGGL(k_csum, dim3(1024), dim3(256), 0, 0, (const uint32_t*)slabs, (int64_t)n_prf * lenf, cs);
        unsigned long long v = 0; hipMemcpy(&v, cs, 8, hipMemcpyDeviceToHost); return v;
    };
    for (int tiled = 0; tiled < 2; ++tiled) {
        auto launch = [&]() {
            if (tiled) hipLaunchKernelGGL((k_fc1_bwd_c<true, true>), grid, dim3(F1C_THREADS), f1c_lds_bytes(), 0, dm, (const uint8_t*)yt, stride, (const float*)g, slabs, lenf, (const uint32_t*)maxw);
            else hipLaunchKernelGGL((k_fc1_bwd_c<true, false>), grid, dim3(F1C_THREADS), f1c_lds_bytes(), 0, dm, (const uint8_t*)yT, stride, (const float*)g, slabs, lenf, (const uint32_t*)maxw);
        };
        CK(hipMemset(slabs, 0, (size_t)n_prf * lenf * 4));
        for (int i = 0; i < 3; ++i) launch();
        CK(hipDeviceSynchronize());
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(e1); CK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-12s %8.1f us a launch (%.2f TB/s of %.2f GB), slab checksum %016llx\n", tiled ? "tile-major" : "item-major", 1e3 * ms / 20,
               ((double)J * nb + 256.0 * nb) / (ms / 20 * 1e-3) / 1e12, ((double)J * nb + 256.0 * nb) / 1e9, csum());
    }
    return 0;
}
