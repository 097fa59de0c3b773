// Standalone timing harness for k_mvn_enc_fwd_b2 (compiles in ~30 s instead of the library's 2.5 min): synthetic operands at
// the headline shape, the library's own kernel sources and pack kernels.  -DFB2_STAMPS prints the per-phase cycle stamps.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -o tools/fwd2_bench tools/fwd2_bench.hip
#include "../vipsy_amd/csrc/vx_common.h"
#include "../vipsy_amd/csrc/k_mvn_enc.hip"
#include "../vipsy_amd/csrc/k_mvn_packed.hip"
#include "../vipsy_amd/csrc/k_irt_lik.hip"
#include "../vipsy_amd/csrc/k_irt_lik_r.hip"
#include "../vipsy_amd/csrc/k_irt_lik_b.hip"
#include "../vipsy_amd/csrc/k_irt_lik_h.hip"
#include "../vipsy_amd/csrc/k_mvn_bwd_t.hip"
#include "../vipsy_amd/csrc/k_mvn_bwd_b.hip"
#include "../vipsy_amd/csrc/k_mvn_fwd_b.hip"
#include "../vipsy_amd/csrc/k_mvn_fwd_b2.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 1; } } while (0)

__global__ void k_fill(float* p, int64_t n, float amp, uint32_t seed) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = amp * ((float)(x & 0xFFFF) / 32768.0f - 1.0f);
    }
}
__global__ void k_fill_y(uint8_t* p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u + 77u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (uint8_t)((x >> 7) & 1);
    }
}

__global__ void k_csum(const uint32_t* p, int64_t n, unsigned long long* out) {
    unsigned long long a = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        a += (unsigned long long)p[i] * (unsigned long long)(((uint32_t)i * 2654435761u) | 1u);
    atomicAdd(out, a);
}

int main(int argc, char** argv) {
    const int D = 100, H = 64, J = 500;
    const int64_t nb = argc > 1 ? atoll(argv[1]) : 983040;
    const int Rp = pk_rows(D), T = D * (D + 1) / 2;
    float *W1, *b1, *W21, *b21, *W22, *b22, *Wp, *bp, *WpT, *sc, *h, *x, *eps, *ldT, *ent, *hT, *epsT;
    uint32_t *gtab, *gt2; uint8_t *y, *img, *w1img, *ximg; uint16_t* hs;
    CK(hipMalloc(&W1, H * J * 4)); CK(hipMalloc(&b1, H * 4)); CK(hipMalloc(&W21, D * 64 * 4)); CK(hipMalloc(&b21, D * 4));
    CK(hipMalloc(&W22, (size_t)T * 64 * 4)); CK(hipMalloc(&b22, T * 4));
    CK(hipMalloc(&Wp, (size_t)Rp * 64 * 4)); CK(hipMalloc(&bp, Rp * 4)); CK(hipMalloc(&WpT, (size_t)Rp * 64 * 4)); CK(hipMalloc(&gtab, (Rp / 8 + 8) * 4));
    CK(hipMalloc(&sc, 64)); CK(hipMalloc(&y, nb * J));
    const int n_tiles = fb_tiles(D);
    CK(hipMalloc(&img, fb_img_floats(D) * 4)); gt2 = (uint32_t*)(img + (int64_t)n_tiles * FB_IMG_BYTES);
    CK(hipMalloc(&w1img, fb_w1img_floats(J) * 4));
    CK(hipMalloc(&h, nb * 64 * 4)); CK(hipMalloc(&x, nb * D * 4)); CK(hipMalloc(&eps, nb * D * 4)); CK(hipMalloc(&ldT, nb * D * 4));
    CK(hipMalloc(&ent, nb * 4)); CK(hipMalloc(&hT, nb * 64 * 4)); CK(hipMalloc(&epsT, nb * D * 4)); CK(hipMalloc(&hs, nb * 64 * 4));
    CK(hipMalloc(&ximg, (size_t)((nb + 63) / 64) * LB_XT_BYTES));
    k_fill<<<256, 256>>>(W1, H * J, 0.045f, 1); k_fill<<<1, 64>>>(b1, H, 0.045f, 2);
    k_fill<<<64, 256>>>(W21, D * 64, 0.125f, 3); k_fill<<<1, 128>>>(b21, D, 0.125f, 4);
    k_fill<<<1024, 256>>>(W22, (int64_t)T * 64, 0.125f, 5); k_fill<<<32, 256>>>(b22, T, 0.125f, 6);
    k_fill_y<<<4096, 256>>>(y, nb * J);
    hipLaunchKernelGGL(k_pack_heads, dim3(Rp), dim3(64), 0, 0, D, 64, W21, b21, W22, b22, Wp, bp, gtab, WpT);
    CK(hipMemset(sc, 0, 64));
    hipLaunchKernelGGL(k_enc_scales_max, dim3(FB_SC_BLOCKS), dim3(256), 0, 0, D, J, W1, b1, W21, b21, W22, b22, sc);
    hipLaunchKernelGGL(k_enc_scales, dim3(1), dim3(64), 0, 0, sc);
    hipLaunchKernelGGL(k_pack_w1_b, dim3((J + 15) / 16), dim3(256), 0, 0, J, W1, (const float*)sc, w1img);
    hipLaunchKernelGGL(k_pack_heads_b, dim3(n_tiles), dim3(256), 0, 0, n_tiles, pk_off_total(D) / 8, Wp, bp, gtab, (const float*)sc, img, gt2);
    EncDims dm; dm.D = D; dm.J = J; dm.H = 64; dm.Hp = 64; dm.DS = enc_ds(D); dm.T = T; dm.nb = nb;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned long long* cs; CK(hipMalloc(&cs, 8));
    auto csum = [&](const void* p, size_t bytes) -> unsigned long long {
        hipMemset(cs, 0, 8);
        hipLaunchKernelGGL(k_csum, dim3(2048), dim3(256), 0, 0, (const uint32_t*)p, (int64_t)(bytes / 4), cs);
        unsigned long long v = 0; hipMemcpy(&v, cs, 8, hipMemcpyDeviceToHost); return v;
    };
    float* eps_given = nullptr;
    if (getenv("FWD2_EPS_IN")) { CK(hipMalloc(&eps_given, nb * D * 4)); k_fill<<<1024, 256>>>(eps_given, nb * D, 1.0f, 9); }
    auto run = [&](auto nsc, auto w3c) -> int {
        constexpr int NS = decltype(nsc)::value;
        constexpr bool W3 = decltype(w3c)::value;
        const size_t lds = W3 ? fb2s_lds_bytes(D) : fb2_lds_bytes(D, J, NS);
        if (W3 && !fb2s_shape_ok(D, J)) { printf("SH shape not ok\n"); return 1; }
        CK(hipFuncSetAttribute((const void*)k_mvn_enc_fwd_b2<NS, W3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipMemset(h, 0, nb * 64 * 4); hipMemset(x, 0, nb * D * 4); hipMemset(eps, 0, nb * D * 4); hipMemset(ldT, 0, nb * D * 4);
        hipMemset(ent, 0, nb * 4); hipMemset(hT, 0, nb * 64 * 4); hipMemset(epsT, 0, nb * D * 4); hipMemset(hs, 0, nb * 64 * 4);
        hipMemset(ximg, 0, (size_t)((nb + 63) / 64) * LB_XT_BYTES);
        const int wg = FB2_WAVES * 32 * NS;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((k_mvn_enc_fwd_b2<NS, W3>), dim3((unsigned)((nb + wg - 1) / wg)), dim3(FB2_THREADS), lds, 0, dm, (const uint8_t*)y,
                               (const int64_t*)nullptr, (int64_t)0, (const uint8_t*)w1img, (const float*)b1, (const uint8_t*)img,
                               (const uint32_t*)gt2, (const float*)sc, (const float*)eps_given, (uint64_t)1234, 0u, (const uint32_t*)nullptr, 0u, h, x, eps, ldT, ent, hT,
                               epsT, ximg, hs);
            hipEventRecord(e1); CK(hipEventSynchronize(e1));
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("k_mvn_enc_fwd_b2<%d,%d> nb=%lld%s: %.3f ms (lds %zu)\n", NS, (int)W3, (long long)nb, eps_given ? " eps given" : "", ms, lds);
        }
        std::vector<float> o(4);
        CK(hipMemcpy(o.data(), x + 1000, 16, hipMemcpyDeviceToHost));
        float hv, ev, lv; uint16_t hsv; uint8_t xb;
        CK(hipMemcpy(&hv, hT + 3 * nb + 777, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&ev, epsT + 5 * nb + 4321, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&lv, ldT + 7 * nb + 99, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hsv, hs + 64 * nb + 2 * nb + 55, 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&xb, ximg + 123457, 1, hipMemcpyDeviceToHost));
        printf("check x %g %g %g %g  hT %g epsT %g ldT %g hs %u ximg %u\n", o[0], o[1], o[2], o[3], hv, ev, lv, (unsigned)hsv, (unsigned)xb);
        printf("csum h %llx x %llx eps %llx ldT %llx ent %llx hT %llx epsT %llx hs %llx ximg %llx\n", csum(h, nb * 64 * 4), csum(x, nb * D * 4),
               csum(eps, nb * D * 4), csum(ldT, nb * D * 4), csum(ent, nb * 4), csum(hT, nb * 64 * 4), csum(epsT, nb * D * 4), csum(hs, nb * 64 * 4),
               csum(ximg, (size_t)((nb + 63) / 64) * LB_XT_BYTES));
        return 0;
    };
    if (argc > 2) {                                            // small batch: the SPLIT kernel of k_mvn_fwd_b.hip, gathered rows
        int64_t* rows; CK(hipMalloc(&rows, nb * 8));
        std::vector<int64_t> hr(nb); for (int64_t i = 0; i < nb; ++i) hr[i] = (i * 7919 + 13) % nb;
        CK(hipMemcpy(rows, hr.data(), nb * 8, hipMemcpyHostToDevice));
        const size_t ldsb = fb_lds_bytes(D, J);
        CK(hipFuncSetAttribute((const void*)k_mvn_enc_fwd_b<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
        const unsigned gs = (unsigned)(((nb + 63) / 64) * 2);
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_mvn_enc_fwd_b<true>, dim3(gs), dim3(FB_THREADS), ldsb, 0, dm, (const uint8_t*)y, (const int64_t*)(argv[2][0] == 'd' ? nullptr : rows), (int64_t)0,
                               (const uint8_t*)w1img, (const float*)b1, (const uint8_t*)img, (const uint32_t*)gt2, (const float*)sc, (const float*)nullptr,
                               (uint64_t)1234, 0u, (const uint32_t*)nullptr, 0u, h, x, eps, ldT, ent, hT, epsT, ximg, hs, (int64_t)0);
            hipEventRecord(e1); CK(hipEventSynchronize(e1));
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("k_mvn_enc_fwd_b<true> nb=%lld grid %u: %.1f us\n", (long long)nb, gs, 1000.f * ms);
        }
        return 0;
    }
    if (run(std::integral_constant<int, 1>{}, std::false_type{})) return 1;
    if (run(std::integral_constant<int, 1>{}, std::true_type{})) return 1;
    return 0;
}
