// Stand-alone check + timing of k_irt_lik_b (bf16x3 likelihood kernel) against a double-precision CPU reference.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I vipsy_amd/csrc -o tools/likb_test tools/likb_test.hip
#include "k_irt_lik_b.hip"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static uint32_t rs = 12345;
static float rnd() { rs = rs * 1664525u + 1013904223u; return (rs >> 8) * (1.0f / 16777216.0f); }

static long long* g_stamps = nullptr;
template <int GEN, int ABL = 0>
static float run(int D, int J, int64_t nb, int model, int n_pr_req, bool check, int reps, float miss, int gxt) {
    LikBDims dm;
    dm.D = D; dm.J = J; dm.model = model; dm.groups = (J + LB_JC - 1) / LB_JC;
    const int64_t n_ptiles = (nb + LB_P - 1) / LB_P;
    dm.n_pr = (int)(n_ptiles < n_pr_req ? n_ptiles : n_pr_req);
    dm.gxt = gxt; dm.Dc = 1.3f; dm.scale = 2.5f; dm.nb = nb; dm.slab_len = (int64_t)D * J + 3 * J;
    uint8_t *y, *img; float *x, *a, *b, *c, *d, *gxp, *llp, *slabs, *gxT, *llo;
    const int64_t ystride = (nb + 63) / 64 * 64;
    CK(hipMalloc(&y, (size_t)(J + 1) * ystride)); CK(hipMalloc(&x, nb * D * 4)); CK(hipMalloc(&a, D * J * 4)); CK(hipMalloc(&b, J * 4));
    CK(hipMalloc(&c, J * 4)); CK(hipMalloc(&d, J * 4));
    const int64_t nbp = n_ptiles * LB_P;
    CK(hipMalloc(&gxp, (size_t)dm.groups * LB_DP * nbp * 4)); CK(hipMalloc(&llp, (size_t)dm.groups * nbp * 4));
    CK(hipMalloc(&slabs, (size_t)dm.n_pr * dm.slab_len * 4)); CK(hipMalloc(&img, (size_t)n_ptiles * LB_XT_BYTES));
    CK(hipMalloc(&gxT, (size_t)nb * D * 4)); CK(hipMalloc(&llo, nb * 4));
    std::vector<uint8_t> hy(nb * J);
    std::vector<float> hx(nb * D), ha(D * J), hb(J), hc(J), hd(J);
    for (auto& v : hy) { float u = rnd(); v = u < miss ? 255 : (rnd() < 0.5f ? 1 : 0); }
    for (auto& v : hx) v = 2.f * (rnd() - 0.5f);
    for (auto& v : ha) v = 0.5f * (rnd() - 0.5f);
    for (auto& v : hb) v = rnd() - 0.5f;
    for (auto& v : hc) v = -2.f + rnd();
    for (auto& v : hd) v = 2.f + rnd();
    {
        std::vector<uint8_t> hyT((size_t)(J + 1) * ystride, 254);
        for (int64_t i = 0; i < nb; ++i) for (int j = 0; j < J; ++j) hyT[(size_t)j * ystride + i] = hy[i * J + j];
        for (int64_t i = 0; i < ystride; ++i) hyT[(size_t)J * ystride + i] = 254;
        CK(hipMemcpy(y, hyT.data(), hyT.size(), hipMemcpyHostToDevice));
    }
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(a, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(c, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d, hd.data(), hd.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(slabs, 0, (size_t)dm.n_pr * dm.slab_len * 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_irt_lik_b<GEN, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, LB_LDS_BYTES));
    hipEvent_t e0, e1, e2;
    hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
    const dim3 grid(dm.groups * dm.n_pr);
    float ms_img = 0, ms = 0;
    for (int r = 0; r < reps + 1; ++r) {
        if (r == 1) hipEventRecord(e0);
        hipLaunchKernelGGL(k_lik_ximg, dim3((unsigned)n_ptiles), dim3(256), 0, 0, D, nb, x, img);
    }
    hipEventRecord(e1);
    for (int r = 0; r < reps + 1; ++r) {
        if (r == 1) hipEventRecord(e1);
        hipLaunchKernelGGL((k_irt_lik_b<GEN, ABL>), grid, dim3(LB_THREADS), LB_LDS_BYTES, 0, dm, y, ystride, img, a, b,
                           GEN ? c : nullptr, model == 4 ? d : nullptr, gxp, llp, slabs, g_stamps);
    }
    hipEventRecord(e2);
    CK(hipEventSynchronize(e2));
    CK(hipGetLastError());
    if (reps > 0) { hipEventElapsedTime(&ms, e1, e2); ms /= reps; }
    hipLaunchKernelGGL(k_lik_reduce_parts, dim3((unsigned)n_ptiles), dim3(256), (size_t)64 * (D | 1) * sizeof(float), 0, gxp, llp, x, dm.groups, D, nb, nbp, dm.scale, gxT, llo);
    CK(hipDeviceSynchronize());
    (void)ms_img;
    if (check) {
        std::vector<float> gg((size_t)nb * D), gl((size_t)nb), gs((size_t)dm.n_pr * dm.slab_len);
        CK(hipMemcpy(gg.data(), gxT, gg.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(gl.data(), llo, gl.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(gs.data(), slabs, gs.size() * 4, hipMemcpyDeviceToHost));
        std::vector<double> rll(nb, 0.0), rgx(nb * D, 0.0), rslab(dm.slab_len, 0.0);
        const double eps = 1.1920928955078125e-07;
        for (int64_t i = 0; i < nb; ++i) {
            double sq = 0;
            for (int k = 0; k < D; ++k) sq += (double)hx[i * D + k] * hx[i * D + k];
            rll[i] = -0.5 * sq;
            for (int k = 0; k < D; ++k) rgx[i * D + k] = -(double)dm.scale * hx[i * D + k];
            for (int j = 0; j < J; ++j) {
                const uint8_t yy = hy[i * J + j];
                if (yy == 255) { rll[i] += -1.1920928244535389e-07; continue; }
                double z = hb[j];
                for (int k = 0; k < D; ++k) z += (double)hx[i * D + k] * ha[k * J + j];
                z *= dm.Dc;
                const double sg = 1.0 / (1.0 + exp(-z));
                double cc = 0, dd = 1;
                if (GEN) { cc = 1.0 / (1.0 + exp(-(double)hc[j])); if (model == 4) dd = 1.0 / (1.0 + exp(-(double)hd[j])); }
                double P = cc + (dd - cc) * sg;
                const bool inside = P >= eps && P <= 1 - eps;
                double Pc = P < eps ? eps : (P > 1 - eps ? 1 - eps : P);
                rll[i] += yy ? log(Pc) : log(1 - Pc);
                double dP = inside ? ((double)yy - Pc) / (Pc * (1 - Pc)) : 0.0;
                double dz = dP * (dd - cc) * sg * (1 - sg);
                double R = dm.scale * dm.Dc * dz;
                for (int k = 0; k < D; ++k) { rgx[i * D + k] += R * ha[k * J + j]; rslab[(int64_t)k * J + j] += R * hx[i * D + k]; }
                rslab[(int64_t)D * J + j] += R;
                if (GEN) {
                    rslab[(int64_t)(D + 1) * J + j] += dm.scale * dP * (1 - sg) * cc * (1 - cc);
                    if (model == 4) rslab[(int64_t)(D + 2) * J + j] += dm.scale * dP * sg * dd * (1 - dd);
                }
            }
        }
        double ell = 0, mll = 0, egx = 0, mgx = 0, es = 0, msl = 0;
        for (int64_t i = 0; i < nb; ++i) {
            double v = gl[i];
            ell = fmax(ell, fabs(v - rll[i])); mll = fmax(mll, fabs(rll[i]));
            for (int k = 0; k < D; ++k) {
                double w = gg[(size_t)k * nb + i];
                egx = fmax(egx, fabs(w - rgx[i * D + k])); mgx = fmax(mgx, fabs(rgx[i * D + k]));
            }
        }
        int64_t worst = -1;
        for (int64_t e = 0; e < dm.slab_len; ++e) {
            double v = 0;
            for (int q = 0; q < dm.n_pr; ++q) v += gs[(size_t)q * dm.slab_len + e];
            if (fabs(v - rslab[e]) > es) { es = fabs(v - rslab[e]); worst = e; }
            msl = fmax(msl, fabs(rslab[e]));
        }
        printf("D=%d J=%d nb=%lld model=%d miss=%.2f gxt=%d: ll err %.3g (max %.3g)  gx err %.3g (max %.3g)  slab err %.3g (max %.3g, worst at row %lld col %lld)\n",
               D, J, (long long)nb, model, miss, gxt, ell, mll, egx, mgx, es, msl, (long long)(worst / J), (long long)(worst % J));
    }
    hipFree(y); hipFree(x); hipFree(a); hipFree(b); hipFree(c); hipFree(d); hipFree(gxp); hipFree(llp); hipFree(slabs); hipFree(img); hipFree(gxT); hipFree(llo);
    return ms;
}

int main(int argc, char** argv) {
    const bool timing = argc > 1 && argv[1][0] == 't';
    if (argc > 1 && argv[1][0] == 'p') {                    // profile mode: the full kernel only
        printf("k_irt_lik_b 1M x 500 x 100: %.3f ms\n", run<0, 0>(100, 500, 1000000, 2, 64, false, 3, 0.0f, 1));
        return 0;
    }
    run<0>(100, 500, 1000, 2, 8, true, 0, 0.0f, 1);
    run<0>(100, 500, 777, 2, 3, true, 0, 0.3f, 0);
    run<0>(101, 260, 300, 2, 2, true, 0, 0.1f, 1);
    run<1>(100, 500, 500, 4, 4, true, 0, 0.2f, 1);
    run<1>(108, 132, 200, 3, 2, true, 0, 0.0f, 0);
    if (timing) {
#define TM(ABL, what) printf("ABL=%2d %-40s %.3f ms\n", ABL, what, run<0, ABL>(100, 500, 1000000, 2, 64, false, 5, 0.0f, 1))
        CK(hipMalloc(&g_stamps, 1024 * 8 * 8)); CK(hipMemset(g_stamps, 0, 1024 * 8 * 8));
        TM(32, "full + stamps");
        {
            static long long hst[1024 * 8];
            CK(hipMemcpy(hst, g_stamps, sizeof(hst), hipMemcpyDeviceToHost));
            double acc[8] = {0}; int nblk = 0;
            for (int bq = 0; bq < 256; ++bq) { if (!hst[bq * 8 + 6]) continue; for (int q = 0; q < 8; ++q) acc[q] += (double)hst[bq * 8 + q]; ++nblk; }
            const double tiles = 1000000.0 / 64 / 64;     // tiles per workgroup
            const char* nm[8] = {"top barrier", "1 Z1 | gx1 store, ll0", "2 GA0,gx0 | cells1", "vmcnt wait", "mid barrier", "3 Z0' | ll1, gx0 store", "4 gx1,GA1 | cells0'", "-"};
            for (int q = 0; q < 7; ++q) printf("   %-24s %8.0f ticks / tile\n", nm[q], acc[q] / nblk / tiles);
        }
        TM(0, "full");
        TM(8, "no scheduling hints");
        TM(1, "no cell math");
        TM(2, "no barriers");
        TM(4, "no DMA");
        TM(64, "no x DMA");
        TM(128, "no y DMA");
        TM(16, "no output stores");
        TM(23, "no cells, barriers, DMA, stores");
        printf("3PL: %.3f ms\n", run<1, 0>(100, 500, 1000000, 3, 64, false, 5, 0.0f, 1));
    }
    return 0;
}
