// D = 1 IRT (irt_1pl..4pl with a Normal guide; vi.py:22-66, 588-595, 617-625, 684/705), one fused pass:
// x = loc + exp(raw) eps -> masked Bernoulli log-lik -> prior/entropy -> gradients.
// Layout: item-per-lane.  Each lane keeps the parameters and gradient accumulators of items
// lane, lane+64, ... in registers; a wave walks its persons, the response row is read coalesced
// (one byte per lane per 64-item slice), and only two values per person cross lanes.
#pragma once
#include "vx_common.h"

#define I1_THREADS 256

struct Irt1dDims {
    int J, model;
    float Dc, scale;
    int64_t nb;
};

template <int MODEL, int IPL>
__global__ __launch_bounds__(I1_THREADS) void k_irt1d(
    Irt1dDims dm, const uint8_t* __restrict__ y, const int64_t* __restrict__ rows, int64_t gid0,
    const float* __restrict__ loc, const float* __restrict__ raw, const float* __restrict__ eps_in,
    uint64_t seed, uint32_t step, uint32_t stream, const float* __restrict__ a, const float* __restrict__ b,
    const float* __restrict__ c_un, const float* __restrict__ d_un, float* __restrict__ gloc,
    float* __restrict__ graw, float* __restrict__ elbo, float* __restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [4*J] block-level item-grad reduction
    const int J = dm.J;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t n_waves = (int64_t)gridDim.x * (I1_THREADS / 64);
    const int64_t wg = (int64_t)blockIdx.x * (I1_THREADS / 64) + wave;
    // persons are handed out in groups of 64 so that lane l owns person g*64 + l of each group
    const int64_t n_groups = (dm.nb + 63) / 64;
    float aq[IPL], bq[IPL], cq[IPL], dq[IPL], oq[IPL], ga[IPL], gb[IPL], gc[IPL], gd[IPL];
#pragma unroll
    for (int q = 0; q < IPL; ++q) {
        const int j = lane + 64 * q;
        const bool ok = j < J;
        aq[q] = (MODEL >= 2) ? (ok ? a[j] : 0.f) : 1.0f;
        bq[q] = ok ? b[j] : 0.f;
        cq[q] = (MODEL >= 3 && ok) ? fminf(sigmoidf_(c_un[j]), 1.0f - VX_EPS32) : 0.f;
        dq[q] = (MODEL >= 4 && ok) ? fminf(sigmoidf_(d_un[j]), 1.0f - VX_EPS32) : 1.0f;
        oq[q] = (MODEL >= 4 && ok) ? fmaxf(sigmoidf_(-d_un[j]), VX_EPS32) : 0.f;
        ga[q] = gb[q] = gc[q] = gd[q] = 0.f;
    }
    for (int64_t grp = wg; grp < n_groups; grp += n_waves) {
        const int64_t i = grp * 64 + lane;
        const bool valid = i < dm.nb;
        int64_t row = 0;
        float l = 0.f, r = 0.f, e = 0.f;
        if (valid) {
            row = rows ? rows[i] : i;
            l = loc[i]; r = raw[i];
            e = eps_in ? eps_in[i] : philox_normal4(seed, step, stream, gid0 + row, 0u)[0];
        }
        const float sig = __expf(r);
        const float xv = l + sig * e;
        float my_ll = 0.f, my_gx = 0.f;
        const int cnt = (int)((dm.nb - grp * 64) < 64 ? (dm.nb - grp * 64) : 64);
        for (int pp = 0; pp < cnt; ++pp) {
            const float x = __shfl(xv, pp, 64);
            const int64_t prow = __shfl(row, pp, 64);
            const uint8_t* yr = y + prow * J;
            float llp = 0.f, gxp = 0.f;
#pragma unroll
            for (int q = 0; q < IPL; ++q) {
                const int j = lane + 64 * q;
                if (j < J) {
                    const unsigned yy = yr[j];
                    const float z = dm.Dc * (x * aq[q] + bq[q]);
                    float lp, dz, dc, dd;
                    irt_cell<MODEL>(z, yy, cq[q], dq[q], oq[q], lp, dz, dc, dd);
                    const float t = dm.Dc * dz;
                    llp += lp;
                    gxp += t * aq[q];
                    gb[q] += t;
                    if (MODEL >= 2) ga[q] += t * x;
                    if (MODEL >= 3) gc[q] += dc;
                    if (MODEL >= 4) gd[q] += dd;
                }
            }
            llp = wave_sum(llp);
            gxp = wave_sum(gxp);
            if (lane == pp) { my_ll = llp; my_gx = gxp; }
        }
        if (valid) {
            const float gxt = dm.scale * (my_gx - xv);                 // d ELBO / d x (likelihood + prior)
            gloc[i] = -gxt;
            graw[i] = -(gxt * sig * e + dm.scale);                      // + scale from the entropy term
            elbo[i] = my_ll - 0.5f * xv * xv + 0.5f * e * e + r;        // log p(y|x) + log p(x) - log q(x)
        }
    }
    // block-level reduction of the item gradients, one slab per block: [a: J | b: J | c: J | d: J]
    for (int e = tid; e < 4 * J; e += I1_THREADS) smem[e] = 0.f;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < IPL; ++q) {
        const int j = lane + 64 * q;
        if (j < J) {
            if (MODEL >= 2) atomicAdd(&smem[j], ga[q]);
            atomicAdd(&smem[J + j], gb[q]);
            if (MODEL >= 3) atomicAdd(&smem[2 * J + j], gc[q]);
            if (MODEL >= 4) atomicAdd(&smem[3 * J + j], gd[q]);
        }
    }
    __syncthreads();
    float* slab = slabs + (int64_t)blockIdx.x * 4 * J;
    for (int e = tid; e < 4 * J; e += I1_THREADS) slab[e] = dm.scale * smem[e];
}
