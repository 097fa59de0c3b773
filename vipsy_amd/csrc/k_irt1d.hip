// D = 1 IRT (irt_1pl..4pl with a Normal guide; vi.py:22-66, 588-595, 617-625, 684/705), one fused pass:
// x = loc + exp(raw) eps -> masked Bernoulli log-lik -> prior/entropy -> gradients.
//
// Layout: item-per-lane.  Lane l keeps the parameters and gradient accumulators of items 4l..4l+3 (+256, ...)
// in registers; a wave walks its persons in groups of 64 (lane l also owns the per-person scalars of person
// 64 g + l), the response row is read as one coalesced 4-byte word per lane per 256-item slice, the next
// person's words are requested before the current one is evaluated, and the only cross-lane traffic is two
// DPP wave reductions (log-lik, d/dx) per person -- no LDS in the person loop.
#pragma once
#include "vx_common.h"

#define I1_THREADS 256

struct Irt1dDims {
    int J, model;
    float Dc, scale;
    int64_t nb;
};

// WPL = 4-item words per lane (items 256 w + 4 lane + 0..3); J <= 1024 -> WPL <= 4
// HALF (WPL == 1, J <= 128): the items fit 32 lanes, so each lane half takes its own person (lanes 32..63 repeat the item
// parameters and walk persons 32..63 of the group): two persons per wave iteration instead of one with half the lanes idle
__device__ __forceinline__ float half_sums_dpp(float v, int half) {         // sum over the 32 lanes of this lane's half
    v += dpp_mov0<0xB1, 0xF>(v);
    v += dpp_mov0<0x4E, 0xF>(v);
    v += dpp_mov0<0x141, 0xF>(v);
    v += dpp_mov0<0x140, 0xF>(v);
    v += dpp_mov0<0x142, 0xA>(v);                                          // row_bcast:15 -> lanes 31 / 63 hold the half sums
    const float lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 31));
    const float hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
    return half ? hi : lo;
}

template <int MODEL, int WPL, bool WORDS, bool HALF = false>
__global__ __launch_bounds__(I1_THREADS) void k_irt1d(
    Irt1dDims dm, const uint8_t* __restrict__ y, const int64_t* __restrict__ rows, int64_t gid0,
    const float* __restrict__ loc, const float* __restrict__ raw, const float* __restrict__ eps_in,
    uint64_t seed, uint32_t step, const uint32_t* __restrict__ step_dev, uint32_t stream, const float* __restrict__ a,
    const float* __restrict__ b, const float* __restrict__ c_un, const float* __restrict__ d_un,
    float* __restrict__ gloc, float* __restrict__ graw, float* __restrict__ elbo, float* __restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [4*J] block-level item-grad reduction
    __shared__ float el_w[I1_THREADS / 64];
    float el_acc = 0.f;
    if (step_dev) step = *step_dev;                                // replayed from a HIP graph: the counter lives on the device
    static_assert(!HALF || WPL == 1, "HALF: one word per lane");
    const int half = HALF ? (threadIdx.x >> 5) & 1 : 0;
    const int ilane = HALF ? (threadIdx.x & 31) : (threadIdx.x & 63);       // the lane's position on the item axis
    constexpr int IPL = 4 * WPL;
    const int J = dm.J;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t n_waves = (int64_t)gridDim.x * (I1_THREADS / 64);
    const int64_t wg = (int64_t)blockIdx.x * (I1_THREADS / 64) + wave;
    const int64_t n_groups = (dm.nb + 63) / 64;
    float aq[IPL], bq[IPL], cq[IPL], dq[IPL], oq[IPL], ga[IPL], gb[IPL], gc[IPL], gd[IPL];
#pragma unroll
    for (int q = 0; q < IPL; ++q) {
        const int j = 256 * (q >> 2) + 4 * ilane + (q & 3);
        const bool ok = j < J;
        aq[q] = (MODEL >= 2) ? (ok ? a[j] : 0.f) : 1.0f;
        bq[q] = ok ? b[j] : 0.f;
        cq[q] = (MODEL >= 3 && ok) ? fminf(sigmoidf_(c_un[j]), 1.0f - VX_EPS32) : 0.f;
        dq[q] = (MODEL >= 4 && ok) ? fminf(sigmoidf_(d_un[j]), 1.0f - VX_EPS32) : 1.0f;
        oq[q] = (MODEL >= 4 && ok) ? fmaxf(sigmoidf_(-d_un[j]), VX_EPS32) : 0.f;
        ga[q] = gb[q] = gc[q] = gd[q] = 0.f;
    }
    // one person's response words for this lane; 254 = outside the problem (j >= J)
    auto load_words = [&](uint32_t (&w)[WPL], int64_t prow) {
        const uint8_t* yr = y + prow * J;
#pragma unroll
        for (int u = 0; u < WPL; ++u) {
            const int j0 = 256 * u + 4 * ilane;
            if (WORDS) {                                   // J % 4 == 0: a word is entirely inside or outside
                uint32_t v = 0xFEFEFEFEu;
                if (j0 < J) v = *(const uint32_t*)(yr + j0);
                w[u] = v;
            } else {
                uint32_t v = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) v |= (uint32_t)((j0 + e < J) ? yr[j0 + e] : 254u) << (8 * e);
                w[u] = v;
            }
        }
    };
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
        const int row_lo = (int)(uint32_t)row, row_hi = (int)(uint32_t)((uint64_t)row >> 32);
        float my_ll = 0.f, my_gx = 0.f;
        const int cnt = (int)((dm.nb - grp * 64) < 64 ? (dm.nb - grp * 64) : 64);
        uint32_t wcur[WPL], wnext[WPL];
        // the response row of person pp (HALF: of person pp in the lower lane half, pp + 32 in the upper one; a person
        // past the end of the group reads a valid row and is switched off below)
        auto person_row = [&](int pp) -> int64_t {
            const int64_t r0 = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane(row_hi, pp) << 32) |
                                         (uint32_t)__builtin_amdgcn_readlane(row_lo, pp));
            if (!HALF) return r0;
            const int p1 = (pp + 32 < cnt) ? pp + 32 : pp;
            const int64_t r1 = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane(row_hi, p1) << 32) |
                                         (uint32_t)__builtin_amdgcn_readlane(row_lo, p1));
            return half ? r1 : r0;
        };
        const int n_it = HALF ? (cnt < 32 ? cnt : 32) : cnt;
        load_words(wcur, person_row(0));
        for (int pp = 0; pp < n_it; ++pp) {
            const int pn = (pp + 1 < n_it) ? pp + 1 : pp;                   // prefetch the next person's words
            load_words(wnext, person_row(pn));
            float x = lane_bcast(xv, pp);
            bool live = true;                                               // HALF: the upper half may have run out of persons
            if (HALF) {
                const float x1 = lane_bcast(xv, pp + 32 < 64 ? pp + 32 : 63);
                x = half ? x1 : x;
                live = !half || pp + 32 < cnt;
            }
            float llp = 0.f, gxp = 0.f;
#pragma unroll
            for (int q = 0; q < IPL; ++q) {
                const unsigned yy = live ? ((wcur[q >> 2] >> (8 * (q & 3))) & 0xFFu) : 254u;
                const float z = dm.Dc * fmaf(x, aq[q], bq[q]);
                float lp, dz, dc, dd;
                irt_cell<MODEL>(z, yy, cq[q], dq[q], oq[q], lp, dz, dc, dd);    // branch-free; y >= 254 -> no gradient
                const float t = dm.Dc * dz;
                llp += lp;
                gxp = fmaf(t, aq[q], gxp);
                gb[q] += t;
                if (MODEL >= 2) ga[q] = fmaf(t, x, ga[q]);
                if (MODEL >= 3) gc[q] += dc;
                if (MODEL >= 4) gd[q] += dd;
            }
            if (HALF) {
                llp = half_sums_dpp(llp, half);
                gxp = half_sums_dpp(gxp, half);
                if (lane == pp + 32 * half) { my_ll = llp; my_gx = gxp; }
            } else {
                llp = wave_sum_dpp(llp);
                gxp = wave_sum_dpp(gxp);
                if (lane == pp) { my_ll = llp; my_gx = gxp; }
            }
#pragma unroll
            for (int u = 0; u < WPL; ++u) wcur[u] = wnext[u];
        }
        if (valid) {
            const float gxt = dm.scale * (my_gx - xv);                 // d ELBO / d x (likelihood + prior)
            gloc[i] = -gxt;
            graw[i] = -(gxt * sig * e + dm.scale);                      // + scale from the entropy term
            const float el = my_ll - 0.5f * xv * xv + 0.5f * e * e + r; // log p(y|x) + log p(x) - log q(x)
            elbo[i] = el;
            el_acc += el;
        }
    }
    // block-level reduction of the item gradients, one slab per block: [a: J | b: J | c: J | d: J].  Every wave holds one
    // partial per item in the same lane: a slot per wave, then a sum in fixed order (bit-reproducible, no float atomics)
    el_acc = wave_sum_dpp(el_acc);
    if (lane == 0) el_w[wave] = el_acc;
    __syncthreads();
    float* wslot = smem + (size_t)(tid >> 6) * 4 * J;
    for (int e = lane; e < 4 * J; e += 64) wslot[e] = 0.f;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < IPL; ++q) {
        const int j = 256 * (q >> 2) + 4 * ilane + (q & 3);
        if (HALF) {                                                        // the two lane halves hold the same items
            ga[q] = half_sum32(ga[q]); gb[q] = half_sum32(gb[q]);
            if (MODEL >= 3) gc[q] = half_sum32(gc[q]);
            if (MODEL >= 4) gd[q] = half_sum32(gd[q]);
        }
        if (j < J && (!HALF || half == 0)) {
            if (MODEL >= 2) wslot[j] = ga[q];
            wslot[J + j] = gb[q];
            if (MODEL >= 3) wslot[2 * J + j] = gc[q];
            if (MODEL >= 4) wslot[3 * J + j] = gd[q];
        }
    }
    __syncthreads();
    float* slab = slabs + (int64_t)blockIdx.x * (4 * J + 1);
    for (int e = tid; e < 4 * J; e += I1_THREADS) {
        float acc = smem[e];
#pragma unroll
        for (int w = 1; w < I1_THREADS / 64; ++w) acc += smem[(size_t)w * 4 * J + e];
        slab[e] = dm.scale * acc;
    }
    if (tid == 0) {                                                    // column 4 J: this block's share of the ELBO
        float acc = el_w[0];
#pragma unroll
        for (int w = 1; w < I1_THREADS / 64; ++w) acc += el_w[w];
        slab[4 * J] = dm.scale * acc;
    }
}

// ---------------------------------------------------------------------------------------------
// Score-function (REINFORCE) estimator for the Normal guide of the D = 1 models (north_star; SURVEY.md App. A.5 -- the
// reference itself takes pathwise gradients there, vi.py:684,705): the same forward and item gradients as k_irt1d, and
//     log_r_i = scale (ll_i + log p(x_i) - log q(x_i))                       (= scale * elbo[i] of the step kernel)
//     d loss / d loc_i = - (log_r_i - baseline_i) eps_i exp(-raw_i)          (d log q / d loc = (x - loc) / sigma^2)
//     d loss / d raw_i = - (log_r_i - baseline_i) (eps_i^2 - 1)              (d log q / d raw, sigma = exp(raw))
// baseline: none, an explicit control variate (base_beta < 0), or a decaying average updated after use (base_beta >= 0),
// indexed by rows[i] when rows are given and base_by_row is set (a per-person average of a subsampled run).
// ---------------------------------------------------------------------------------------------
__global__ void k_irt1d_score(int64_t nb, float scale, const float* __restrict__ elbo, const float* __restrict__ eps,
                              const float* __restrict__ raw, const int64_t* __restrict__ rows, float* __restrict__ baseline,
                              float base_beta, int base_by_row, float* __restrict__ log_r_out, float* __restrict__ gloc,
                              float* __restrict__ graw) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += (int64_t)gridDim.x * blockDim.x) {
        const float lr = scale * elbo[i];
        float f = lr;
        if (baseline) {
            const int64_t bi = (base_by_row && rows) ? rows[i] : i;
            const float bv = baseline[bi];
            f = lr - bv;
            if (base_beta >= 0.f) baseline[bi] = fmaf(base_beta, bv, (1.0f - base_beta) * lr);
        }
        const float e = eps[i];
        gloc[i] = -f * e * __expf(-raw[i]);
        graw[i] = -f * (e * e - 1.0f);
        if (log_r_out) log_r_out[i] = lr;
    }
}

