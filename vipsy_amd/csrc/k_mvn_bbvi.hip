// Black-box MVN guide with per-person (or one shared) Cholesky rows (VIRT.guide for x_feature > 1,
// vi.py:706-723; constraints.lower_cholesky -> L = tril(M,-1) + diag(exp(diag M))).
//   forward : x_i = mu_i + L_i eps_i,  ent_i = 0.5 |eps_i|^2 + sum_k M_i,kk
//   backward: d LOSS / d mu_i = -gx_i;  d LOSS / d M_i = -(tril(gx_i eps_i^T), diag * exp(M_kk) + scale)
// One wave per batch row; lanes walk the columns l of a row k of M (coalesced), one DPP reduction per row.
// This path is HBM-bound on the D x D state ("3-bbvi variant").
#pragma once
#include "vx_common.h"

#define BB_THREADS 256

// M: per-person [n_local][D][D] (shared == 0) or one [D][D] (shared == 1), unconstrained
__global__ __launch_bounds__(BB_THREADS) void k_mvn_bbvi_fwd(
    int D, int64_t nb, const int64_t* __restrict__ rows, int64_t gid0, const float* __restrict__ loc,
    const float* __restrict__ M, int shared, const float* __restrict__ eps_in, uint64_t seed, uint32_t step,
    uint32_t stream, float* __restrict__ x, float* __restrict__ eps_out, float* __restrict__ ent,
    const uint32_t* __restrict__ step_dev = nullptr /*or the step counter in device memory (a captured step)*/) {
    if (step_dev) step = *step_dev;
    const int lane = threadIdx.x & 63;
    const int64_t nw = (int64_t)gridDim.x * (BB_THREADS / 64);
    for (int64_t i = (int64_t)blockIdx.x * (BB_THREADS / 64) + (threadIdx.x >> 6); i < nb; i += nw) {
        const int64_t prow = rows ? rows[i] : i;
        const float* Mi = shared ? M : M + prow * D * D;
        float e0 = 0.f, e1 = 0.f;                      // eps of columns lane, lane + 64
        if (eps_in) {
            if (lane < D) e0 = eps_in[i * D + lane];
            if (lane + 64 < D) e1 = eps_in[i * D + lane + 64];
        } else {
            if (lane < D) e0 = philox_normal4(seed, step, stream, gid0 + prow, (uint32_t)(lane >> 2))[lane & 3];
            if (lane + 64 < D) e1 = philox_normal4(seed, step, stream, gid0 + prow, (uint32_t)((lane + 64) >> 2))[lane & 3];
        }
        if (lane < D) eps_out[i * D + lane] = e0;
        if (lane + 64 < D) eps_out[i * D + lane + 64] = e1;
        float logdet = 0.f;
        for (int k = 0; k < D; ++k) {
            float part = 0.f;
            if (lane <= k) {
                const float m = Mi[k * D + lane];
                part = (lane == k) ? __expf(m) * e0 : m * e0;
                if (lane == k) logdet += m;
            }
            if (lane + 64 <= k) {
                const float m = Mi[k * D + lane + 64];
                part += (lane + 64 == k) ? __expf(m) * e1 : m * e1;
                if (lane + 64 == k) logdet += m;
            }
            const float xk = wave_sum_dpp(part);
            if (lane == 0) x[i * D + k] = loc[prow * D + k] + xk;
        }
        const float sq = wave_sum_dpp(e0 * e0 + e1 * e1);
        const float ld = wave_sum_dpp(logdet);
        if (lane == 0) ent[i] = 0.5f * sq + ld;
    }
}

// gloc: dense [n_local][D] (rows of the batch are written, the caller zeroed the rest);
// gM  : dense [n_local][D][D] (shared == 0), or one [D][D] slab PER BLOCK (shared == 1) that the host sums in fixed
//       order: inside the block the waves accumulate in 64-bit fixed point (fx_add), so neither level depends on the order
//       in which the hardware retires atomics -- the shared-covariance gradient is bit-reproducible
__global__ __launch_bounds__(BB_THREADS) void k_mvn_bbvi_bwd(
    int D, int64_t nb, float scale, const int64_t* __restrict__ rows, const float* __restrict__ M, int shared,
    const float* __restrict__ gx, const float* __restrict__ eps, float* __restrict__ gloc, float* __restrict__ gM) {
    extern __shared__ __attribute__((aligned(16))) long long smem_fx[];   // shared == 1: [D][D] block accumulator
    const int lane = threadIdx.x & 63;
    if (shared) {
        for (int e = threadIdx.x; e < D * D; e += BB_THREADS) smem_fx[e] = 0;
        __syncthreads();
    }
    const int64_t nw = (int64_t)gridDim.x * (BB_THREADS / 64);
    for (int64_t i = (int64_t)blockIdx.x * (BB_THREADS / 64) + (threadIdx.x >> 6); i < nb; i += nw) {
        const int64_t prow = rows ? rows[i] : i;
        const float* Mi = shared ? M : M + prow * D * D;
        const float e0 = lane < D ? eps[i * D + lane] : 0.f;
        const float e1 = lane + 64 < D ? eps[i * D + lane + 64] : 0.f;
        if (lane < D) gloc[prow * D + lane] = -gx[i * D + lane];
        if (lane + 64 < D) gloc[prow * D + lane + 64] = -gx[i * D + lane + 64];
        for (int k = 0; k < D; ++k) {
            const float gk = gx[i * D + k];                                   // uniform
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int l = lane + 64 * h;
                if (l < D) {
                    float v = 0.f;
                    if (l < k) v = gk * (h ? e1 : e0);
                    else if (l == k) v = gk * (h ? e1 : e0) * __expf(Mi[k * D + k]) + scale;
                    if (shared) { if (l <= k) fx_add(&smem_fx[k * D + l], -v); }
                    else gM[(prow * D + k) * D + l] = -v;
                }
            }
        }
    }
    if (shared) {
        __syncthreads();
        float* slab = gM + (int64_t)blockIdx.x * D * D;
        for (int e = threadIdx.x; e < D * D; e += BB_THREADS) slab[e] = fx_get(smem_fx[e]);
    }
}
