// Score-function (REINFORCE) estimator for the multivariate Normal guides, x_feature > 1 (BASELINE.json north_star; SURVEY.md
// App. A.5).  The reference's guides are reparameterised (vi.py:693,715,723: MultivariateNormal has rsample), so this mode has
// no reference output: an opt-in of this build, checked against oracle/vi_oracle.py::irt_particle(estimator = "score").
//
// With x_i = loc_i + L_i eps_i held fixed, log q(x_i) = -sum_k log L_kk - 0.5 |L^-1 (x_i - loc_i)|^2 + const gives
//     d log q / d loc   = u,            u = L^-T eps                      (one back-substitution per person)
//     d log q / d L_kc  = u_k eps_c     (c < k)
//     d log q / d M_kk  = u_k eps_k L_kk - 1                             (L_kk = exp(M_kk), vi.py:452-454 / :711-714)
// -- the SAME shape as the pathwise gradient (gx_k eps_c off the diagonal, gx_k eps_k L_kk + scale on it) with
// w_i u_i in the place of gx_i = d ELBO / d x_i and -w_i in the place of the entropy term's +scale, w_i = log_r_i -
// baseline_i, log_r_i = scale (ll_i + ent_i) (the constants of log p(x) and log q cancel).  This kernel therefore only
// makes the OPERANDS the existing guide-backward kernels take: gx (person-major), gxT and the DIAG-row operand gdT
// (dimension-major), and w for the callers that add the diagonal term themselves.
//
// One lane = one person; the rows of L come from
//     KIND 0  the encoder heads: M_jk = W22[(j, k)] . h_i + b22[(j, k)]  (the (B, D, D) matrix is never stored: every row is a
//             dot product against the person's h, the weight row read once per wave through the scalar cache)
//     KIND 1  the person's own unconstrained M[row][D][D];   KIND 2  the shared M[D][D].
// Plain fp32 vector code: this is an opt-in estimator, not the benchmarked step (1M x 100-dim: T H = 323 k multiply-adds per
// person, ~10 ms).
#pragma once
#include "vx_common.h"

#define MS_THREADS 64

__host__ __device__ inline size_t ms_lds_bytes(int D, int H, int kind) {
    return (size_t)(D + (kind == 0 ? H : 0)) * MS_THREADS * sizeof(float);
}

template <int KIND>
__global__ __launch_bounds__(MS_THREADS) void k_mvn_score_operands(
    int D, int H, int64_t nb, float scale, const int64_t* __restrict__ rows, const float* __restrict__ h /*[nb][H]*/,
    const float* __restrict__ W22 /*[T][H]*/, const float* __restrict__ b22 /*[T]*/, const float* __restrict__ M,
    const float* __restrict__ eps /*[nb][D]*/, const float* __restrict__ ll, const float* __restrict__ ent,
    float* __restrict__ baseline, float base_beta, int base_by_row, float* __restrict__ log_r_out, float* __restrict__ w_out,
    float* __restrict__ gx /*[nb][D] or null*/, float* __restrict__ gxT /*[D][nb] or null*/, float* __restrict__ gdT /*or null*/) {
    extern __shared__ float ms_lds[];
    float* const u_l = ms_lds;                                         // [D][64]
    float* const h_l = ms_lds + (size_t)D * MS_THREADS;                // [H][64]   (KIND 0)
    const int lane = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * MS_THREADS + lane;
    const bool live = i < nb;
    const int64_t ic = live ? i : nb - 1;                              // absent lanes: the last person, never stored
    const int64_t row = rows ? rows[ic] : ic;
    float w = 0.f;
    {
        const float lr = scale * (ll[ic] + ent[ic]);
        w = lr;
        if (baseline) {
            const int64_t bi = (base_by_row && rows) ? row : ic;
            const float bv = baseline[bi];
            w = lr - bv;
            if (live && base_beta >= 0.f) baseline[bi] = fmaf(base_beta, bv, (1.0f - base_beta) * lr);
        }
        if (live) {
            if (log_r_out) log_r_out[i] = lr;
            if (w_out) w_out[i] = w;
        }
    }
    if (KIND == 0)
        for (int m = 0; m < H; ++m) h_l[m * MS_THREADS + lane] = h[ic * H + m];
    const float* Mi = KIND == 1 ? M + row * (int64_t)D * D : M;
    // entry (j, k), j >= k, of the unconstrained matrix
    auto entry = [&](int j, int k) -> float {
        if (KIND == 0) {
            const int64_t t = (int64_t)j * (j + 1) / 2 + k;            // row-major lower triangle (torch.tril_indices, vi.py:453)
            const float* wr = W22 + t * H;                             // uniform address: scalar loads
            float acc = b22[t];
            for (int m = 0; m < H; ++m) acc = fmaf(wr[m], h_l[m * MS_THREADS + lane], acc);
            return acc;
        }
        return Mi[(int64_t)j * D + k];
    };
    for (int k = D - 1; k >= 0; --k) {
        float acc = 0.f;
        for (int j = D - 1; j > k; --j) acc = fmaf(entry(j, k), u_l[j * MS_THREADS + lane], acc);
        const float lkk = __expf(entry(k, k));
        const float e = eps[ic * D + k];
        const float uk = (e - acc) / lkk;
        u_l[k * MS_THREADS + lane] = uk;
        if (live) {
            const float g = w * uk;
            if (gx) gx[i * D + k] = g;
            if (gxT) gxT[(int64_t)k * nb + i] = g;
            if (gdT) gdT[(int64_t)k * nb + i] = fmaf(g * e, lkk, -w);
        }
    }
}

// The diagonal term the black-box guides' backward adds as a constant (k_mvn_bbvi_bwd: -(... + scale), run with scale = 0 in
// this mode) is -w_i here: d LOSS / d M_kk += w_i for the person's own M, += sum_i w_i (fixed order) for the shared one.
__global__ __launch_bounds__(256) void k_mvn_score_diag(int D, int64_t nb, const int64_t* __restrict__ rows,
                                                       const float* __restrict__ w, int shared, float* __restrict__ gM) {
    if (!shared) {
        for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nb * D; e += (int64_t)gridDim.x * blockDim.x) {
            const int64_t i = e / D;
            const int k = (int)(e - i * D);
            const int64_t row = rows ? rows[i] : i;
            gM[(row * D + k) * (int64_t)D + k] += w[i];
        }
        return;
    }
    __shared__ double part[256];                                       // one block: a fixed summation order
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < nb; i += 256) s += (double)w[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    if ((int)threadIdx.x < D) gM[(int64_t)threadIdx.x * D + threadIdx.x] += (float)part[0];
}
