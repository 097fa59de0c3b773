// HO-DINA with exact enumeration of the 2^K attribute patterns (VCHoDina.model / .guide, vi.py:897-934;
// dina vi.py:69-83; pattern table vi.py:825-837; TraceEnum_ELBO: log-space sum-product per person).
//
// The reference materialises p[c, i, j] (C x B x J floats).  Here one wave handles one person at a time
// with the C = 2^K patterns spread over the lanes (pattern c = CPL * lane + i), and the two contractions
// over the pattern lattice are fast zeta transforms instead of dense products:
//     B_c  = sum_j eta_cj delta_j          = SUBSET sum of f(S) = sum_{j: q_j = S} delta_j   (eta_cj = [c >= q_j])
//     E_j  = sum_c r_c eta_cj              = SUPERSET sum of r evaluated at S = q_j
//     tau_k= sum_c rho_c alpha_ck - pi_k sum_c rho_c = SUPERSET sum of rho at {k} and at {}
// Item parameters (log g, log(1-g), ...) live in registers of lane j; their gradients accumulate there too.
#pragma once
#include "vx_common.h"

#define HD_THREADS 256
#define HD_WAVES (HD_THREADS / 64)

struct HoDinaDims {
    int K, J, C;
    float scale;
    int64_t nb;
    // VCCDM (vi.py:819-865): uniform prior over the patterns, no theta / lambda; dino = the reference's dino()
    // (vi.py:86-101) INCLUDING its in-place sequencing: eta = [item needs >= 2 attributes] * [c masters one of them]
    int uniform_prior, dino;       // uniform_prior: 0 HO-DINA prior, 1 uniform (VCCDM), 2 per-person row of pattern scores (VaeCCDM)
    int unmasked;                  // VaeCCDM (vi.py:882-891): a missing response stays in `obs` as -1 (no mask)
    const uint32_t* step_dev = nullptr;   // or the step counter in device memory (a captured step): read instead of `step`
    // persons a wave takes at a time (1..64; hd_group_size): a wave walks its group ONE PERSON AFTER THE OTHER (~2 us each), so a
    // small batch in groups of 64 is two waves busy for 200 us (VCCDM with the reference's 100 rows a step: 204 us of a 215 us
    // step); groups shrink until the batch fills 4 096 waves
    int gsz = 64;
};

__host__ __device__ inline int hd_group_size(int64_t nb, int64_t max_waves) {
    int64_t g = (nb + max_waves - 1) / max_waves;
    return (int)(g < 1 ? 1 : g > 64 ? 64 : g);
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// in-place zeta transform over the pattern lattice; v[i] holds pattern c = CPL*lane + i.
// SUPER = false: v[c] <- sum_{S subset of c} v[S];  SUPER = true: v[c] <- sum_{S superset of c} v[S]
template <int LOGCPL, bool SUPER>
__device__ __forceinline__ void zeta(float (&v)[1 << LOGCPL], int K, int lane) {
    constexpr int CPL = 1 << LOGCPL;
#pragma unroll
    for (int b = 0; b < LOGCPL; ++b) {
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            if (SUPER) { if (!(i & (1 << b))) v[i] += v[i | (1 << b)]; }
            else       { if (i & (1 << b)) v[i] += v[i ^ (1 << b)]; }
        }
    }
    for (int b = LOGCPL; b < K; ++b) {
        const int lb = 1 << (b - LOGCPL);
        const bool has = (lane & lb) != 0;
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const float t = __shfl_xor(v[i], lb, 64);
            if (SUPER ? !has : has) v[i] += t;
        }
    }
}

// Bernoulli log-prob of y under a CONSTANT success probability P (clamped like torch clamp_probs):
// returns lp and dlp/dP (0 when P is outside [eps, 1-eps]); y == 255 -> missing cell.
__device__ __forceinline__ void bern_const(float P, float Q /* = 1 - P, accurate */, unsigned y, float& lp, float& dP,
                                           bool unmasked = false) {
    const bool inside = (P >= VX_EPS32) && (Q >= VX_EPS32);
    const float Pc = fminf(fmaxf(P, VX_EPS32), 1.0f - VX_EPS32);
    const float Qc = fminf(fmaxf(Q, VX_EPS32), 1.0f - VX_EPS32);
    if (y == 255u) {
        if (!unmasked) { lp = VX_LOGP_MISSING; dP = 0.f; return; }
        // Bernoulli(p).log_prob(-1) = -logit - softplus(logit) = 2 log(1 - p) - log p;  d/dP = (-1 - p) / (p (1 - p))
        lp = 2.0f * logf(Qc) - logf(Pc);
        dP = inside ? (-1.0f - Pc) / (Pc * Qc) : 0.f;
        return;
    }
    lp = (y != 0u) ? logf(Pc) : logf(Qc);
    dP = inside ? ((y != 0u) ? 1.0f / Pc : -1.0f / Qc) : 0.f;
}

// slab layout (one per block): [g_un: J | s_un: J | lam0: K | lam1_un: K]  (d ELBO, scaled)
template <int LOGCPL, int JPL>
__global__ __launch_bounds__(HD_THREADS) void k_hodina(
    HoDinaDims dm, const uint8_t* __restrict__ y, const int64_t* __restrict__ rows, int64_t gid0,
    const float* __restrict__ loc, const float* __restrict__ raw, const float* __restrict__ eps_in,
    uint64_t seed, uint32_t step, uint32_t stream, const float* __restrict__ q, const float* __restrict__ lam0,
    const float* __restrict__ lam1_un, const float* __restrict__ g_un, const float* __restrict__ s_un,
    float* __restrict__ gloc, float* __restrict__ graw, float* __restrict__ elbo, float* __restrict__ slabs,
    const float* __restrict__ zrow = nullptr /*[nb][C] pattern scores (uniform_prior == 2)*/,
    const float* __restrict__ zoff = nullptr /*[C] column offsets: prior weight = exp(z - off)*/,
    float* __restrict__ gla = nullptr /*[nb][C] out: d ELBO / d log(prior weight), scaled*/) {
    constexpr int CPL = 1 << LOGCPL;
    if (dm.step_dev) step = *dm.step_dev;                          // replayed from a HIP graph: the counter lives on the device
    extern __shared__ __attribute__((aligned(16))) float smem[];   // per wave: [C] scatter/gather table; then block reduce
    const int K = dm.K, J = dm.J, C = dm.C;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    long long* tabx = (long long*)smem + (size_t)wave * C;          // per wave: [C] fixed-point sums / float table (aliased)
    float* tab = (float*)tabx;
    const int64_t n_waves = (int64_t)gridDim.x * HD_WAVES;
    const int64_t wg = (int64_t)blockIdx.x * HD_WAVES + wave;
    const int G = dm.gsz;
    const int64_t n_groups = (dm.nb + G - 1) / G;
    // ---- per-item constants in lane j (+64, ...)
    float gj[JPL], sj[JPL], og[JPL], os[JPL], gg[JPL], gs[JPL];
    int qpat[JPL];
#pragma unroll
    for (int u = 0; u < JPL; ++u) {
        const int j = lane + 64 * u;
        gj[u] = sj[u] = 0.5f; og[u] = os[u] = 0.5f; gg[u] = gs[u] = 0.f; qpat[u] = 0;
        if (j < J) {
            gj[u] = fminf(sigmoidf_(g_un[j]), 1.0f - VX_EPS32);
            og[u] = fmaxf(sigmoidf_(-g_un[j]), VX_EPS32);            // 1 - g, from the leaf (no cancellation)
            sj[u] = fminf(sigmoidf_(s_un[j]), 1.0f - VX_EPS32);
            os[u] = fmaxf(sigmoidf_(-s_un[j]), VX_EPS32);            // 1 - s
            for (int k = 0; k < K; ++k)
                if (q[(int64_t)k * J + j] != 0.f) qpat[u] |= (1 << k);
        }
    }
    // ---- per-attribute constants in lane k
    const float l0 = (lane < K && !dm.uniform_prior) ? lam0[lane] : 0.f;
    const float l1 = (lane < K && !dm.uniform_prior) ? __expf(lam1_un[lane]) : 0.f;
    float gl0 = 0.f, gl1 = 0.f;

    for (int64_t grp = wg; grp < n_groups; grp += n_waves) {
        const int64_t i = grp * G + lane;
        const bool valid = lane < G && i < dm.nb;
        int64_t row = 0;
        float lc = 0.f, rw = 0.f, e = 0.f;
        if (valid) {
            row = rows ? rows[i] : i;
            if (!dm.uniform_prior) {
                lc = loc[i]; rw = raw[i];
                e = eps_in ? eps_in[i] : philox_normal4(seed, step, stream, gid0 + row, 0u)[0];
            }
        }
        const float sig = __expf(rw);
        const float thv = lc + sig * e;
        float my_elbo = 0.f, my_gth = 0.f;
        const int cnt = (int)((dm.nb - grp * G) < G ? (dm.nb - grp * G) : G);
        for (int pp = 0; pp < cnt; ++pp) {
            const float th = lane_bcast(thv, pp);
            const int64_t prow = __shfl(row, pp, 64);
            const uint8_t* yr = y + prow * J;
            // -- item side: lp0 = log Bern(y; g), lp1 = log Bern(y; 1 - s)   (p_cj in {g_j, 1 - s_j}, vi.py:82)
            float d0[JPL], d1[JPL], base = 0.f;
#pragma unroll
            for (int c4 = lane; c4 < CPL * 64; c4 += 64) if (c4 < C) tabx[c4] = 0;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int u = 0; u < JPL; ++u) {
                const int j = lane + 64 * u;
                d0[u] = d1[u] = 0.f;
                if (j < J) {
                    const unsigned yy = yr[j];
                    float lp0, lp1;
                    bern_const(gj[u], og[u], yy, lp0, d0[u], dm.unmasked != 0);
                    bern_const(os[u], sj[u], yy, lp1, d1[u], dm.unmasked != 0);
                    base += lp0;
                    if (!dm.dino || __popc(qpat[u]) >= 2)
                        fx_add(&tabx[qpat[u]], lp1 - lp0);             // f(S) = sum of delta_j over items with q_j = S
                                                                       // (fixed point: the order of the adds does not matter)
                }
            }
            base = wave_sum_dpp(base);
            __builtin_amdgcn_wave_barrier();
            float Bc[CPL];
#pragma unroll
            for (int ii = 0; ii < CPL; ++ii) { const int c = CPL * lane + ii; Bc[ii] = (c < C) ? fx_get(tabx[c]) : 0.f; }
            __builtin_amdgcn_wave_barrier();                              // the table is reused as floats below
            zeta<LOGCPL, false>(Bc, K, lane);
            if (dm.dino) {
                // eta_cj = [c meets q_j]: B_c = sum_j delta_j - sum_{q_j disjoint from c} delta_j = Z[full] - Z[~c]
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int ii = 0; ii < CPL; ++ii) { const int c = CPL * lane + ii; if (c < C) tab[c] = Bc[ii]; }
                __builtin_amdgcn_wave_barrier();
                const float zfull = tab[C - 1];
#pragma unroll
                for (int ii = 0; ii < CPL; ++ii) { const int c = CPL * lane + ii; Bc[ii] = (c < C) ? zfull - tab[(C - 1) ^ c] : 0.f; }
                __builtin_amdgcn_wave_barrier();
            }
            // -- attribute side (vi.py:911-912): t_k = theta lam1_k + lam0_k in lane k
            const float tk = th * l1 + l0;
            const float pik = sigmoidf_(tk);
            float S0 = (lane < K) ? -softplusf_(tk) : 0.f;                 // log(1 - pi_k)
            S0 = wave_sum_dpp(S0);
            float Ac[CPL];
#pragma unroll
            for (int ii = 0; ii < CPL; ++ii) Ac[ii] = S0;
            for (int k = 0; k < K; ++k) {
                const float t = lane_bcast(tk, k);
#pragma unroll
                for (int ii = 0; ii < CPL; ++ii)
                    if ((CPL * lane + ii) & (1 << k)) Ac[ii] += t;
            }
            // -- Categorical(probs): renormalise, clamp to [eps, 1-eps], log  (torch probs_to_logits)
            float pr[CPL], fc[CPL], psum = 0.f;
            bool ins[CPL];
#pragma unroll
            for (int ii = 0; ii < CPL; ++ii) {
                const int c = CPL * lane + ii;
                const int64_t ip = grp * G + pp;
                pr[ii] = (c < C) ? (dm.uniform_prior == 2 ? __expf(zrow[ip * C + c] - zoff[c])
                                    : dm.uniform_prior ? 1.0f : __expf(Ac[ii])) : 0.f;       // Categorical(1 / C): vi.py:849
                psum += pr[ii];
            }
            psum = wave_sum_dpp(psum);
            float fmx = -3.0e38f;
#pragma unroll
            for (int ii = 0; ii < CPL; ++ii) {
                const int c = CPL * lane + ii;
                pr[ii] = pr[ii] / psum;
                ins[ii] = (pr[ii] >= VX_EPS32) && (pr[ii] <= 1.0f - VX_EPS32);
                const float lg = logf(fminf(fmaxf(pr[ii], VX_EPS32), 1.0f - VX_EPS32));
                fc[ii] = (c < C) ? lg + base + Bc[ii] : -3.0e38f;
                fmx = fmaxf(fmx, fc[ii]);
            }
            fmx = wave_max_dpp(fmx);
            float rs = 0.f, rc[CPL];
#pragma unroll
            for (int ii = 0; ii < CPL; ++ii) { rc[ii] = (CPL * lane + ii < C) ? __expf(fc[ii] - fmx) : 0.f; rs += rc[ii]; }
            rs = wave_sum_dpp(rs);
            const float lse = fmx + logf(rs);
            const float rinv = 1.0f / rs;
            float rins = 0.f;
#pragma unroll
            for (int ii = 0; ii < CPL; ++ii) { rc[ii] *= rinv; rins += ins[ii] ? rc[ii] : 0.f; }
            rins = wave_sum_dpp(rins);
            float rho[CPL];
#pragma unroll
            for (int ii = 0; ii < CPL; ++ii) rho[ii] = (ins[ii] ? rc[ii] : 0.f) - pr[ii] * rins;
            if (gla) {                                                    // d ELBO / d log a_ic (through the row normalisation)
#pragma unroll
                for (int ii = 0; ii < CPL; ++ii) {
                    const int c = CPL * lane + ii;
                    if (c < C) gla[(grp * G + pp) * C + c] = dm.scale * rho[ii];
                }
            }
            // -- E_j = sum_c r_c eta_cj.  DINA: superset sums gathered at q_j.  DINO: 1 - (subset sum at ~q_j), and 0
            //    for items that need fewer than two attributes.
            if (dm.dino) zeta<LOGCPL, false>(rc, K, lane);
            else zeta<LOGCPL, true>(rc, K, lane);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ii = 0; ii < CPL; ++ii) { const int c = CPL * lane + ii; if (c < C) tab[c] = rc[ii]; }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int u = 0; u < JPL; ++u) {
                const int j = lane + 64 * u;
                if (j < J) {
                    float E = tab[qpat[u]];
                    if (dm.dino) E = (__popc(qpat[u]) >= 2) ? 1.0f - tab[(C - 1) ^ qpat[u]] : 0.f;
                    gg[u] += (1.0f - E) * d0[u];                          // d/dg through patterns that do NOT master item j
                    gs[u] -= E * d1[u];                                   // d/ds: p = 1 - s
                }
            }
            // -- tau_k = sum_c rho_c alpha_ck - pi_k sum_c rho_c
            zeta<LOGCPL, true>(rho, K, lane);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ii = 0; ii < CPL; ++ii) { const int c = CPL * lane + ii; if (c < C) tab[c] = rho[ii]; }
            __builtin_amdgcn_wave_barrier();
            float tau = 0.f;
            if (lane < K) tau = tab[1 << lane] - pik * tab[0];
            gl0 += tau;
            gl1 += tau * th;
            float gth = wave_sum_dpp(tau * l1);
            gth -= th;                                                    // prior N(0,1)
            if (lane == pp) {
                my_gth = gth;
                my_elbo = dm.uniform_prior ? lse : lse - 0.5f * th * th;   // + 0.5 eps^2 + raw added below
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (valid) {
            if (dm.uniform_prior) {
                elbo[i] = my_elbo;
            } else {
                const float gt = dm.scale * my_gth;
                gloc[i] = -gt;
                graw[i] = -(gt * sig * e + dm.scale);
                elbo[i] = my_elbo + 0.5f * e * e + rw;
            }
        }
    }
    // ---- block reduction of item / attribute gradients -> one slab per block: a slot per wave (every wave holds one
    // partial per item / attribute in the same lane), summed in fixed order: bit-reproducible, no float atomics
    __syncthreads();
    const int len = 2 * J + 2 * K;
    float* red = smem + (size_t)wave * len;                               // [waves][2J + 2K]
#pragma unroll
    for (int u = 0; u < JPL; ++u) {
        const int j = lane + 64 * u;
        if (j < J) {
            red[j] = gg[u] * gj[u] * og[u];                                // chain through sigmoid: g (1 - g)
            red[J + j] = gs[u] * sj[u] * os[u];
        }
    }
    if (lane < K) {
        red[2 * J + lane] = gl0;
        red[2 * J + K + lane] = gl1 * l1;                                  // chain through exp
    }
    __syncthreads();
    float* slab = slabs + (int64_t)blockIdx.x * len;
    for (int e2 = tid; e2 < len; e2 += HD_THREADS) {
        float acc = smem[e2];
#pragma unroll
        for (int w = 1; w < HD_WAVES; ++w) acc += smem[(size_t)w * len + e2];
        slab[e2] = dm.scale * acc;
    }
}
