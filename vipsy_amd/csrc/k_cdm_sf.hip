// Bernoulli-guide DINA / DINO with the score-function (REINFORCE) estimator -- VCDM / VaeCDM of the reference
// (vi.py:726-816; dina vi.py:69-83, dino vi.py:86-101 with its in-place sequencing; BinEncoder vi.py:458-470) as
// pyro's Trace_ELBO evaluates a non-reparameterised guide site:
//
//   attr_ik ~ Bernoulli(p_ik), p = sigmoid(u) (guide logits u: rows of the `attr_p` leaf, or the encoder output)
//   log_r_i  = scale [ sum_k log prior(attr_ik) + sum_j log Bern(y_ij | P_ij) - sum_k log q(attr_ik) ]
//   loss     = - sum_i log_r_i
//   d loss / d u_ik = - (log_r_i - baseline_i) (attr_ik - p_ik)        (score term unscaled; 0 where a clamp is active)
//                     baseline: none (pyro's Trace_ELBO), a per-person decaying average of log_r, or the leave-one-out mean
//                     over the particles of the step -- any baseline that does not depend on attr_i keeps the estimator unbiased
//   d loss / d g_un_j, s_un_j : pathwise through scale log Bern(y_ij | P_ij), P_ij = eta_ij ? 1 - s_j : g_j
//
// One lane = one person; the item loop is uniform across the wave, so the item gradients reduce to COUNTS of the four
// (eta, y) combinations per item -- wave ballots + popcounts, integer adds: exact and order-independent.
// This is HBM / latency-bound byte work (K <= 10 bit masks, J table look-ups per person), no matrix pipe.
#pragma once
#include "vx_common.h"

#define CS_THREADS 256
#define CS_MAXK 10

struct CdmSfDims {
    int K, J, dino, clamp_t;          // clamp_t: the guide probabilities come from a unit_interval leaf (SigmoidTransform clamp)
    float scale, lp1, lp0;            // log prior(attr = 1), log prior(attr = 0) (clamp_probs applied by the host)
    float base_beta;                  // >= 0: baseline[.] <- beta * baseline + (1 - beta) * log_r after use (decaying average)
    int base_by_row;                  // the baseline is indexed by the local person row (else by the batch position)
    int64_t nb;
    const uint32_t* step_dev = nullptr;   // or the step counter in device memory (a captured step): read instead of `step`
};

// u01 draws for the attribute bits of person gid: word (k & 3) of Philox block k >> 2 (same rule in oracle/vi_oracle.py)
__device__ __forceinline__ float cs_uniform(uint64_t seed, uint32_t step, uint32_t stream, int64_t gid, int k, u32x4& cache,
                                            int& cached_blk) {
    const int blk = k >> 2;
    if (blk != cached_blk) {
        cache = philox4x32_10((uint32_t)gid, (uint32_t)((uint64_t)gid >> 32), step, (stream << 16) | (uint32_t)blk,
                              (uint32_t)seed, (uint32_t)(seed >> 32));
        cached_blk = blk;
    }
    const uint32_t w = (k & 3) == 0 ? cache.x : (k & 3) == 1 ? cache.y : (k & 3) == 2 ? cache.z : cache.w;
    return u01(w);
}

// counts[block][4][J] (int32): c = 2 eta + y
__global__ __launch_bounds__(CS_THREADS) void k_cdm_sf(
    CdmSfDims dm, const uint8_t* __restrict__ y, const int64_t* __restrict__ rows, int64_t gid0,
    const float* __restrict__ u /*[nb][K] guide logits, batch order*/, const uint8_t* __restrict__ attr_in /*[nb][K] or null*/,
    uint64_t seed, uint32_t step, uint32_t stream, const float* __restrict__ q /*[K][J]*/,
    const float* __restrict__ g_un, const float* __restrict__ s_un, float* __restrict__ baseline /*control variate or null*/,
    float* __restrict__ gu /*[nb][K]*/, float* __restrict__ log_r /*[nb]*/, uint8_t* __restrict__ attr_out /*[nb][K] or null*/,
    int* __restrict__ counts) {
    extern __shared__ __attribute__((aligned(16))) char smem_cs[];
    if (dm.step_dev) step = *dm.step_dev;                  // replayed from a HIP graph: the counter lives on the device
    const int K = dm.K, J = dm.J;
    float* L = (float*)smem_cs;                            // [J][4]: log-lik term of (eta, y) = c >> 1, c & 1
    int* cnt = (int*)(L + 4 * J);                          // [4][J]
    uint32_t* req = (uint32_t*)(cnt + 4 * J);              // [J]: required-attribute mask | count << 16
    const int tid = threadIdx.x, lane = tid & 63;
    for (int j = tid; j < J; j += CS_THREADS) {
        uint32_t m = 0, c = 0;
        for (int k = 0; k < K; ++k)
            if (q[(int64_t)k * J + j] != 0.f) { m |= 1u << k; ++c; }
        req[j] = m | (c << 16);
        // torch Bernoulli(probs = P).log_prob: clamp_probs, then y log P + (1 - y) log(1 - P)
        const float g = sigmoidf_(g_un[j]), s = sigmoidf_(s_un[j]);
        const float P0 = fminf(fmaxf(g, VX_EPS32), 1.0f - VX_EPS32), P1 = fminf(fmaxf(1.0f - s, VX_EPS32), 1.0f - VX_EPS32);
        L[4 * j + 0] = log1pf(-P0); L[4 * j + 1] = logf(P0);
        L[4 * j + 2] = log1pf(-P1); L[4 * j + 3] = logf(P1);
        cnt[j] = 0; cnt[J + j] = 0; cnt[2 * J + j] = 0; cnt[3 * J + j] = 0;
    }
    __syncthreads();
    const int64_t n_groups = (dm.nb + 63) / 64;
    for (int64_t grp = (int64_t)blockIdx.x * (CS_THREADS / 64) + (tid >> 6); grp < n_groups;
         grp += (int64_t)gridDim.x * (CS_THREADS / 64)) {
        const int64_t i = grp * 64 + lane;
        const bool live = i < dm.nb;
        const int64_t ic = live ? i : dm.nb - 1;
        const int64_t row = rows ? rows[ic] : ic;
        // ---- the guide draw and its log-probabilities
        uint32_t abits = 0;
        float lq = 0.f, lpa = 0.f, pk[CS_MAXK];
        bool ins[CS_MAXK];
        u32x4 cache = {0, 0, 0, 0};
        int cached = -1;
#pragma unroll
        for (int k = 0; k < CS_MAXK; ++k) {
            if (k < K) {
                const float ps = sigmoidf_(u[ic * K + k]);
                float p = ps;
                bool in_t = true;
                if (dm.clamp_t) {                          // SigmoidTransform: clamp(sigmoid, tiny, 1 - eps)
                    p = fminf(fmaxf(ps, 1.17549435e-38f), 1.0f - VX_EPS32);
                    in_t = ps >= 1.17549435e-38f && ps <= 1.0f - VX_EPS32;
                }
                const float pc = fminf(fmaxf(p, VX_EPS32), 1.0f - VX_EPS32);
                ins[k] = in_t && p >= VX_EPS32 && p <= 1.0f - VX_EPS32;
                pk[k] = pc;
                bool bit;
                if (attr_in) bit = attr_in[ic * K + k] != 0;
                else bit = cs_uniform(seed, step, stream, gid0 + row, k, cache, cached) < p;
                if (bit) abits |= 1u << k;
                lq += bit ? logf(pc) : log1pf(-pc);
                lpa += bit ? dm.lp1 : dm.lp0;
                if (attr_out && live) attr_out[i * K + k] = bit ? 1 : 0;
            }
        }
        // ---- likelihood: the item loop is wave-uniform; counts of (eta, y) by ballot
        float ll = 0.f;
        const uint8_t* yr = y + row * J;
        for (int j = 0; j < J; ++j) {
            const uint32_t rq = req[j], m = rq & 0xffffu, c = rq >> 16;
            const uint32_t missing = __builtin_popcount(m & ~abits);
            const bool eta = dm.dino ? (c > 1 && missing < c) : (missing == 0);
            const bool yy = yr[j] == 1;
            const int code = (eta ? 2 : 0) + (yy ? 1 : 0);
            ll += L[4 * j + code];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const unsigned long long b = __ballot(live && code == cc);
                if (lane == 0 && b) atomicAdd(&cnt[cc * J + j], __builtin_popcountll(b));
            }
        }
        const float lr = dm.scale * (lpa + ll - lq);
        if (live) {
            log_r[i] = lr;
            float f = lr;
            if (baseline) {
                const int64_t bi = dm.base_by_row ? row : i;
                const float bv = baseline[bi];
                f = lr - bv;
                if (dm.base_beta >= 0.f) baseline[bi] = fmaf(dm.base_beta, bv, (1.0f - dm.base_beta) * lr);
            }
#pragma unroll
            for (int k = 0; k < CS_MAXK; ++k)
                if (k < K) gu[i * K + k] = ins[k] ? -f * (((abits >> k) & 1u ? 1.0f : 0.f) - pk[k]) : 0.f;
        }
    }
    __syncthreads();
    int* out = counts + (int64_t)blockIdx.x * 4 * J;
    for (int e = tid; e < 4 * J; e += CS_THREADS) out[e] = cnt[e];
}

// gitem = d LOSS / d [g_un: J | s_un: J] from the (eta, y) counts summed over the blocks (integers: any order)
__global__ void k_cdm_sf_items(int J, int n_blocks, float scale, const int* __restrict__ counts, const float* __restrict__ g_un,
                               const float* __restrict__ s_un, float* __restrict__ gitem) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= J) return;
    long long n[4] = {0, 0, 0, 0};
    for (int b = 0; b < n_blocks; ++b)
        for (int c = 0; c < 4; ++c) n[c] += counts[((int64_t)b * 4 + c) * J + j];
    const float g = sigmoidf_(g_un[j]), s = sigmoidf_(s_un[j]);
    // eta = 0: P = g;  eta = 1: P = 1 - s;  d log-lik / dP = y / P - (1 - y) / (1 - P) inside the clamp, else 0
    const float P0 = fminf(fmaxf(g, VX_EPS32), 1.0f - VX_EPS32), P1 = fminf(fmaxf(1.0f - s, VX_EPS32), 1.0f - VX_EPS32);
    const bool in0 = g >= VX_EPS32 && g <= 1.0f - VX_EPS32, in1 = (1.0f - s) >= VX_EPS32 && (1.0f - s) <= 1.0f - VX_EPS32;
    const float d0 = in0 ? (float)n[1] / P0 - (float)n[0] / (1.0f - P0) : 0.f;
    const float d1 = in1 ? (float)n[3] / P1 - (float)n[2] / (1.0f - P1) : 0.f;
    gitem[j] = -scale * d0 * g * (1.0f - g);
    gitem[J + j] = scale * d1 * s * (1.0f - s);                      // dP / ds = -1
}

// ---- BinEncoder (vi.py:458-470): h = softplus(W1 yin + b1), u = W2 h + b2 (the logits of the attribute probabilities).
// One block = 256 / HS persons x HS hidden-unit slots (H <= HS, HS = 64 or 128); yin = the response bytes as they are
// (0 / 1; 255 -> -1).
template <int HS>
__global__ __launch_bounds__(256) void k_bin_enc_fwd(int K, int J, int H, int64_t nb, const uint8_t* __restrict__ y,
                                                     const int64_t* __restrict__ rows, const float* __restrict__ W1,
                                                     const float* __restrict__ b1, const float* __restrict__ W2,
                                                     const float* __restrict__ b2, float* __restrict__ h, float* __restrict__ u) {
    constexpr int PPB = 256 / HS;
    __shared__ float hs[PPB][HS];
    const int tid = threadIdx.x, hh = tid % HS, sub = tid / HS;
    for (int64_t i0 = (int64_t)blockIdx.x * PPB; i0 < nb; i0 += (int64_t)gridDim.x * PPB) {
        const int64_t i = i0 + sub;
        float acc = 0.f;
        if (i < nb && hh < H) {
            const int64_t row = rows ? rows[i] : i;
            const uint8_t* yr = y + row * J;
            acc = b1[hh];
            for (int j = 0; j < J; ++j) acc = fmaf(W1[(int64_t)hh * J + j], (float)(int8_t)yr[j], acc);
            acc = softplusf_(acc);
            h[i * H + hh] = acc;
        }
        hs[sub][hh] = (hh < H) ? acc : 0.f;
        __syncthreads();
        if (i < nb && hh < K) {
            float a = b2[hh];
            for (int t = 0; t < H; ++t) a = fmaf(W2[hh * H + t], hs[sub][t], a);
            u[i * K + hh] = a;
        }
        __syncthreads();
    }
}

// ghpre[i][hh] = (sum_k gu[i][k] W2[k][hh]) (1 - exp(-h)),  head-gradient slab per block: [W2: K*H | b2: K]  (d LOSS)
template <int HS>
__global__ __launch_bounds__(256) void k_bin_enc_bwd_small(int K, int H, int64_t nb, const float* __restrict__ W2,
                                                           const float* __restrict__ h, const float* __restrict__ gu,
                                                           float* __restrict__ ghpre, float* __restrict__ slabs) {
    constexpr int PPB = 256 / HS;
    __shared__ float red[PPB][HS];
    const int tid = threadIdx.x, hh = tid % HS, sub = tid / HS;
    float gw[CS_MAXK], gb[CS_MAXK];
#pragma unroll
    for (int k = 0; k < CS_MAXK; ++k) { gw[k] = 0.f; gb[k] = 0.f; }
    for (int64_t i = (int64_t)blockIdx.x * PPB + sub; i < nb; i += (int64_t)gridDim.x * PPB) {
        const float hv = hh < H ? h[i * H + hh] : 0.f;
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < CS_MAXK; ++k)
            if (k < K) {
                const float gk = gu[i * K + k];
                if (hh < H) a = fmaf(gk, W2[k * H + hh], a);
                gw[k] = fmaf(gk, hv, gw[k]);
                if (hh == 0) gb[k] += gk;
            }
        if (hh < H) ghpre[i * H + hh] = a * (1.0f - __expf(-hv));
    }
    float* slab = slabs + (int64_t)blockIdx.x * (K * H + K);
    for (int k = 0; k < K; ++k) {
        red[sub][hh] = gw[k];
        __syncthreads();
        if (sub == 0 && hh < H) {
            float t = red[0][hh];
#pragma unroll
            for (int q = 1; q < PPB; ++q) t += red[q][hh];
            slab[k * H + hh] = t;
        }
        __syncthreads();
        red[sub][hh] = gb[k];
        __syncthreads();
        if (tid == 0) {
            float t = red[0][0];
#pragma unroll
            for (int q = 1; q < PPB; ++q) t += red[q][0];
            slab[K * H + k] = t;
        }
        __syncthreads();
    }
}

// leave-one-out control variate over the S particles of a step: out[i] = mean_{s' != s} log_r[s'][i]
__global__ void k_loo_baseline(const float* __restrict__ lr_all /*[S][nb]*/, int S, int64_t nb, int s, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += (int64_t)gridDim.x * blockDim.x) {
        float acc = 0.f;
        for (int t = 0; t < S; ++t)
            if (t != s) acc += lr_all[(int64_t)t * nb + i];
        out[i] = acc / (float)(S - 1);
    }
}
