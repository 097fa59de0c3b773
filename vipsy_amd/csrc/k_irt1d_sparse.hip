// D = 1 IRT on OBSERVED cells only (BASELINE config 4: 90 % of the responses missing).
//
// The dense kernel (k_irt1d.hip) evaluates every cell, and with one latent dimension the cell math (exp2 / rcp / log2
// + ~20 VALU) is the bound, not HBM.  The responses never change between steps, so the host compacts them once
// (vipsy_amd/engine.py::_sparse_lists) and each step touches the observed ~10 % only, in ONE pass:
//
//   slot    lane = one person slot, a wave = a group of 64 slots.  pidx[slot] names the person (-1 = empty): the host
//           orders the persons of a window by their number of observed cells, so the 64 lists of a group have (almost)
//           the same length and no lane idles through another lane's tail.
//   lists   pent [n_groups][Lq][64] of 8-byte quads: four uint16 codes  item | y << 15  (0xFFFF = past the end of the
//           list), so a wave reads 512 contiguous bytes per four cells and the next quad is in flight while the current
//           one is evaluated.
//   items   (Dc a_j, Dc b_j) gathered from LDS as one 8-byte read.
//   d/d item  accumulated per block in LDS with INTEGER atomics, so the sums do not depend on the order the hardware
//           retires them (run-to-run bit-reproducible).  For a cell t = dlp/dz with |t| <= 1 always, so the pair
//           (t x, t) is quantised to two 32-bit integers with scales the host derives from the persons a block can see
//           (no overflow by construction) and added as ONE 64-bit atomic  q_a 2^32 + q_b.  A person with |x| > SP_XB
//           (8 prior standard deviations) takes a separate full-range 64-bit fixed-point slot for t x instead.  The
//           3PL / 4PL asymptote gradients are unbounded (1 / P) and use full-range slots as well.
//           One slab [a | b | c | d][J] (+ the block's ELBO share) per block, summed by k_reduce_wide.
//
// A missing cell contributes the reference's constant log Bern(0 | clamp 0) (vi.py:621-624) and no gradient: added as
// (J - observed) * constant per person.
#pragma once
#include "vx_common.h"

#define SP_THREADS 512
#define SP_XB 8.0f

struct Irt1dSpDims {
    int J, model, Lq;
    float Dc, scale;
    float sb, inv_sb;                 // quantisation of t  (power of two); t x uses sb / SP_XB
    int64_t n_groups;
};

__device__ __forceinline__ int sp_rint(float v) { return (int)__builtin_rintf(v); }

template <int MODEL>
__global__ __launch_bounds__(SP_THREADS) void k_irt1d_sp(
    Irt1dSpDims dm, const uint2* __restrict__ pent /*[n_groups][Lq][64]*/, const int32_t* __restrict__ glen,
    const int32_t* __restrict__ pidx, int64_t gid0, const float* __restrict__ loc, const float* __restrict__ raw,
    const float* __restrict__ eps_in, uint64_t seed, uint32_t step, const uint32_t* __restrict__ step_dev,
    uint32_t stream, const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c_un,
    const float* __restrict__ d_un, float* __restrict__ gloc, float* __restrict__ graw, float* __restrict__ elbo,
    float* __restrict__ slabs) {
    if (step_dev) step = *step_dev;                                    // replayed from a HIP graph: the counter lives on the device
    // Adam's count of this step for a fused optimiser tail (k_reduce_adam reads it; nobody in its launch reads step_dev)
    if (blockIdx.x == 0 && threadIdx.x == 0) ((uint32_t*)slabs)[(size_t)gridDim.x * (4 * (size_t)dm.J + 1)] = step + 1u;
    __shared__ float el_w[SP_THREADS / 64];
    float el_acc = 0.f;
    // LDS: ab [J] float2 | acc [J] packed (t x, t) | big [J] fixed point t x | (3PL+) accc, accd [J], cs, ds, os [J] floats
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int J = dm.J;
    float2* ab = (float2*)smem_raw;
    long long* acc = (long long*)(ab + J);
    long long* big = acc + J;
    long long* accc = big + J;
    long long* accd = accc + J;
    float* cs = (float*)(accd + J);
    float* ds = cs + J;
    float* os = ds + J;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int j = tid; j < J; j += SP_THREADS) {
        ab[j] = make_float2(dm.Dc * ((MODEL >= 2) ? a[j] : 1.0f), dm.Dc * b[j]);
        acc[j] = 0; big[j] = 0;
        if (MODEL >= 3) {
            accc[j] = 0; accd[j] = 0;
            cs[j] = fminf(sigmoidf_(c_un[j]), 1.0f - VX_EPS32);
            ds[j] = (MODEL >= 4) ? fminf(sigmoidf_(d_un[j]), 1.0f - VX_EPS32) : 1.0f;
            os[j] = (MODEL >= 4) ? fmaxf(sigmoidf_(-d_un[j]), VX_EPS32) : 0.f;
        }
    }
    __syncthreads();
    const float sb = dm.sb, sa = dm.sb * (1.0f / SP_XB);
    const int64_t n_waves = (int64_t)gridDim.x * (SP_THREADS / 64);
    for (int64_t grp = (int64_t)blockIdx.x * (SP_THREADS / 64) + wave; grp < dm.n_groups; grp += n_waves) {
        const int i = pidx[grp * 64 + lane];
        const bool valid = i >= 0;
        float l = 0.f, r = 0.f, e = 0.f;
        if (valid) {
            l = loc[i]; r = raw[i];
            e = eps_in ? eps_in[i] : philox_normal4(seed, step, stream, gid0 + i, 0u)[0];
        }
        const float sig = __expf(r);
        const float x = l + sig * e;
        const bool far = fabsf(x) > SP_XB;                             // practically never: full-range slot for t x
        const float xs = far ? 0.f : x * sa;
        const int nq = __builtin_amdgcn_readfirstlane(glen[grp]);      // quads in the longest list of the group
        const uint2* ent = pent + grp * (int64_t)dm.Lq * 64 + lane;
        float ll = 0.f, gx = 0.f;
        int nobs = 0;
        uint2 cur = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
        if (nq > 0) cur = ent[0];
        for (int q = 0; q < nq; ++q) {
            uint2 nxt = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
            if (q + 1 < nq) nxt = ent[(int64_t)(q + 1) * 64];          // wave-uniform: the next quad is in flight
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t word = (k < 2) ? cur.x : cur.y;
                const uint32_t code = (k & 1) ? (word >> 16) : (word & 0xFFFFu);
                if (code != 0xFFFFu) {                                 // lists of a group end together (sorted): rare
                    const int j = (int)(code & 0x7FFFu);
                    const bool y1 = (code & 0x8000u) != 0;
                    const float2 p = ab[j];
                    const float z = fmaf(x, p.x, p.y);                 // Dc (a x + b)
                    float t;                                           // dlp/dz
                    if (MODEL <= 2) {
                        // Bernoulli(logit z) with the reference's clamp of the probability to [eps, 1 - eps]:
                        // with w = -z for y = 1 and z for y = 0,  lp = -softplus(w),  dlp/dz = +-sigmoid(w)
                        const float ZL = 15.942384719848633f;          // logit(1 - eps32)
                        const float w = y1 ? -z : z;
                        const float wc = __builtin_amdgcn_fmed3f(w, -ZL, ZL);
                        const float ex = __builtin_amdgcn_exp2f(-1.4426950408889634f * fabsf(wc));
                        const float u = 1.0f + ex;
                        const float rc = fast_rcp(u);
                        ll -= fmaf(__builtin_amdgcn_logf(u), 0.6931471805599453f, fmaxf(wc, 0.f));
                        const float sg = (wc >= 0.f) ? rc : ex * rc;   // sigmoid(wc)
                        const float s0 = (wc == w) ? sg : 0.f;         // zero gradient where the clamp is active
                        t = y1 ? s0 : -s0;
                    } else {
                        float lp, dz, dc, dd;
                        irt_cell<MODEL>(z, y1 ? 1u : 0u, cs[j], ds[j], os[j], lp, dz, dc, dd);
                        ll += lp;
                        t = dz;
                        fx_add(&accc[j], dc);
                        if (MODEL >= 4) fx_add(&accd[j], dd);
                    }
                    gx = fmaf(t, p.x, gx);
                    nobs += 1;
                    // (t x, t) as one 64-bit integer add: q_a 2^32 + q_b
                    const int qb = sp_rint(t * sb), qa = sp_rint(t * xs);
                    const unsigned long long pk = ((unsigned long long)(unsigned)(qa + (qb >> 31)) << 32) | (unsigned)qb;
                    atomicAdd((unsigned long long*)&acc[j], pk);
                    if (far) fx_add(&big[j], t * x);
                }
            }
            cur = nxt;
        }
        if (valid) {
            ll += (float)(J - nobs) * VX_LOGP_MISSING;
            const float gxt = dm.scale * (gx - x);                     // d ELBO / d x (likelihood + prior)
            gloc[i] = -gxt;
            graw[i] = -(gxt * sig * e + dm.scale);
            const float el = ll - 0.5f * x * x + 0.5f * e * e + r;
            elbo[i] = el;
            el_acc += el;
        }
    }
    el_acc = wave_sum_dpp(el_acc);
    if (lane == 0) el_w[wave] = el_acc;
    __syncthreads();
    float* slab = slabs + (int64_t)blockIdx.x * (4 * J + 1);
    if (tid == 0) {                                                    // column 4 J: this block's share of the ELBO
        float acc = el_w[0];
#pragma unroll
        for (int w = 1; w < SP_THREADS / 64; ++w) acc += el_w[w];
        slab[4 * J] = dm.scale * acc;
    }
    const float ua = dm.Dc * dm.scale * dm.inv_sb * SP_XB, ub = dm.Dc * dm.scale * dm.inv_sb;
    for (int j = tid; j < J; j += SP_THREADS) {
        const long long s = acc[j];
        const int qb = (int)(s & 0xFFFFFFFFll);
        const int qa = (int)((s - (long long)qb) >> 32);
        slab[j] = (MODEL >= 2) ? fmaf((float)qa, ua, dm.Dc * dm.scale * fx_get(big[j])) : 0.f;
        slab[J + j] = (float)qb * ub;
        slab[2 * J + j] = (MODEL >= 3) ? dm.scale * fx_get(accc[j]) : 0.f;
        slab[3 * J + j] = (MODEL >= 4) ? dm.scale * fx_get(accd[j]) : 0.f;
    }
}
