// HO-DINA with exact enumeration, the pattern-lattice contractions on the bf16 MFMA (VCHoDina.model / .guide,
// vi.py:897-934; dina vi.py:69-83; pattern table vi.py:825-837) -- the same outputs as k_hodina (k_hodina.hip), for
// 5 <= K <= 8 attributes (C = 2^K = 32 NT patterns) and J <= 32 items.
//
// k_hodina walks one person per wave with the patterns spread over the lanes, and every reduction over the patterns is a
// cross-lane exchange.  Here a wave takes 32 persons at once and the reference's C x B x J tensor contractions are what
// they are, small GEMMs with a constant 0/1 operand:
//
//   prior    A[c][n]  = S0_n + sum_k alpha[c][k] t_k[n]                (t = theta lam1 + lam0,  S0 = sum_k log(1 - pi_k))
//   items    F[c][n]  = lg[c][n] + base_n + sum_j eta[c][j] delta_j[n]  (delta = log Bern(y; 1 - s) - log Bern(y; g))
//   d/d item E[j][n]  = sum_c eta[c][j] r[c][n]                         (r = softmax_c F: the posterior over the patterns)
//   d/d attr T[k][n]  = sum_c alpha[c][k] rm[c][n],  T[8] = sum_c rm    (rm = r where the prior is not clamped)
//
// on v_mfma_f32_32x32x16_bf16 with the PATTERNS as the rows of the forward tiles and the PERSONS as the columns: a lane
// then holds 16 NT patterns of ONE person (its partner lane + 32 the other half), so the softmax over the patterns is
// in-lane arithmetic plus one v_permlane32_swap, and the forward accumulators are, as they stand, the B operand of the
// backward products (k index permuted -- the constant operand is laid out to match).  alpha / eta are exact in bf16;
// the person-side operands are split into bf16 terms (three forward: fp32-exact products; two backward: 2^-17 relative,
// below the accumulation noise of the gradient sums).
//
// The prior is Categorical(probs) with torch's clamp of the normalised probabilities to [eps, 1 - eps]; in the log
// domain that is a clamp of A (the probabilities sum to one by construction: prod_k (pi_k + 1 - pi_k)), and a clamped
// pattern passes no gradient to the attribute side.
#pragma once
#include "k_hodina.hip"

#define HM_THREADS 256
#define HM_WAVES 4
#define HM_YS 48                                                          // LDS row stride of a person's response bytes

__host__ __device__ inline size_t hm_lds_bytes(int NT) {
    const size_t img = (size_t)NT * 2 * 1024 * 4;                        // FA | FE | BE | BA fragment images
    const size_t red = (size_t)HM_WAVES * 64 * 40 * 4;                    // end-of-kernel reduction (aliases the images)
    return (img > red ? img : red) + 32 * 3 * 16 + (size_t)HM_WAVES * 32 * HM_YS;
}

// two bf16 terms of eight fp32 values (round to nearest): v = hi + mid up to 2^-17 relative
__device__ __forceinline__ void hm_split2(const float (&v)[8], bf16x8& fh, bf16x8& fm) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)v[j];
        fh[j] = h;
        fm[j] = (__bf16)(v[j] - (float)h);
    }
}

template <int NT>
__global__ __launch_bounds__(HM_THREADS, 2) void k_hodina_m(
    HoDinaDims dm, const uint8_t* __restrict__ y, const int64_t* __restrict__ rows, int64_t gid0,
    const float* __restrict__ loc, const float* __restrict__ raw, const float* __restrict__ eps_in,
    uint64_t seed, uint32_t step, uint32_t stream, const float* __restrict__ q, const float* __restrict__ lam0,
    const float* __restrict__ lam1_un, const float* __restrict__ g_un, const float* __restrict__ s_un,
    float* __restrict__ gloc, float* __restrict__ graw, float* __restrict__ elbo, float* __restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    if (dm.step_dev) step = *dm.step_dev;                                 // replayed from a HIP graph: the counter lives on the device
    const int K = dm.K, J = dm.J;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 31, h = lane >> 5;
    constexpr size_t IMG = (size_t)NT * 2 * 1024;
    constexpr size_t IMG4 = IMG * 4 > (size_t)HM_WAVES * 64 * 40 * 4 ? IMG * 4 : (size_t)HM_WAVES * 64 * 40 * 4;
    uint4* FA = (uint4*)smem_raw;                                         // [T][s][lane]   prior, forward
    uint4* FE = (uint4*)(smem_raw + IMG);                                 // [T][q][lane]   items, forward
    uint4* BE = (uint4*)(smem_raw + 2 * IMG);                             // [2 T + s][lane] items, backward
    uint4* BA = (uint4*)(smem_raw + 3 * IMG);                             // [2 T + s][lane] attributes, backward
    float4* tabI = (float4*)(smem_raw + IMG4);                            // [32 items][3 classes]: delta, lp0, d0, d1
    uint8_t* ybuf = smem_raw + IMG4 + 32 * 3 * 16 + (size_t)wave * 32 * HM_YS;

    // ---- constants of the step, once per block ----------------------------------------------------------------------
    for (int e = tid; e < 32 * 3; e += HM_THREADS) {
        const int j = e / 3, cls = e - 3 * j;
        float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < J) {
            const float gj = fminf(sigmoidf_(g_un[j]), 1.0f - VX_EPS32), og = fmaxf(sigmoidf_(-g_un[j]), VX_EPS32);
            const float sj = fminf(sigmoidf_(s_un[j]), 1.0f - VX_EPS32), os = fmaxf(sigmoidf_(-s_un[j]), VX_EPS32);
            const unsigned yy = cls == 2 ? 255u : (unsigned)cls;
            float lp0, lp1, d0, d1;
            bern_const(gj, og, yy, lp0, d0);
            bern_const(os, sj, yy, lp1, d1);
            t4 = make_float4(lp1 - lp0, lp0, d0, d1);
        }
        tabI[e] = t4;
    }
    for (int e = tid; e < NT * 2 * 64; e += HM_THREADS) {
        const int l = e & 63, s = (e >> 6) & 1, T = e >> 7, r = l & 31, hh = l >> 5;
        uint32_t fa[4] = {0, 0, 0, 0}, fe[4] = {0, 0, 0, 0}, be[4] = {0, 0, 0, 0}, ba[4] = {0, 0, 0, 0};
        const int c_row = 32 * T + r;                                     // forward: the lane's pattern
        int qp_row = -1;                                                  // backward: the lane's item
        if (r < J) {
            qp_row = 0;
            for (int k = 0; k < K; ++k)
                if (q[(int64_t)k * J + r] != 0.f) qp_row |= (1 << k);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t one = (j & 1) ? 0x3F800000u : 0x00003F80u;
            // forward, prior: k index 16 s + 8 hh + j = (term, attribute j); the fourth octet is padding
            if (!(s == 1 && hh == 1) && j < K && ((c_row >> j) & 1)) fa[j >> 1] |= one;
            // forward, items: k index = item 16 s + 8 hh + j
            const int it = 16 * s + 8 * hh + j;
            if (it < J) {
                int qp = 0;
                for (int k = 0; k < K; ++k)
                    if (q[(int64_t)k * J + it] != 0.f) qp |= (1 << k);
                const bool eta = dm.dino ? (__popc(qp) >= 2 && (c_row & qp) != 0) : ((c_row & qp) == qp);
                if (eta) fe[j >> 1] |= one;
            }
            // backward: k index j of k-step 2 T + s = pattern 32 T + 16 s + 8 (j >> 2) + 4 hh + (j & 3) (the C layout)
            const int c = 32 * T + 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3);
            if (qp_row >= 0) {
                const bool eta = dm.dino ? (__popc(qp_row) >= 2 && (c & qp_row) != 0) : ((c & qp_row) == qp_row);
                if (eta) be[j >> 1] |= one;
            }
            if ((r < K && ((c >> r) & 1)) || r == 8 || r == 12) ba[j >> 1] |= one;   // rows 8 / 12: sum over the patterns
        }
        FA[e] = make_uint4(fa[0], fa[1], fa[2], fa[3]);
        FE[e] = make_uint4(fe[0], fe[1], fe[2], fe[3]);
        BE[e] = make_uint4(be[0], be[1], be[2], be[3]);
        BA[e] = make_uint4(ba[0], ba[1], ba[2], ba[3]);
    }
    __syncthreads();

    // attribute constants: this lane's four attributes 4 h + m (backward rows), all eight for the forward operand
    float l0a[8], l1a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        l0a[k] = k < K ? lam0[k] : 0.f;
        l1a[k] = k < K ? __expf(lam1_un[k]) : 0.f;
    }
    float gg[16], gs[16], gl0[4], gl1[4];
#pragma unroll
    for (int r = 0; r < 16; ++r) gg[r] = gs[r] = 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m) gl0[m] = gl1[m] = 0.f;
    const float LE = -15.942385f /* log eps32 */, LH = -1.1920929e-07f /* log(1 - eps32) */;
    const float L2E = 1.4426950408889634f;

    const int64_t n_groups = (dm.nb + 31) / 32;
    const int64_t n_waves = (int64_t)gridDim.x * HM_WAVES;
    // the inputs of a group are fetched one group ahead: a wave is alone on its SIMD (the accumulators take the register
    // file), so nothing else would hide the latency of the loads
    struct Fetch { int64_t row; float lc, rw, e; uint32_t w[4]; bool valid; };
    auto fetch = [&](int64_t grp) {
        Fetch f;
        f.row = 0; f.lc = 0.f; f.rw = 0.f; f.e = 0.f;
        f.w[0] = f.w[1] = f.w[2] = f.w[3] = 0xFFFFFFFFu;
        const int64_t i = grp * 32 + n;
        f.valid = grp < n_groups && i < dm.nb;
        if (f.valid) {
            f.row = rows ? rows[i] : i;
            f.lc = loc[i]; f.rw = raw[i];
            if (eps_in) f.e = eps_in[i];
            const uint8_t* yr = y + f.row * J;                            // items 16 h .. 16 h + 15 by this lane
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                const int it = 16 * h + b;
                const uint32_t v = it < J ? (uint32_t)yr[it] : 255u;
                f.w[b >> 2] = (f.w[b >> 2] & ~(0xFFu << (8 * (b & 3)))) | (v << (8 * (b & 3)));
            }
        }
        return f;
    };
    int64_t grp = (int64_t)blockIdx.x * HM_WAVES + wave;
    Fetch nxt = fetch(grp);
    for (; grp < n_groups; grp += n_waves) {
        const Fetch cur = nxt;
        nxt = fetch(grp + n_waves);
        const int64_t i = grp * 32 + n;
        const bool valid = cur.valid;
        const float lc = cur.lc, rw = cur.rw;
        const float e = (valid && !eps_in) ? philox_normal4(seed, step, stream, gid0 + cur.row, 0u)[0] : cur.e;
        const float sig = __expf(rw);
        const float th = lc + sig * e;
        // ---- the person's responses, shared with the partner lane through LDS
        __builtin_amdgcn_wave_barrier();                                  // the previous group's readers are done
        *(uint4*)(ybuf + n * HM_YS + 16 * h) = make_uint4(cur.w[0], cur.w[1], cur.w[2], cur.w[3]);
        __builtin_amdgcn_wave_barrier();
        // ---- item side of the forward operand: delta of items 8 h + j and 16 + 8 h + j, three bf16 terms each
        bf16x8 Bd[2][3];
        float base = 0.f;
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            const uint2 yb = *(const uint2*)(ybuf + n * HM_YS + 16 * qq + 8 * h);
            float dv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t yy = ((j < 4 ? yb.x : yb.y) >> (8 * (j & 3))) & 0xFFu;
                const int cls = yy == 255u ? 2 : (yy != 0u ? 1 : 0);
                const float4 t4 = tabI[(16 * qq + 8 * h + j) * 3 + cls];
                dv[j] = t4.x;
                base += t4.y;
            }
            split3_frag(dv, Bd[qq][0], Bd[qq][1], Bd[qq][2]);
        }
        base = half_sum32(base);
        // ---- attribute side: t_k = theta lam1_k + lam0_k, S0 = sum_k log(1 - pi_k)
        float tk[8], S0 = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            tk[k] = k < K ? fmaf(th, l1a[k], l0a[k]) : 0.f;
            if (k < K) S0 -= softplusf_(tk[k]);
        }
        bf16x8 t_h, t_m, t_l, Bt0, Bt1;
        split3_frag(tk, t_h, t_m, t_l);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            Bt0[j] = h ? t_m[j] : t_h[j];
            Bt1[j] = h ? (__bf16)0.f : t_l[j];
        }
        // ---- pass 1: the maximum over the patterns of F = log prior + log likelihood.  F is not kept: 16 NT values per
        // lane would take the register file and leave one wave per SIMD; the forward products are cheap (8 MFMAs a
        // tile) and are simply made again in pass 2
        auto forward_tile = [&](int T, f32x16& a, f32x16& b) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { a[r] = S0; b[r] = base; }
            a = mfma_bf16(__builtin_bit_cast(bf16x8, FA[(T * 2 + 0) * 64 + lane]), Bt0, a);
            a = mfma_bf16(__builtin_bit_cast(bf16x8, FA[(T * 2 + 1) * 64 + lane]), Bt1, a);
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const bf16x8 fe = __builtin_bit_cast(bf16x8, FE[(T * 2 + qq) * 64 + lane]);
                b = mfma_bf16(fe, Bd[qq][0], b);
                b = mfma_bf16(fe, Bd[qq][1], b);
                b = mfma_bf16(fe, Bd[qq][2], b);
            }
        };
        // (pass 1 only has to land near the maximum -- it is the reference point of the exponentials, and the log-sum-exp
        // below is exact for any reference -- so it uses the leading bf16 term of each operand: 3 MFMAs a tile)
        float mx = -3.0e38f;
#pragma unroll 1
        for (int T = 0; T < NT; ++T) {
            f32x16 a, b;
#pragma unroll
            for (int r = 0; r < 16; ++r) { a[r] = S0; b[r] = base; }
            a = mfma_bf16(__builtin_bit_cast(bf16x8, FA[(T * 2 + 0) * 64 + lane]), Bt0, a);
            b = mfma_bf16(__builtin_bit_cast(bf16x8, FE[(T * 2 + 0) * 64 + lane]), Bd[0][0], b);
            b = mfma_bf16(__builtin_bit_cast(bf16x8, FE[(T * 2 + 1) * 64 + lane]), Bd[1][0], b);
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, __builtin_amdgcn_fmed3f(a[r], LE, LH) + b[r]);
        }
        mx = fmaxf(mx, half_swap32(mx));
        // (Rounds 2-5 kept a sched_barrier here: with one tile the passes are straight-line code, the scheduler slides pass 2's
        // products up between pass 1's reads, and the gradients came out wrong.  What was wrong was not the interleaving but
        // the swap's scratch register inside an accumulator in flight -- vx_common.h, permlane32_swap; docs/HARDWARE.md rule 40.)
        const float moff = -mx * L2E;                                     // exp(F - mx') = exp2(F log2 e + moff), mx' = -moff / log2 e
        // ---- pass 2: posterior weights and the backward products; the forward accumulators are the B operand as they stand
        f32x16 aE = zero16(), aT = zero16();
        float rs = 0.f;
#pragma unroll 1
        for (int T = 0; T < NT; ++T) {
            f32x16 a, b;
            forward_tile(T, a, b);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float rv[8];
                uint32_t keep[4];                                         // 0xFFFF per bf16 element whose prior is not clamped
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float av = a[8 * s + j];
                    const float lg = __builtin_amdgcn_fmed3f(av, LE, LH);   // log of the clamped prior probability
                    const float rc = __builtin_amdgcn_exp2f(fmaf(lg + b[8 * s + j], L2E, moff));
                    rs += rc;
                    rv[j] = rc;
                    const uint32_t kk = (lg == av) ? ((j & 1) ? 0xFFFF0000u : 0x0000FFFFu) : 0u;
                    if (j & 1) keep[j >> 1] |= kk; else keep[j >> 1] = kk;
                }
                bf16x8 r_h, r_m;
                hm_split2(rv, r_h, r_m);
                typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
                u32x4v mh = __builtin_bit_cast(u32x4v, r_h), mm = __builtin_bit_cast(u32x4v, r_m);
#pragma unroll
                for (int w2 = 0; w2 < 4; ++w2) { mh[w2] &= keep[w2]; mm[w2] &= keep[w2]; }   // a clamped prior passes no gradient
                const bf16x8 be = __builtin_bit_cast(bf16x8, BE[(2 * T + s) * 64 + lane]);
                const bf16x8 ba = __builtin_bit_cast(bf16x8, BA[(2 * T + s) * 64 + lane]);
                aE = mfma_bf16(be, r_h, aE);
                aE = mfma_bf16(be, r_m, aE);
                aT = mfma_bf16(ba, __builtin_bit_cast(bf16x8, mh), aT);
                aT = mfma_bf16(ba, __builtin_bit_cast(bf16x8, mm), aT);
            }
        }
        rs = half_sum32(rs);
        const float rinv = 1.0f / rs;
        const float lse = (__builtin_amdgcn_logf(rs) - moff) * 0.6931471805599453f;   // log-sum-exp about mx' = -moff / log2 e
        // ---- d / d item: rows (r & 3) + 8 (r >> 2) + 4 h of E are this lane's items, the column its person
        if (valid) {
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const uint32_t yw = *(const uint32_t*)(ybuf + n * HM_YS + 8 * qd + 4 * h);
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int it = m + 8 * qd + 4 * h;
                    const uint32_t yy = (yw >> (8 * m)) & 0xFFu;
                    const int cls = yy == 255u ? 2 : (yy != 0u ? 1 : 0);
                    const float4 t4 = tabI[it * 3 + cls];
                    const float E = aE[4 * qd + m] * rinv;
                    gg[4 * qd + m] = fmaf(1.0f - E, t4.z, gg[4 * qd + m]);   // d/dg through patterns that do NOT master the item
                    gs[4 * qd + m] = fmaf(-E, t4.w, gs[4 * qd + m]);           // d/ds: p = 1 - s
                }
            }
        }
        // ---- d / d attribute: rows 0..3 (+ 4 h) of T, the pattern sum in row 8 / 12 (register 4)
        float gthp = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float tkm = h ? tk[4 + m] : tk[m];
            const float pik = sigmoidf_(tkm);
            const float tau = (4 * h + m < K) ? (aT[m] - pik * aT[4]) * rinv : 0.f;
            if (valid) { gl0[m] += tau; gl1[m] = fmaf(tau, th, gl1[m]); }
            gthp = fmaf(tau, h ? l1a[4 + m] : l1a[m], gthp);
        }
        const float gth = half_sum32(gthp) - th;                          // prior N(0, 1)
        if (valid && h == 0) {
            const float gt = dm.scale * gth;
            gloc[i] = -gt;
            graw[i] = -(gt * sig * e + dm.scale);
            elbo[i] = lse - 0.5f * th * th + 0.5f * e * e + rw;
        }
    }
    // ---- block reduction -> one slab per block [g_un: J | s_un: J | lam0: K | lam1_un: K], fixed order
    __syncthreads();
    float* red = (float*)smem_raw + ((size_t)wave * 64 + lane) * 40;
#pragma unroll
    for (int r = 0; r < 16; ++r) { red[r] = gg[r]; red[16 + r] = gs[r]; }
#pragma unroll
    for (int m = 0; m < 4; ++m) { red[32 + m] = gl0[m]; red[36 + m] = gl1[m]; }
    __syncthreads();
    const int len = 2 * J + 2 * K;
    float* slab = slabs + (int64_t)blockIdx.x * len;
    for (int o = tid; o < len; o += HM_THREADS) {
        int hh, slot;
        float chain;
        if (o < 2 * J) {
            const int j = o < J ? o : o - J;
            hh = (j >> 2) & 1;
            slot = (o < J ? 0 : 16) + (j & 3) + 4 * (j >> 3);
            if (o < J) chain = fminf(sigmoidf_(g_un[j]), 1.0f - VX_EPS32) * fmaxf(sigmoidf_(-g_un[j]), VX_EPS32);
            else chain = fminf(sigmoidf_(s_un[j]), 1.0f - VX_EPS32) * fmaxf(sigmoidf_(-s_un[j]), VX_EPS32);
        } else {
            const int k = (o - 2 * J) < K ? (o - 2 * J) : (o - 2 * J - K);
            hh = k >> 2;
            slot = ((o - 2 * J) < K ? 32 : 36) + (k & 3);
            chain = (o - 2 * J) < K ? 1.0f : __expf(lam1_un[k]);
        }
        float acc = 0.f;
        for (int w = 0; w < HM_WAVES; ++w)
            for (int p = 0; p < 32; ++p) acc += ((const float*)smem_raw)[((size_t)w * 64 + 32 * hh + p) * 40 + slot];
        slab[o] = dm.scale * acc * chain;
    }
}
