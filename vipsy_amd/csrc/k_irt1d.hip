// D = 1 IRT (irt_1pl..4pl with a Normal guide; vi.py:22-66, 588-595, 617-625, 684/705), one fused pass:
// x = loc + exp(raw) eps -> masked Bernoulli log-lik -> prior/entropy -> gradients.
//
// Layout (round 5): PERSON-PER-LANE.  A workgroup of four waves takes a chunk of 64 persons at a time; lane l of every wave is
// person 64 c + l, and the four waves divide the ITEMS between them (wave w: items w ceil(J / 4) ..).  The per-person sums
// (log-lik, d/dx) are plain in-lane accumulations -- no cross-lane traffic per person -- and meet through LDS once a chunk
// (two numbers a person and wave).  What crosses lanes is the ITEM gradient (a sum over the persons = over the lanes): eight
// or four items at a time the lanes park their terms (dz x, dz[, dc, dd]) in LDS, [item][person][quantity], and the wave reads
// them back transposed -- lane (item, quantity, a quarter of the persons) adds 16 of them, one ds_bpermute and one
// v_permlane32_swap join the quarters -- in a fixed order: bit-reproducible, no atomics.  ~35 LDS / vector instructions per
// sixteen (item, quantity) pairs beside ~55 of cell arithmetic an item.
// Item constants come from a table in LDS (one wave-uniform 16- or 32-byte read an item: broadcast, conflict-free).
//
// The form before it (rounds 1-4) put the ITEMS on the lanes (four per lane, two persons per wave iteration for J <= 128) and
// reduced log-lik and d/dx over the wave for every person: 100 items filled 25 of 32 lane slots, the two DPP reductions, the
// row-pointer broadcasts and the lane selects came to ~45 instructions per person pair beside 4 x 60 of cell arithmetic, and a
// wave owned whole groups of 64 persons -- 1 563 groups on 1 024 SIMDs at BASELINE config 2 (100 k persons x 100 items): 40 us
// where the cell arithmetic alone is ~20.  Here the work unit is (64 persons) x (a quarter of the items), 6 252 units at config 2.
#pragma once
#include "vx_common.h"

#define I1_THREADS 256
#define I1_WAVES (I1_THREADS / 64)
#define I1_PAIRS 16                                                   // (item, quantity) pairs per transposed reduction

struct Irt1dDims {
    int J, model;
    float Dc, scale;
    int64_t nb;
    // k_irt1d_items: persons a wave takes at a time (1..64).  It walks its group ONE PERSON AFTER THE OTHER (the items on the
    // lanes, ~1 us a person), so the reference's 100 rows a step in groups of 64 were two waves busy for 69 us: the group shrinks
    // with the batch until it fills the chip's waves (i1_group_size)
    int gsz = 64;
};

__host__ __device__ inline int i1_group_size(int64_t nb, int64_t max_waves) {
    int64_t g = (nb + max_waves - 1) / max_waves;
    return (int)(g < 1 ? 1 : g > 64 ? 64 : g);
}

// LDS of a workgroup (floats): item table [J][NPF] | item sums [J][NQ] | per wave: parked terms [16 / NQ][64 NQ + pad] (the
// wave's (ll, gx)[64] hand-over at the end of a chunk lies in the same region) | per-chunk hand-over: x[64], row[64] (two words)
// -- 22.2 KB at BASELINE config 2 (4PL, J = 100): seven workgroups a CU, so that its 1 563 chunks are all resident at once
__host__ __device__ inline int i1_npf(int model) { return model >= 3 ? 8 : 2; }
__host__ __device__ inline int i1_nq(int model) { return model >= 3 ? 4 : 2; }
__host__ __device__ inline int i1_scr_stride(int model) { return 64 * i1_nq(model) + (model >= 3 ? 4 : 2); }
__host__ __device__ inline size_t i1_lds_bytes(int J, int model) {
    return sizeof(float) * ((size_t)J * (i1_npf(model) + i1_nq(model)) + (size_t)I1_WAVES * (I1_PAIRS / i1_nq(model)) * i1_scr_stride(model) +
                            64 * 3);
}

template <int MODEL, bool WORDS>
__global__ __launch_bounds__(I1_THREADS) void k_irt1d(
    Irt1dDims dm, const uint8_t* __restrict__ y, const int64_t* __restrict__ rows, int64_t gid0,
    const float* __restrict__ loc, const float* __restrict__ raw, const float* __restrict__ eps_in,
    uint64_t seed, uint32_t step, const uint32_t* __restrict__ step_dev, uint32_t stream, const float* __restrict__ a,
    const float* __restrict__ b, const float* __restrict__ c_un, const float* __restrict__ d_un,
    float* __restrict__ gloc, float* __restrict__ graw, float* __restrict__ elbo, float* __restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NPF = MODEL >= 3 ? 8 : 2, NQ = MODEL >= 3 ? 4 : 2, SCR = 64 * NQ + (MODEL >= 3 ? 4 : 2);
    constexpr int PAIRS = I1_PAIRS, I1_RB = PAIRS / NQ;                 // items per transposed reduction: 8 (1PL / 2PL) or 4
    constexpr int GRP = 64 / PAIRS, NSRC = 64 / GRP;                   // lane = (item, quantity) x a group of NSRC = 16 persons
    const int J = dm.J;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* const ptab = smem;                                         // [J][NPF]: Dc a, Dc b (, c, d, 1 - d, pad)
    float* const acc = ptab + (size_t)J * NPF;                        // [J][NQ]: sums over this workgroup's persons
    float* const scr0 = acc + (size_t)J * NQ;                          // the waves' parked terms, I1_RB * SCR floats each
    float* const scr = scr0 + (size_t)wave * I1_RB * SCR;
    float* const xs = scr0 + (size_t)I1_WAVES * I1_RB * SCR;           // [64]
    uint32_t* const rws = (uint32_t*)(xs + 64);                        // [2][64]: the persons' response rows
    static_assert(I1_RB * SCR >= 128, "the (ll, gx) hand-over of a wave fits its parked-terms region");
    if (step_dev) step = *step_dev;                                    // replayed from a HIP graph: the counter lives on the device
    // Adam's count of this step for a fused optimiser tail (k_reduce_adam reads it; nobody in its launch reads step_dev)
    if (blockIdx.x == 0 && threadIdx.x == 0) ((uint32_t*)slabs)[(size_t)gridDim.x * (4 * (size_t)dm.J + 1)] = step + 1u;
    for (int j = tid; j < J; j += I1_THREADS) {
        ptab[j * NPF + 0] = dm.Dc * ((MODEL >= 2) ? a[j] : 1.0f);
        ptab[j * NPF + 1] = dm.Dc * b[j];
        if constexpr (MODEL >= 3) {
            ptab[j * NPF + 2] = fminf(sigmoidf_(c_un[j]), 1.0f - VX_EPS32);
            ptab[j * NPF + 3] = (MODEL >= 4) ? fminf(sigmoidf_(d_un[j]), 1.0f - VX_EPS32) : 1.0f;
            ptab[j * NPF + 4] = (MODEL >= 4) ? fmaxf(sigmoidf_(-d_un[j]), VX_EPS32) : 0.f;
            ptab[j * NPF + 5] = 0.f; ptab[j * NPF + 6] = 0.f; ptab[j * NPF + 7] = 0.f;
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[j * NQ + q] = 0.f;
    }
    // this wave's items
    const int ipw = (J + I1_WAVES - 1) / I1_WAVES;
    const int jw0 = wave * ipw < J ? wave * ipw : J, jw1 = jw0 + ipw < J ? jw0 + ipw : J;
    // the reduction's lane roles: pair = (item jj of the block, quantity q), group rg of NSRC consecutive persons.  A 4-byte LDS
    // read is served in two lane halves over 32 banks: the 16 pairs of a group fall on 16 different banks (item stride SCR = 4
    // or 2 mod 32), and the odd groups walk their persons from 16 / NQ further on, on the other 16
    const int pair = lane % PAIRS, rg = lane / PAIRS, rjj = pair / NQ, rq = pair % NQ;
    const int rrot = (rg & 1) * (16 / NQ);
    const int64_t n_chunks = (dm.nb + 63) / 64;
    float el_acc = 0.f;
    __syncthreads();
    for (int64_t ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
        const int64_t i = ch * 64 + lane;
        const bool valid = i < dm.nb;
        float l = 0.f, r = 0.f, e = 0.f, sig = 1.f, xv = 0.f;
        if (wave == 0) {                                               // the persons' scalars: once a chunk, handed over through LDS
            int64_t row = 0;
            if (valid) {
                row = rows ? rows[i] : i;
                l = loc[i]; r = raw[i];
                e = eps_in ? eps_in[i] : philox_normal4(seed, step, stream, gid0 + row, 0u)[0];
            }
            sig = __expf(r);
            xv = l + sig * e;
            xs[lane] = xv;
            rws[lane] = (uint32_t)row; rws[64 + lane] = (uint32_t)((uint64_t)row >> 32);
        }
        __syncthreads();
        const float x = xs[lane];
        const uint8_t* const yr = y + (int64_t)(((uint64_t)rws[64 + lane] << 32) | rws[lane]) * J;
        float llp = 0.f, gxp = 0.f;
        // response bytes of a block of eight items: WORDS (J % 4 == 0, rows 4-byte aligned): the (up to) three words that
        // hold them, realigned to the block's first item by v_alignbyte; otherwise byte loads
        uint32_t w0 = 0, w1 = 0, w2 = 0;
        auto load_block = [&](int jb) __attribute__((always_inline)) {
            if constexpr (WORDS) {
                const int wl = J / 4 - 1;
                const int k = jb >> 2;
                const uint32_t* yw = (const uint32_t*)yr;
                w0 = yw[k < wl ? k : wl]; w1 = yw[k + 1 < wl ? k + 1 : wl];
                if constexpr (I1_RB > 4) w2 = yw[k + 2 < wl ? k + 2 : wl];
            } else {
                uint32_t v0 = 0, v1 = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v0 |= (uint32_t)yr[jb + q < J ? jb + q : J - 1] << (8 * q);
                    if constexpr (I1_RB > 4) v1 |= (uint32_t)yr[jb + 4 + q < J ? jb + 4 + q : J - 1] << (8 * q);
                }
                w0 = v0; w1 = v1;
            }
        };
        if (jw0 < jw1) load_block(jw0);
        for (int jb = jw0; jb < jw1; jb += I1_RB) {
            uint32_t y0, y1 = 0;                                       // the block's response bytes, item jb first
            if constexpr (WORDS) {
                const uint32_t sh = (uint32_t)(jb & 3);                // wave-uniform
                y0 = __builtin_amdgcn_alignbyte(w1, w0, sh);
                if constexpr (I1_RB > 4) y1 = __builtin_amdgcn_alignbyte(w2, w1, sh);
            } else {
                y0 = w0; y1 = w1;
            }
            if (!valid) { y0 = 0xFEFEFEFEu; y1 = 0xFEFEFEFEu; }       // no person on this lane: cells outside the problem
            if (jb + I1_RB < jw1) load_block(jb + I1_RB);              // the next block's words while this one is evaluated
            const int nj = jw1 - jb < I1_RB ? jw1 - jb : I1_RB;        // wave-uniform
#pragma unroll
            for (int jj = 0; jj < I1_RB; ++jj) {
                if (jj < nj) {
                    const int j = jb + jj;
                    const float yf = (float)(((jj < 4 ? y0 : y1) >> (8 * (jj & 3))) & 0xFFu);     // v_cvt_f32_ubyteN
                    const float aD = ptab[j * NPF + 0], bD = ptab[j * NPF + 1];
                    float cj = 0.f, dj = 1.f, oj = 0.f;
                    if constexpr (MODEL >= 3) { cj = ptab[j * NPF + 2]; dj = ptab[j * NPF + 3]; oj = ptab[j * NPF + 4]; }
                    const float z = fmaf(x, aD, bD);
                    float lp, dz, dc, dd;
                    irt_cell_f<MODEL>(z, yf, cj, dj, oj, lp, dz, dc, dd);  // branch-free; y >= 254 -> no gradient
                    llp += lp;
                    gxp = fmaf(dz, aD, gxp);
                    float* const ps = scr + jj * SCR + lane * NQ;
                    if constexpr (NQ == 4) *(f32x4*)ps = f32x4{dz * x, dz, dc, dd};
                    else { typedef float f32x2_ __attribute__((ext_vector_type(2))); *(f32x2_*)ps = f32x2_{dz * x, dz}; }
                }
            }
            // transposed read: this lane's (item, quantity) over its NSRC persons, in a fixed order
            {
                float t = 0.f;
                const float* const src = scr + rjj * SCR + rq;
#pragma unroll 8
                for (int s2 = 0; s2 < NSRC; ++s2) t += src[(rg * NSRC + ((s2 + rrot) & (NSRC - 1))) * NQ];
                t += __shfl_xor(t, 16, 64);                            // groups 0 + 1, 2 + 3
                t = half_sum32(t);                                     // + the other lane half: every lane holds its pair's sum
                if (lane < PAIRS && rjj < nj) acc[(jb + rjj) * NQ + rq] += t;   // (this wave's items are nobody else's)
            }
        }
        scr[lane] = llp;                                               // (the wave's parked terms are done with: LDS runs a wave's
        scr[64 + lane] = gxp;                                          // operations in order)
        __syncthreads();
        if (wave == 0 && valid) {
            float my_ll = 0.f, my_gx = 0.f;
#pragma unroll
            for (int w = 0; w < I1_WAVES; ++w) { my_ll += scr0[w * I1_RB * SCR + lane]; my_gx += scr0[w * I1_RB * SCR + 64 + lane]; }
            const float gxt = dm.scale * (my_gx - xv);                 // d ELBO / d x (likelihood + prior); gx carries Dc already
            gloc[i] = -gxt;
            graw[i] = -(gxt * sig * e + dm.scale);                      // + scale from the entropy term
            const float el = my_ll - 0.5f * xv * xv + 0.5f * e * e + r; // log p(y|x) + log p(x) - log q(x)
            elbo[i] = el;
            el_acc += el;
        }
    }
    // one slab per block: [a: J | b: J | c: J | d: J | ELBO share]; t = Dc dz enters ga and gb
    __syncthreads();
    float* slab = slabs + (int64_t)blockIdx.x * (4 * J + 1);
    for (int e2 = tid; e2 < 4 * J; e2 += I1_THREADS) {
        const int q = e2 / J, j = e2 - q * J;
        float v = 0.f;
        if (q == 0) { if (MODEL >= 2) v = dm.Dc * acc[j * NQ + 0]; }
        else if (q == 1) v = dm.Dc * acc[j * NQ + 1];
        else if (MODEL >= 3 && q == 2) v = acc[j * NQ + 2];
        else if (MODEL >= 4 && q == 3) v = acc[j * NQ + 3];
        slab[e2] = dm.scale * v;
    }
    if (wave == 0) {                                                   // column 4 J: this block's share of the ELBO
        el_acc = wave_sum_dpp(el_acc);
        if (lane == 0) slab[4 * J] = dm.scale * el_acc;
    }
}

// ---------------------------------------------------------------------------------------------
// The ITEM-PER-LANE form (rounds 1-4), kept for J > 128 where it is the faster one: lane l keeps the parameters and gradient
// accumulators of items 4l..4l+3 (+256, ...) in registers; a wave walks its persons in groups of 64 (lane l also owns the
// per-person scalars of person 64 g + l), the response row is read as one coalesced 4-byte word per lane per 256-item slice,
// the next person's words are requested before the current one is evaluated, and the only cross-lane traffic is two DPP wave
// reductions (log-lik, d/dx) per person -- amortised over J / 64 cells a lane -- no LDS in the person loop.  Measured on one
// box (tools/irt1d_bench.hip and bench.py): dense 2PL 1M x 500 0.48 ms here against 0.74 ms for the person-per-lane form (whose
// item-gradient reduction costs ~8 instructions an ITEM and 64 persons); 4PL 100 k x 100 40 us here against 28 us there (100
// items fill 25 of 32 lane slots, the per-person reductions and broadcasts are a fifth of the loop); 4PL 100 k x 160: 38 against
// 56 us; 4PL 100 k x 256: 57 us either way; 2PL 1M x 128 / 256 / 500: 0.141 / 0.28 / 0.75 ms person-per-lane against 0.29 / 0.30 /
// 0.53 ms here.  Dispatch: J <= 256 -> k_irt1d, else this kernel.
// ---------------------------------------------------------------------------------------------
// WPL = 4-item words per lane (items 256 w + 4 lane + 0..3); 128 < J <= 1024 -> WPL = 1, 2 or 4
template <int MODEL, int WPL, bool WORDS>
__global__ __launch_bounds__(I1_THREADS) void k_irt1d_items(
    Irt1dDims dm, const uint8_t* __restrict__ y, const int64_t* __restrict__ rows, int64_t gid0,
    const float* __restrict__ loc, const float* __restrict__ raw, const float* __restrict__ eps_in,
    uint64_t seed, uint32_t step, const uint32_t* __restrict__ step_dev, uint32_t stream, const float* __restrict__ a,
    const float* __restrict__ b, const float* __restrict__ c_un, const float* __restrict__ d_un,
    float* __restrict__ gloc, float* __restrict__ graw, float* __restrict__ elbo, float* __restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [4*J] block-level item-grad reduction
    __shared__ float el_w[I1_THREADS / 64];
    float el_acc = 0.f;
    if (step_dev) step = *step_dev;                                // replayed from a HIP graph: the counter lives on the device
    // Adam's count of this step for a fused optimiser tail (k_reduce_adam reads it; nobody in its launch reads step_dev)
    if (blockIdx.x == 0 && threadIdx.x == 0) ((uint32_t*)slabs)[(size_t)gridDim.x * (4 * (size_t)dm.J + 1)] = step + 1u;
    const int ilane = threadIdx.x & 63;                            // the lane's position on the item axis
    constexpr int IPL = 4 * WPL;
    const int J = dm.J;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t n_waves = (int64_t)gridDim.x * (I1_THREADS / 64);
    const int64_t wg = (int64_t)blockIdx.x * (I1_THREADS / 64) + wave;
    const int G = dm.gsz;
    const int64_t n_groups = (dm.nb + G - 1) / G;
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
        const int64_t i = grp * G + lane;
        const bool valid = lane < G && i < dm.nb;
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
        const int cnt = (int)((dm.nb - grp * G) < G ? (dm.nb - grp * G) : G);
        uint32_t wcur[WPL], wnext[WPL];
        auto person_row = [&](int pp) -> int64_t {                  // the response row of person pp of the group
            return (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane(row_hi, pp) << 32) |
                             (uint32_t)__builtin_amdgcn_readlane(row_lo, pp));
        };
        const int n_it = cnt;
        load_words(wcur, person_row(0));
        for (int pp = 0; pp < n_it; ++pp) {
            const int pn = (pp + 1 < n_it) ? pp + 1 : pp;                   // prefetch the next person's words
            load_words(wnext, person_row(pn));
            const float x = lane_bcast(xv, pp);
            float llp = 0.f, gxp = 0.f;
#pragma unroll
            for (int q = 0; q < IPL; ++q) {
                const unsigned yy = (wcur[q >> 2] >> (8 * (q & 3))) & 0xFFu;
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
            llp = wave_sum_dpp(llp);
            gxp = wave_sum_dpp(gxp);
            if (lane == pp) { my_ll = llp; my_gx = gxp; }
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
        if (j < J) {
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

