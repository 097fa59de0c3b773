// The default form (VX_F16X2=0 selects k_mvn_enc_bwd_w_t) of the head weight gradients
//     gWp[r][hh] = sum_p V[r][p] h[p][hh],   V = (G or GD row) * (E row or ones),
// on the fp16 MFMA (v_mfma_f32_32x32x16_f16, fp32 accumulate) with two-term operands ("f16x2", vx_common.h): THREE cross
// products (lo hi, hi lo, hi hi) at the accuracy of the fp32 chain.
//   h  : two fp16 arrays hs[2][64][nb] (h 2^sh, sc[3] of k_enc_scales) written by the forward kernel (k_split2_f16
//        otherwise), staged by DMA, 16-byte fragments;
//   V  : fp32 product (8 persons per lane half and chunk), scaled by ONE power of two for the whole launch -- from the
//        largest |gx|, |eps|, |gd| of the step, collected by the hidden-gradient kernel that runs before this one
//        (k_mvn_enc_bwd_h_b; k_absmax3 otherwise) -- then split in registers.
// Fragment layout of the 32x32x16 MFMA: lane (r = lane & 31, h = lane >> 5) holds A[row r][k = 8h + j] and
// B[k = 8h + j][col r], j = 0..7; here k = person within a 16-person chunk.
// (included by vx_abi.hip after k_mvn_bwd_t.hip, whose row layout and helpers it shares)

typedef uint16_t u16x8 __attribute__((ext_vector_type(8)));

// row map (128-byte row units): G [0, DR) | E [DR, 2 DR) | H area: 2 terms x 64 rows x 64 bytes = 64 units |
// GD [2 DR + 64, 3 DR + 64) | C (ones, zeros) [3 DR + 64, 3 DR + 80)   -- the map of k_mvn_bwd_t.hip
__host__ __device__ inline int bb_row_H(int D) { return 2 * bt_dr(D); }
__host__ __device__ inline int bb_row_GD(int D) { return 2 * bt_dr(D) + 64; }
__host__ __device__ inline int bb_row_ones(int D) { return 3 * bt_dr(D) + 64; }
__host__ __device__ inline int bb_rows(int D) { return 3 * bt_dr(D) + 80; }
#define BB_BUF 59392
__host__ __device__ inline size_t bb_lds_bytes(int D) { return (size_t)BB_BUF + (size_t)bb_rows(D) * 128; }
// byte offset inside the H area of hidden row hh of term s3, 16-byte slot s (persons 8 s .. 8 s + 7 of the tile):
// four rows share a 256-byte bank row; the slot is XORed with (hh >> 2) & 3 so that the 16 lanes of a ds_read_b128
// group (hh = 0-3, 12-15, 20-27 (+32)) land on 16 different slots
__host__ __device__ inline uint32_t bb_haddr(int s3, int hh, int s) {
    return (uint32_t)(s3 * 4096 + (hh >> 2) * 256 + ((((hh & 3) << 2) | (s ^ ((hh >> 2) & 3))) << 4));
}

// hs[s][i] (s = 0, 1) = the two fp16 terms of v[i] * scale[0] (round to nearest at both stages)
__global__ void k_split2_f16(const float* __restrict__ v, int64_t n, const float* __restrict__ scale, uint16_t* __restrict__ hs) {
    const float sc0 = scale[0];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        split2h_bits(v[i] * sc0, hs[i], hs[n + i]);
}

// out[w] = float bits of max |v_w|, w = 0, 1, 2 (the step's operand maxima when the hidden-gradient kernel that normally
// collects them is not the one that ran); out must have been cleared
__global__ void k_absmax3(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c, int64_t n,
                          uint32_t* __restrict__ out) {
    float m[3] = {0.f, 0.f, 0.f};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        m[0] = fmaxf(m[0], fabsf(a[i])); m[1] = fmaxf(m[1], fabsf(b[i])); m[2] = fmaxf(m[2], fabsf(c[i]));
    }
#pragma unroll
    for (int w = 0; w < 3; ++w) {
        const float r = wave_max_dpp(m[w]);
        if ((threadIdx.x & 63) == 0) atomic_max_raise(out + w, r);        // (conditional: atomics on one word serialise in the L2)
    }
}
__global__ void k_clear_words(uint32_t* __restrict__ w, int n) { if ((int)threadIdx.x < n) w[threadIdx.x] = 0u; }


// BWB_WAVES waves of BWB_RT row tiles each (BWB_WAVES * BWB_RT * 32 = BT_ROWS rows a workgroup, as k_mvn_bwd_t.hip).  Round 3:
// eight waves of two row tiles instead of four of four -- 64 accumulator registers a wave, the kernel fits 256 registers
// (the MFMAs take the VGPR form) and two waves share a SIMD: one wave's products / split / LDS waits run under the other's
// MFMAs: 3.12 -> 2.77 ms; then SIXTEEN waves of one row tile (128 registers, four waves a SIMD): 2.82 -> 2.72 ms on one box
// (tools/bwb_bench.hip).  (BWB_WAVES 4 / BWB_RT 4 is the round-2 form.)
#ifndef BWB_WAVES
#define BWB_WAVES 16
#endif
#ifndef BWB_PRESCALE
#define BWB_PRESCALE 0                                                  // 1: the E rows take V's power of two in LDS instead of a multiply per
                                                                       // product -- 8 vector instructions a group less, and no faster
                                                                       // (2.88-2.91 against 2.83-2.84 ms on one box, tools/bwb_bench.hip)
#endif
#ifndef BWB_MIX
#define BWB_MIX 1                                                       // 1: the two fp16 terms of a product straight from v_fma_mix*_f16 (below)
#endif
#define BWB_RT (16 / BWB_WAVES)
#define BWB_THREADS (64 * BWB_WAVES)
__global__ __launch_bounds__(BWB_THREADS, 1) void k_mvn_enc_bwd_w_b(
    EncDims dm, const uint16_t* __restrict__ hs /*[2][64][nb] fp16 terms of h 2^sh*/, const float* __restrict__ epsT,
    const float* __restrict__ gdT, const float* __restrict__ gxT, const uint32_t* __restrict__ gtab,
    const float* __restrict__ sc /*k_enc_scales*/, const uint32_t* __restrict__ maxw /*float bits: max |gx|, |gd|, |eps|*/,
    float* __restrict__ slabs, int64_t slab_len) {
    extern __shared__ __attribute__((aligned(16))) char smem_bb[];
    const int D = dm.D;
    const int64_t nb = dm.nb;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    constexpr uint32_t BUF = BB_BUF;
    const int Rp = pk_rows(D);
    int rb_, blk_pr;
    bt_decode(rb_, blk_pr);
    const int64_t rbase = (int64_t)rb_ * BT_ROWS + (int64_t)wave * BWB_RT * 32;
    const int rE = bt_row_E(D), rH = bb_row_H(D), rGD = bb_row_GD(D), rOnes = bb_row_ones(D), rZero = rOnes + 1;
    const bool need_gd = (int64_t)(rb_ + 1) * BT_ROWS > pk_off_total(D);
    // one power of two for every V of the launch: |V| <= max(|gx|max |eps|max, |gd|max, |gx|max)
    float v_scale, out_inv;
    {
        const float gm = __builtin_bit_cast(float, maxw[0]), dmx = __builtin_bit_cast(float, maxw[1]), em = __builtin_bit_cast(float, maxw[2]);
        const int sv = f16_scale_exp(fmaxf(fmaxf(gm * em, dmx), gm));
        v_scale = ldexpf(1.0f, sv);
        out_inv = ldexpf(1.0f, -sv) / sc[3];                              // takes 2^(sv + sh) off the accumulators
    }

    __shared__ uint32_t row_used[16];                                  // bit r: row unit r of the tile map is read by some row of the slab
    if (tid < 16) row_used[tid] = 0u;
    __syncthreads();
    // ---- per-lane LDS addresses: fp32 rows, persons 16 c + 8 half + 0..7 = chunks 4c + 2 half and + 1
    uint32_t aG[BWB_RT][2], aE[BWB_RT][2], aHf[2][2][2];                 // [..][chunk c]; H: [term][hidden tile][chunk]
#pragma unroll
    for (int t = 0; t < BWB_RT; ++t) {
        const int64_t pr = rbase + 32 * t + l31;
        int g = rZero, e = rOnes;
        if (pr < Rp) {
            const uint32_t code = gtab[pr >> 3];
            const uint32_t type = code >> 28, k = (code >> 12) & 0xFFFFu, l0 = code & 0xFFFu, jx = (uint32_t)(pr & 7);
            if (type == PK_OFF) { if (l0 + jx < k) { g = (int)k; e = rE + (int)(l0 + jx); } }
            else if (type == PK_LOC) { if (k + jx < (uint32_t)D) g = (int)(k + jx); }
            else if (type == PK_DIAG) { if (k + jx < (uint32_t)D) g = rGD + (int)(k + jx); }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) { aG[t][c] = bt_addr(g, 4 * c + 2 * half); aE[t][c] = bt_addr(e, 4 * c + 2 * half); }
        // which row units of the tile this workgroup's 512 rows read at all (see "only what the slab reads" below)
        if (half == 0) { atomicOr(&row_used[g >> 5], 1u << (g & 31)); atomicOr(&row_used[e >> 5], 1u << (e & 31)); }
    }
    // second 16-byte piece of a row's pair: chunk index + 1 = slot XOR 1 (4c + 2 half is even)
#pragma unroll
    for (int s3 = 0; s3 < 2; ++s3)
#pragma unroll
        for (int ht = 0; ht < 2; ++ht)
#pragma unroll
            for (int c = 0; c < 2; ++c) aHf[s3][ht][c] = (uint32_t)rH * 128u + bb_haddr(s3, 32 * ht + l31, 2 * c + half);

    for (int b = 0; b < 2; ++b) {
        // the "ones" row holds the power of two of V (the E rows are multiplied by it as they land: scale_own below)
        if (tid < 64) *(float*)(smem_bb + b * BUF + bt_addr(rOnes + (tid >> 5), (tid & 31) >> 2) + 4 * (tid & 3)) = (tid < 32) ? (BWB_PRESCALE ? v_scale : 1.0f) : 0.f;
    }
    f32x16 acc[BWB_RT][2];
    float bsum[BWB_RT];
#pragma unroll
    for (int t = 0; t < BWB_RT; ++t) { bsum[t] = 0.f; acc[t][0] = zero16(); acc[t][1] = zero16(); }

    const int64_t n_ptiles = (nb + BT_P - 1) / BT_P;
    // DMA transfers of 1 KB (see k_mvn_bwd_t.hip): d < rH / 8: G / E rows; the next 8: the H area (4 per term);
    // then GD rows.  Per-lane global addresses are fixed; only the person offset of the tile is added.
    constexpr int BB_MAXD = (56 + BWB_WAVES - 1) / BWB_WAVES;
    const int dH0 = rH / 8, dGD0 = rGD / 8;
    const int n_dma = need_gd ? rOnes / 8 : dGD0;
    // per-lane 64-bit source address of transfer u at person 0, and the shift of a person index to bytes (bf16 planes 1,
    // fp32 rows 2): a tile adds (i0 << shift) with ONE vector instruction per transfer.  (Choosing the region's base on
    // the scalar unit per transfer kept 30 loop-invariant SGPRs alive; they spilled to VGPR lanes: 70 v_readlane a tile.)
    const char* vbase[BB_MAXD];
    uint32_t vsh[BB_MAXD];
    uint32_t voff[BB_MAXD];
    // Only what the slab reads is staged.  A slab of 512 packed rows reads the G rows of its own k (a dozen of the D), the E
    // rows below its largest k, h, and GD only in the slab with the diagonal rows: 14-23 KB a tile of 32 persons instead of the
    // whole map's 34 KB for every slab (17.7 GB a launch through the L2 -> LDS path before).  A transfer (8 row units: 4 and
    // the 4 eight further on) is issued if one of its units is in row_used.  The kernel's time did not change (it does not
    // wait for these bytes, docs/NOTEBOOK.md); the L2 it shares with the kernel that runs beside it carries 40 % less.
    __syncthreads();                                                   // row_used is complete
    uint32_t need_u = 0u;                                              // bit u: transfer wave + BWB_WAVES u is issued
#pragma unroll
    for (int u = 0; u < BB_MAXD; ++u) {
        const int d = wave + BWB_WAVES * u;
        if (d >= dH0 && d < dGD0) {
            need_u |= 1u << u;
        } else if (d < n_dma) {
            const int r0 = 16 * (d >> 1) + 4 * (d & 1);                // units r0 .. r0 + 3 and r0 + 8 .. r0 + 11
            const uint64_t w2 = ((uint64_t)row_used[(r0 >> 5) + 1] << 32) | row_used[r0 >> 5];
            if ((w2 >> (r0 & 31)) & 0x0F0Full) need_u |= 1u << u;
        }
    }
    need_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)need_u);
#pragma unroll
    for (int u = 0; u < BB_MAXD; ++u) {
        const int d = wave + BWB_WAVES * u;
        if (d >= dH0 && d < dGD0) {                                    // H area: 16 hidden rows of one term
            const int dd = d - dH0, s3 = dd >> 2;
            const int hh = 16 * (dd & 3) + 4 * (lane >> 4) + ((lane & 15) >> 2);
            const int s = (lane & 3) ^ ((hh >> 2) & 3);
            voff[u] = (uint32_t)((((int64_t)s3 * 64 + hh) * nb + 8 * s) * 2);
        } else {
            const int i = 4 * (d & 1) + (lane >> 4), beta = (lane >> 3) & 1;
            const int R = 16 * (d >> 1) + 8 * beta + i;
            const int c = (lane & 7) ^ i;
            int rl = R >= rGD ? R - rGD : R >= rE ? R - rE : R;
            if (rl >= D) rl = D - 1;                                   // padding rows of a region: a harmless duplicate
            voff[u] = (uint32_t)(((int64_t)rl * nb + 4 * c) * 4);
        }
        const bool isH = d >= dH0 && d < dGD0;
        const void* rbase = d >= dGD0 ? (const void*)gdT : isH ? (const void*)hs : d >= rE / 8 ? (const void*)epsT : (const void*)gxT;
        vbase[u] = (const char*)rbase + voff[u];
        vsh[u] = isH ? 1u : 2u;
    }
    auto stage = [&](int64_t tile, int b) __attribute__((always_inline)) {
        const int64_t i0 = tile * BT_P;
        const int pv = (int)((nb - i0) < BT_P ? (nb - i0) : BT_P);
        const uint32_t lbase = lds_addr_uniform(smem_bb + b * BUF) + (uint32_t)wave * 1024u;
        if (pv < BT_P) {                                               // the last tile: absent persons are zeros
            for (int e = tid; e < bb_rows(D) * 32; e += BWB_THREADS) {
                const int row = e >> 5;
                if (row != rOnes) ((float*)(smem_bb + b * BUF))[e] = 0.f;
            }
            __syncthreads();
        }
#pragma unroll
        for (int u = 0; u < BB_MAXD; ++u) {
            const int d = wave + BWB_WAVES * u;
            if (d < n_dma && ((need_u >> u) & 1u)) {
                const bool isH = d >= dH0 && d < dGD0;
                const char* src = vbase[u] + ((uint64_t)i0 << vsh[u]);
                if (pv == BT_P) {
                    dma16(src, lbase + (uint32_t)u * (1024u * BWB_WAVES));
                } else {                                               // persons of this lane's 16 bytes: 8 (H) or 4 (fp32)
                    int p0;
                    if (isH) {
                        const int hh = 16 * ((d - dH0) & 3) + 4 * (lane >> 4) + ((lane & 15) >> 2);
                        p0 = 8 * ((lane & 3) ^ ((hh >> 2) & 3));
                    } else {
                        p0 = 4 * ((lane & 7) ^ (4 * (d & 1) + (lane >> 4)));
                    }
                    if (p0 < pv) dma16(src, lbase + (uint32_t)u * (1024u * BWB_WAVES));
                }
            }
        }
    };

    // V = G * E is formed in fp16 range: the E rows (and the "ones" row) carry the launch's power of two.  Each wave multiplies
    // the E transfers it moved itself, in place, once they have landed and before the barrier that publishes the tile (8
    // multiplies a group less in the loop; the bias sums carry the power of two as well and lose it at the end).
    const int dE0 = rE / 8;
    auto scale_own = [&](int b) __attribute__((always_inline)) {
        char* lb = smem_bb + b * BUF + wave * 1024 + lane * 16;
#pragma unroll
        for (int u = 0; u < BB_MAXD; ++u) {
            const int d = wave + BWB_WAVES * u;
            if (d >= dE0 && d < dH0) {
                f32x4 v = *(f32x4*)(lb + u * (1024 * BWB_WAVES));
                v[0] *= v_scale; v[1] *= v_scale; v[2] *= v_scale; v[3] *= v_scale;
                *(f32x4*)(lb + u * (1024 * BWB_WAVES)) = v;
            }
        }
    };

    // ---- compute: 2 chunks x BWB_RT row tiles = 2 BWB_RT groups per tile and wave, 6 MFMAs per group.  The fragments of group g + 1
    // (4 LDS reads, 8 products, their scaling, 4 x (split of an element pair), the bias sum) are made in the shadow of the
    // MFMAs of group g, a few vector instructions after each MFMA; every slice is a pinned scheduling region.  The barrier
    // of a tile sits before its LAST group, whose operands are in registers already: behind it the buffer is free for the
    // DMA of the tile after next, and the prefetch of that group reads the next tile from the other buffer.
    // Vector instructions in the shadow of an MFMA (tools/slice_ubench.hip): about five single-pass ones are free;
    // v_pk_*_f32 are NOT (they wait for the matrix pipe).  V 2^sv is split into two fp16 terms, six instructions per
    // element pair: heads = v_cvt_pk_f16_f32 (round to nearest, both elements at once), the two halves back as floats
    // (v_cvt_f32_f16, the upper one through SDWA), the two exact remainders, and the remainders packed by the same
    // conversion.
    // Every instruction of a slice is a volatile asm statement: the optimizer otherwise re-vectorizes the scalar
    // arithmetic into v_pk_*_f32 across slices and sinks whole slices out of the MFMA shadow.  A slice alternates pieces
    // of two element pairs: a vector instruction that depends on the one issued just before it waits for it (measured
    // 1.66 x the issue time).
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    uint32_t fq[2][2][4];                                              // [parity of the group][hi, lo][element pair]
    f16x8 hf[2][2][2];                                                 // [parity of the chunk][term][hidden tile]
    f32x4 rg0, rg1, re0, re1;
    float pv_[8], ps_[8], pr_[8], s0, s1, s2, s3;
    auto amul = [](float x, float y) -> float { float d; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y)); return d; };
    auto aadd = [](float x, float y) -> float { float d; asm volatile("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y)); return d; };
    auto read_raw = [&](const char* rb, uint32_t ag, uint32_t ae) __attribute__((always_inline)) {
        rg0 = *(const f32x4*)(rb + ag); rg1 = *(const f32x4*)(rb + (ag ^ 16u));
        re0 = *(const f32x4*)(rb + ae); re1 = *(const f32x4*)(rb + (ae ^ 16u));
    };
    auto products0 = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { pv_[j] = amul(rg0[j], re0[j]); if (!BWB_MIX) ps_[j] = BWB_PRESCALE ? pv_[j] : amul(pv_[j], v_scale); }
        s0 = aadd(pv_[0], pv_[1]); s1 = aadd(pv_[2], pv_[3]);
    };
    auto products1 = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { pv_[4 + j] = amul(rg1[j], re1[j]); if (!BWB_MIX) ps_[4 + j] = BWB_PRESCALE ? pv_[4 + j] : amul(pv_[4 + j], v_scale); }
        s2 = aadd(pv_[4], pv_[5]); s3 = aadd(pv_[6], pv_[7]);
    };
    auto asub = [](float x, float y) -> float { float d; asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y)); return d; };
    float th[2][2];                                                    // [pair parity][element]: the heads as floats
    auto acvt = [](float x0, float x1) -> uint32_t {
        uint32_t d; asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(d) : "v"(x0), "v"(x1)); return d; };
    auto alo16 = [](uint32_t x) -> float { float d; asm volatile("v_cvt_f32_f16_e32 %0, %1" : "=v"(d) : "v"(x)); return d; };
    auto ahi16 = [](uint32_t x) -> float {
        float d; asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(d) : "v"(x)); return d; };
    auto C1 = [&](auto pc, auto nc) __attribute__((always_inline)) {
        constexpr int p = decltype(pc)::value, nx = decltype(nc)::value;
        fq[nx][0][p] = acvt(ps_[2 * p], ps_[2 * p + 1]);
    };
    auto C2 = [&](auto pc, auto nc) __attribute__((always_inline)) {
        constexpr int p = decltype(pc)::value, nx = decltype(nc)::value;
        th[p & 1][0] = alo16(fq[nx][0][p]);
        th[p & 1][1] = ahi16(fq[nx][0][p]);
    };
    auto C3 = [&](auto pc) __attribute__((always_inline)) {
        constexpr int p = decltype(pc)::value;
        pr_[2 * p] = asub(ps_[2 * p], th[p & 1][0]);
        pr_[2 * p + 1] = asub(ps_[2 * p + 1], th[p & 1][1]);
    };
    auto C4 = [&](auto pc, auto nc) __attribute__((always_inline)) {
        constexpr int p = decltype(pc)::value, nx = decltype(nc)::value;
        fq[nx][1][p] = acvt(pr_[2 * p], pr_[2 * p + 1]);
    };
    // BWB_MIX: the split of an element pair in FOUR instructions instead of eight (two scalings, head conversion, two
    // conversions back, two subtractions, remainder conversion).  v_fma_mixlo_f16 / v_fma_mixhi_f16 compute an fp32 fma of
    // operands that are fp32 or one half of a register read as fp16, and write the result, rounded to fp16, into one half
    // of the destination: head = rn16(v * 2^sv + 0), remainder = rn16(v * 2^sv - head).  Both fmas are exact in fp32 (a
    // power-of-two scaling; the difference of a value and its own fp16 rounding), so each result is rounded once, to fp16,
    // exactly as in the longer sequence: the same bits, except that a product of -0 splits into (+0, -0) instead of (-0, +0)
    // (tools/bwb_bench.hip: the checksum of every slab word is the same).  16 vector instructions a group less -- and the
    // same 2.65-2.70 ms: this kernel does not wait for its vector instructions (docs/NOTEBOOK.md, round 4).
    // (the asm statements sit in plain lambdas: operands that are captures of a GENERIC lambda do not compile)
    const float vs_s = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v_scale)));
    auto mix_h_lo = [](uint32_t& d, float v, float sc_) { asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(d) : "v"(v), "s"(sc_)); };
    auto mix_h_hi = [](uint32_t& d, float v, float sc_) { asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(d) : "v"(v), "s"(sc_)); };
    auto mix_l_lo = [](uint32_t& d, float v, float sc_, uint32_t h) {
        asm volatile("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(d) : "v"(v), "s"(sc_), "v"(h)); };
    auto mix_l_hi = [](uint32_t& d, float v, float sc_, uint32_t h) {
        asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(d) : "v"(v), "s"(sc_), "v"(h)); };
    auto H1 = [&](auto pc, auto nc) __attribute__((always_inline)) {     // head of element 2 p -> low half
        constexpr int p = decltype(pc)::value, nx = decltype(nc)::value;
        mix_h_lo(fq[nx][0][p], pv_[2 * p], vs_s);
    };
    auto H2 = [&](auto pc, auto nc) __attribute__((always_inline)) {     // head of element 2 p + 1 -> high half
        constexpr int p = decltype(pc)::value, nx = decltype(nc)::value;
        mix_h_hi(fq[nx][0][p], pv_[2 * p + 1], vs_s);
    };
    auto L1 = [&](auto pc, auto nc) __attribute__((always_inline)) {     // remainder of element 2 p
        constexpr int p = decltype(pc)::value, nx = decltype(nc)::value;
        mix_l_lo(fq[nx][1][p], pv_[2 * p], vs_s, fq[nx][0][p]);
    };
    auto L2 = [&](auto pc, auto nc) __attribute__((always_inline)) {     // remainder of element 2 p + 1
        constexpr int p = decltype(pc)::value, nx = decltype(nc)::value;
        mix_l_hi(fq[nx][1][p], pv_[2 * p + 1], vs_s, fq[nx][0][p]);
    };
    auto frag = [&](auto cc, auto kc) -> f16x8 {
        constexpr int cu = decltype(cc)::value, k = decltype(kc)::value;
        return __builtin_bit_cast(f16x8, u32x4{fq[cu][k][0], fq[cu][k][1], fq[cu][k][2], fq[cu][k][3]});
    };
    auto tile_body = [&](auto bc, bool has_next, int64_t stage_tile) {
        constexpr int b = decltype(bc)::value;
        static_for<2 * BWB_RT>([&](auto gc) {
            constexpr int gi = decltype(gc)::value, c = gi / BWB_RT, t = gi % BWB_RT, cur = gi & 1, nxt = cur ^ 1;
            constexpr int gn = (gi + 1) % (2 * BWB_RT), cn = gn / BWB_RT, tn = gn % BWB_RT;
            constexpr bool last = gi == 2 * BWB_RT - 1;
            if constexpr (last) {
                vx_wait_vmem();
                if (BWB_PRESCALE) scale_own(1 - b);                     // this wave's E rows of the next tile, in place
                __syncthreads();                                       // next tile landed; this tile's buffer is free
                if (stage_tile >= 0) stage(stage_tile, b);
            }
            const char* rb = smem_bb + (last ? 1 - b : b) * BUF;
            constexpr std::integral_constant<int, cur> curc{};
            constexpr std::integral_constant<int, nxt> nxtc{};
            constexpr std::integral_constant<int, 0> I0{};
            constexpr std::integral_constant<int, 1> I1{};
            constexpr std::integral_constant<int, 2> I2{};
            constexpr std::integral_constant<int, 3> I3{};
            const f16x8 vh = frag(curc, I0), vl = frag(curc, I1);
            __builtin_amdgcn_sched_barrier(0);
            acc[t][0] = mfma_f16(vl, hf[c][0][0], acc[t][0]);
            __builtin_amdgcn_sched_barrier(0);
            read_raw(rb, aG[tn][cn], aE[tn][cn]);
            if constexpr (t == BWB_RT - 1) {
#pragma unroll
                for (int s3 = 0; s3 < 2; ++s3)
#pragma unroll
                    for (int ht = 0; ht < 2; ++ht) hf[cn][s3][ht] = *(const f16x8*)(rb + aHf[s3][ht][cn]);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[t][1] = mfma_f16(vl, hf[c][0][1], acc[t][1]);
            __builtin_amdgcn_sched_barrier(0);
            products0();
            __builtin_amdgcn_sched_barrier(0);
            acc[t][0] = mfma_f16(vh, hf[c][1][0], acc[t][0]);
            __builtin_amdgcn_sched_barrier(0);
            products1();
            __builtin_amdgcn_sched_barrier(0);
            acc[t][1] = mfma_f16(vh, hf[c][1][1], acc[t][1]);
            __builtin_amdgcn_sched_barrier(0);
#if BWB_MIX
            H1(I0, nxtc); H1(I1, nxtc); s0 = aadd(s0, s1); H2(I0, nxtc); H2(I1, nxtc); H1(I2, nxtc); H1(I3, nxtc);
            s2 = aadd(s2, s3); H2(I2, nxtc); H2(I3, nxtc); s0 = aadd(s0, s2);
            __builtin_amdgcn_sched_barrier(0);
            acc[t][0] = mfma_f16(vh, hf[c][0][0], acc[t][0]);
            __builtin_amdgcn_sched_barrier(0);
            L1(I0, nxtc); L1(I1, nxtc); L2(I0, nxtc); L2(I1, nxtc);
            if constexpr (last) bsum[tn] = aadd(bsum[tn], has_next ? s0 : 0.f); else bsum[tn] = aadd(bsum[tn], s0);
            L1(I2, nxtc); L1(I3, nxtc);
            __builtin_amdgcn_sched_barrier(0);
            acc[t][1] = mfma_f16(vh, hf[c][0][1], acc[t][1]);
            __builtin_amdgcn_sched_barrier(0);
            L2(I2, nxtc); L2(I3, nxtc);
            __builtin_amdgcn_sched_barrier(0);
#else
            C1(I0, nxtc); C1(I1, nxtc); s0 = aadd(s0, s1); C2(I0, nxtc); C2(I1, nxtc); C3(I0); C1(I2, nxtc);
            C3(I1); C4(I0, nxtc); s2 = aadd(s2, s3); C1(I3, nxtc); s0 = aadd(s0, s2);
            __builtin_amdgcn_sched_barrier(0);
            acc[t][0] = mfma_f16(vh, hf[c][0][0], acc[t][0]);
            __builtin_amdgcn_sched_barrier(0);
            C2(I2, nxtc); C4(I1, nxtc); C2(I3, nxtc);
            if constexpr (last) bsum[tn] = aadd(bsum[tn], has_next ? s0 : 0.f); else bsum[tn] = aadd(bsum[tn], s0);
            C3(I2); C3(I3);
            __builtin_amdgcn_sched_barrier(0);
            acc[t][1] = mfma_f16(vh, hf[c][0][1], acc[t][1]);
            __builtin_amdgcn_sched_barrier(0);
            C4(I2, nxtc); C4(I3, nxtc);
            __builtin_amdgcn_sched_barrier(0);
#endif
        });
    };

    const int64_t t0 = blk_pr, GS = gridDim.y;
    if (t0 < n_ptiles) {
        stage(t0, 0);
        vx_wait_vmem();
        if (BWB_PRESCALE) scale_own(0);
        if (t0 + GS < n_ptiles) stage(t0 + GS, 1);
        __syncthreads();
        // fragments of the first group, outside the pipeline
        read_raw(smem_bb, aG[0][0], aE[0][0]);
#pragma unroll
        for (int s3 = 0; s3 < 2; ++s3)
#pragma unroll
            for (int ht = 0; ht < 2; ++ht) hf[0][s3][ht] = *(const f16x8*)(smem_bb + aHf[s3][ht][0]);
        products0();
        products1();
        {
            constexpr std::integral_constant<int, 0> z{};
            constexpr std::integral_constant<int, 1> o1{};
            constexpr std::integral_constant<int, 2> o2{};
            constexpr std::integral_constant<int, 3> o3{};
#if BWB_MIX
            H1(z, z); H2(z, z); L1(z, z); L2(z, z);
            H1(o1, z); H2(o1, z); L1(o1, z); L2(o1, z);
            H1(o2, z); H2(o2, z); L1(o2, z); L2(o2, z);
            H1(o3, z); H2(o3, z); L1(o3, z); L2(o3, z);
#else
            C1(z, z); C2(z, z); C3(z); C4(z, z);
            C1(o1, z); C2(o1, z); C3(o1); C4(o1, z);
            C1(o2, z); C2(o2, z); C3(o2); C4(o2, z);
            C1(o3, z); C2(o3, z); C3(o3); C4(o3, z);
#endif
        }
        bsum[0] += (s0 + s1) + (s2 + s3);
        int64_t tile = t0;
        while (tile < n_ptiles) {
            tile_body(std::integral_constant<int, 0>{}, tile + GS < n_ptiles, tile + 2 * GS < n_ptiles ? tile + 2 * GS : (int64_t)-1);
            tile += GS;
            if (tile < n_ptiles) {
                tile_body(std::integral_constant<int, 1>{}, tile + GS < n_ptiles, tile + 2 * GS < n_ptiles ? tile + 2 * GS : (int64_t)-1);
                tile += GS;
            }
        }
    }

    float* slab = slabs + (int64_t)blk_pr * slab_len;
#pragma unroll
    for (int t = 0; t < BWB_RT; ++t) {
#pragma unroll
        for (int ht = 0; ht < 2; ++ht) {
            const int hh = 32 * ht + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = rbase + 32 * t + crow32(r, half);
                if (row < Rp) slab[row * 64 + hh] = acc[t][ht][r] * out_inv;
            }
        }
        const float bt = half_sum32(bsum[t]) * (BWB_PRESCALE ? out_inv * sc[3] : 1.0f);   // 2^-sv: the bias sums carry V's power of two
        const int64_t row = rbase + 32 * t + l31;
        if (half == 0 && row < Rp) slab[(int64_t)Rp * 64 + row] = bt;
    }
}
