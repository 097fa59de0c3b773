// Hidden-layer gradient of the amortized MVN guide on the fp16 MFMA (two-term operand splitting "f16x2", vx_common.h; fp32
// accumulate): the same result as k_mvn_enc_bwd_h_t (k_mvn_bwd_t.hip),
//     gh[p][hh] = sum_r Wp[r][hh] V[r][p],    ghpre = gh * (1 - exp(-h))                          (autograd of vi.py:448-455)
// but V is never formed.  For the off-diagonal rows V[(k,l)][p] = gx[p][k] eps[p][l] is a rank-one product, so
//     gh[p][:] = sum_k gx[p][k] U_k[p][:],      U_k[p][hh] = sum_{l<k} W22[(k,l)][hh] eps[p][l]
// and U_k is a GEMM whose per-person operand is eps alone: scaled and split ONCE per 32-person wave tile into fp16
// fragments that stay in registers for every k.  The weights are scaled and split once per step into per-unit images
// (k_pack_heads_hb).  The multiplication by gx[p][k] -- with the powers of two of both operands folded into it -- is a
// 32-FMA epilogue per k on the accumulator.  DIAG rows (operand gd = gx eps e^M + scale, k_mvn_gd) and LOC rows (operand gx)
// are two more small GEMMs, each with the power of two of its own operand.
//   The per-person operands take their power of two from the largest magnitude among the wave's 32 persons (wave maximum):
//   no bound is assumed.
//   MFMA 32x32x16: C rows = hidden units (two tiles of 32), columns = persons; contraction index = l (or k).
//   unit (k, s) = the 16 contraction indices l = 16 s .. 16 s + 15 of one k: 4 fragments of 1 KB (2 hidden tiles x 2
//   terms), 6 MFMAs (three products a tile).  Units stream through a ring of 8 slots in LDS by DMA, shared by the 8 waves
//   of the workgroup (pairs of units = 8 transfers = 2 for each of the waves 0..3, so `vmcnt(2)` counts whole pairs); one
//   barrier per pair.  Two waves per SIMD (the kernel fits 256 registers): what one wave cannot overlap -- the epilogue
//   per k, LDS waits, the barrier -- runs under the other wave's MFMAs.
//   k runs in blocks of 16 so that the number of units per k -- and with it every fragment register -- is static.
// (included by vx_abi.hip after k_mvn_fwd_b.hip)

#define HB_THREADS 512
#define HB_WAVES 8                                                     // two per SIMD: one wave's epilogue / waits under the other's MFMAs
#define HB_UNIT_BYTES 4096
#define HB_NSLOT 8
#define HB_NS 8                                                        // k-steps of 16 covering D <= 128

// OFF units: row k = 1 .. D - 1 has ceil(k / 16) of them.  Closed forms, not loops over k: rows 16 q + 1 .. 16 q + 16 have
// q + 1 units each, so 8 q (q + 1) units lie before row 16 q + 1 (tools/pack_bench.hip: the two 99-trip loops of the pack
// body -- this count, then the search for the unit's row -- were 5 of the 9 us of k_pack_heads_hb).
__host__ __device__ inline int hb_units_off(int D) {
    if (D < 2) return 0;
    const int m = D - 1, q = m >> 4, r = m & 15;
    return 8 * q * (q + 1) + r * (q + 1);
}
// OFF unit u -> its row k and k-step s
__host__ __device__ inline void hb_unit_row(int u, int& k, int& s) {
    int q = (int)((sqrtf(1.0f + 0.5f * (float)u) - 1.0f) * 0.5f);
    while (8 * (q + 1) * (q + 2) <= u) ++q;
    while (q > 0 && 8 * q * (q + 1) > u) --q;
    const int rem = u - 8 * q * (q + 1), r = rem / (q + 1);
    s = rem - r * (q + 1);
    k = 16 * q + 1 + r;
}
__host__ __device__ inline int hb_units(int D) { return hb_units_off(D) + 2 * ((D + 15) / 16); }
__host__ __device__ inline int64_t hb_img_floats(int D) { return (int64_t)hb_units(D) * (HB_UNIT_BYTES / 4); }
__host__ __device__ inline size_t hb_lds_bytes(int D) {
    const size_t a = (size_t)HB_WAVES * D * 32 * sizeof(float) + (size_t)HB_NSLOT * HB_UNIT_BYTES;
    return a > 65536 ? a : 65536;                                      // SPLIT: [8 waves][32 registers][64 lanes] floats
}

// unit image: fragment (hidden tile ht, term sp) at byte (ht * 2 + sp) * 1024 + lane * 16, lane = 32 half + row;
// element j = the weight (times 2^sw, sc[2] of k_enc_scales) of hidden unit 32 ht + row for contraction index
// c = 16 s + 8 half + j:
//   OFF unit (k, s): W22[(k, c)] for c < k;   LOC unit s: W21[c];   DIAG unit s: W22[(c, c)]       (zero past the end)
// in that order (the 64-person kernel replaces its gx tile by the gd tile while the LOC units run).
// Block 0 also clears the words that collect the largest |gx|, |gd|, |eps| and |ghpre| of the step (the kernel below adds
// its waves' maxima; k_mvn_enc_bwd_w_b and k_fc1_bwd_b scale by them).
__device__ __forceinline__ void pack_heads_hb_unit(int u, int D, const float* __restrict__ W21, const float* __restrict__ W22,
                                                   float w_scale, uint8_t* __restrict__ img) {
    const int n_off = hb_units_off(D), ns = (D + 15) / 16;
    if (u >= n_off + 2 * ns) return;
    int type = 0, k = 0, s = 0;                                        // 0 OFF, 1 DIAG, 2 LOC
    if (u < n_off) {
        hb_unit_row(u, k, s);
    } else {
        type = (u - n_off) < ns ? 2 : 1;                               // the LOC units first, the DIAG units last
        s = (u - n_off) % ns;
    }
    uint8_t* out = img + (int64_t)u * HB_UNIT_BYTES;
    if (blockDim.x != 256) {                                           // (any other block size: element by element)
        for (int e = threadIdx.x; e < 2 * 64 * 8; e += blockDim.x) {
            const int j = e & 7, lane = (e >> 3) & 63, ht = e >> 9;
            const int half = lane >> 5, hh = 32 * ht + (lane & 31);
            const int c = 16 * s + 8 * half + j;
            float w = 0.f;
            if (type == 0) { if (c < k) w = W22[((int64_t)k * (k + 1) / 2 + c) * 64 + hh]; }
            else if (type == 1) { if (c < D) w = W22[((int64_t)c * (c + 1) / 2 + c) * 64 + hh]; }
            else { if (c < D) w = W21[(int64_t)c * 64 + hh]; }
            uint16_t* o = (uint16_t*)(out + (ht * 2) * 1024 + lane * 16) + j;
            split2h_bits(w_scale * w, o[0], o[512]);
        }
        return;
    }
    float v[4];                                                        // (256 threads: the loads of all four trips first)
#pragma unroll
    for (int i = 0; i < 4; ++i) {                                      // (ht, lane, j)
        const int e = threadIdx.x + 256 * i;
        const int j = e & 7, lane = (e >> 3) & 63, ht = e >> 9;
        const int half = lane >> 5, hh = 32 * ht + (lane & 31);
        const int c = 16 * s + 8 * half + j;
        v[i] = 0.f;
        if (type == 0) { if (c < k) v[i] = W22[((int64_t)k * (k + 1) / 2 + c) * 64 + hh]; }
        else if (type == 1) { if (c < D) v[i] = W22[((int64_t)c * (c + 1) / 2 + c) * 64 + hh]; }
        else { if (c < D) v[i] = W21[(int64_t)c * 64 + hh]; }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = threadIdx.x + 256 * i;
        const int j = e & 7, lane = (e >> 3) & 63, ht = e >> 9;
        uint16_t* o = (uint16_t*)(out + (ht * 2) * 1024 + lane * 16) + j;
        split2h_bits(w_scale * v[i], o[0], o[512]);
    }
}
__global__ void k_pack_heads_hb(int D, const float* __restrict__ W21, const float* __restrict__ W22,
                                const float* __restrict__ sc, uint8_t* __restrict__ img, uint32_t* __restrict__ maxw) {
    if (blockIdx.x == 0 && threadIdx.x < 4 && maxw) maxw[threadIdx.x] = 0u;
    pack_heads_hb_unit(blockIdx.x, D, W21, W22, sc[2], img);
}

// SPLIT (small batches, at most HB_SPLIT_MAX persons): a workgroup takes ONE 32-person tile and its eight waves share
// the units -- wave w the rows k = 1 + w, 9 + w, .. and the section units s = w -- each reading its units straight from
// the (L2-resident) image, one unit ahead; the eight partial gh tiles are summed through LDS in a fixed order.  The
// per-wave chain (357 units at D = 100) is what a small batch waits for.
#define HB_SPLIT_MAX 16384
template <bool SPLIT>
__global__ __launch_bounds__(HB_THREADS, 1) void k_mvn_enc_bwd_h_b(
    EncDims dm, const uint8_t* __restrict__ img, const float* __restrict__ sc /*k_enc_scales*/, const float* __restrict__ h_in,
    const float* __restrict__ eps_in, const float* __restrict__ gxT, const float* __restrict__ gdT /*DIAG-row operand [D][nb]*/,
    float* __restrict__ ghpre_out /*[nb][64] or null*/, const float* __restrict__ hT /*[64][nb], with ghpreT_out*/,
    float* __restrict__ ghpreT_out /*[64][nb] or null*/,
    uint32_t* __restrict__ maxw /*float bits: largest |gx|, |gd|, |eps|, |ghpre| of the launch, or null*/,
    int64_t i_base = 0 /*first person of this launch (a multiple of 32): the persons before it belong to another launch*/) {
    extern __shared__ __attribute__((aligned(16))) char smem_hb[];
    constexpr int H = 64;
    const int D = dm.D;
    const int64_t nb = dm.nb;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    // [D][32] of this wave; SPLIT: the eight waves work on the SAME 32 persons and share one tile (each loads an eighth of it)
    float* gx_lds = (float*)smem_hb + (SPLIT ? (size_t)0 : (size_t)wave * D * 32);
    const char* ring = smem_hb + (size_t)HB_WAVES * D * 32 * sizeof(float);
    const uint32_t ring_lds = lds_addr_uniform(ring);
    const int64_t i0 = i_base + (SPLIT ? (int64_t)blockIdx.x * 32 : ((int64_t)blockIdx.x * HB_WAVES + wave) * 32);
    const int64_t i = i0 + l31;
    const int64_t ic = i < nb ? i : nb - 1;                            // absent persons: a valid one, never stored
    const int n_units = hb_units(D);
    const int ns = (D + 15) / 16;

    // ---- weight ring: pair q = units 2q, 2q + 1 -> slots (2q) % 8, (2q + 1) % 8; this wave moves pieces w, w + 4, w + 8
    const uint32_t voff = (uint32_t)(wave * 1024 + lane * 16);
    auto stage_pair = [&](int q) __attribute__((always_inline)) {
        if (wave >= 4) return;                                         // waves 0..3 feed the ring for all eight
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int piece = wave + 4 * c;                            // 0..7: unit piece / 4, fragment piece % 4
            int u = 2 * q + (piece >= 4 ? 1 : 0);
            const uint32_t dst = ring_lds + (uint32_t)(u & (HB_NSLOT - 1)) * HB_UNIT_BYTES + (uint32_t)(piece % 4) * 1024u;
            if (u >= n_units) u = n_units - 1;                         // past the end: a harmless duplicate
            dma16s(img + (int64_t)u * HB_UNIT_BYTES, (uint32_t)((piece % 4) * 1024 + lane * 16), dst);
        }
    };
    (void)voff;
    if (!SPLIT) { stage_pair(0); stage_pair(1); stage_pair(2); }

    // diagnostic build (tools/hb2_bench.hip -DHB_STAMPS): cycles per phase of one workgroup of the SPLIT form
#ifdef HB_STAMPS
    uint64_t hst_[8]; int hsn_ = 0;
#define HSTAMP() do { __builtin_amdgcn_s_waitcnt(0); hst_[hsn_++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define HSTAMP() do {} while (0)
#endif
    HSTAMP();
    // ---- B fragments: contraction index c = 16 s + 8 half + j of person ic as two fp16 terms of v 2^e; e from the largest
    // magnitude among the wave's persons (two passes over the values: the maximum, then the split)
    const float w_inv = 1.0f / sc[2];                                  // 2^-sw (a power of two: exact)
    f16x8 bf[2][HB_NS];
    float e_max = 0.f;
    {
        const float* er = eps_in + ic * D;
        auto load8 = [&](int s, float (&v)[8]) __attribute__((always_inline)) {
            const int c0 = 16 * s + 8 * half;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                f32x4 t = {0.f, 0.f, 0.f, 0.f};
                if (c0 + 4 * q + 4 <= D) t = *(const f32x4*)(er + c0 + 4 * q);      // D % 4 == 0 on this path
                v[4 * q + 0] = t[0]; v[4 * q + 1] = t[1]; v[4 * q + 2] = t[2]; v[4 * q + 3] = t[3];
            }
        };
        // (one pass: the values stay in registers between the maximum and the split -- read twice they were two chains of
        // loads; tools/hb2_bench.hip -DHB_STAMPS)
        float vv[HB_NS][8];
        float m = 0.f;
#pragma unroll
        for (int s = 0; s < HB_NS; ++s) load8(s, vv[s]);
#pragma unroll
        for (int s = 0; s < HB_NS; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(vv[s][j]));
        e_max = wave_max_dpp(m);
        const float e_scale = ldexpf(1.0f, f16_scale_exp(e_max));
#pragma unroll
        for (int s = 0; s < HB_NS; ++s) split2h_frag(vv[s], e_scale, bf[0][s], bf[1][s]);
    }
    const float u_inv = w_inv * ldexpf(1.0f, -f16_scale_exp(e_max));   // takes 2^(sw + se) off U_k, folded into gx[p][k]
    HSTAMP();                                                          // 1: eps fragments
    // ---- this wave's gx tile [D][32] (lanes = persons: 128-byte rows of gxT), times u_inv
    float g_max = 0.f;
    if constexpr (SPLIT) {
        // rows k = 16 j + 2 wave + half of the shared tile: at most eight loads a lane, all in flight together; published by the
        // barrier below (round 5: every wave used to load the whole tile for itself -- 8 k cycles of a 128 k-cycle workgroup)
        float t[HB_NS];
#pragma unroll
        for (int q = 0; q < HB_NS; ++q) {
            const int k = 16 * q + 2 * wave + half;
            t[q] = k < D ? gxT[(int64_t)k * nb + ic] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < HB_NS; ++q) {
            const int k = 16 * q + 2 * wave + half;
            g_max = fmaxf(g_max, fabsf(t[q]));
            if (k < D) gx_lds[k * 32 + l31] = t[q] * u_inv;          // (u_inv: the waves share the persons, hence e_max and u_inv)
        }
    } else {
    for (int k0 = 0; k0 < D; k0 += 64) {                               // 32 rows per lane half in flight together (one at a
        float t[32];                                                   // time this loop was a chain of 50 misses)
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const int k = k0 + half + 2 * q;
            t[q] = k < D ? gxT[(int64_t)k * nb + ic] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const int k = k0 + half + 2 * q;
            g_max = fmaxf(g_max, fabsf(t[q]));
            if (k < D) gx_lds[k * 32 + l31] = t[q] * u_inv;
        }
    }
    }
    g_max = wave_max_dpp(g_max);
    // fragments of a dimension-major operand; returns the power of two that takes its scale (and the weights') off
    auto frags_from_T = [&](const float* __restrict__ srcT, float& vmax) __attribute__((always_inline)) -> float {
        float vv[HB_NS][8];                                            // (one pass, every load in flight: as the eps fragments above)
        float m = 0.f;
#pragma unroll
        for (int s = 0; s < HB_NS; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = 16 * s + 8 * half + j;
                vv[s][j] = (c < D) ? srcT[(int64_t)c * nb + ic] : 0.f;
            }
#pragma unroll
        for (int s = 0; s < HB_NS; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(vv[s][j]));
        vmax = wave_max_dpp(m);
        const int se = f16_scale_exp(vmax);
        const float scl = ldexpf(1.0f, se);
#pragma unroll
        for (int s = 0; s < HB_NS; ++s) split2h_frag(vv[s], scl, bf[0][s], bf[1][s]);
        return w_inv * ldexpf(1.0f, -se);
    };

    f32x16 gh0 = zero16(), gh1 = zero16();
    f16x8 A[4];                                                        // fragments of the CURRENT unit: [ht * 2 + term]
    int u = 0;                                                         // unit counter (uniform)
    auto slot_of = [&](int uu) -> const char* { return ring + (size_t)(uu & (HB_NSLOT - 1)) * HB_UNIT_BYTES; };
    auto read_ht = [&](int uu, int ht) __attribute__((always_inline)) {
        const char* sb = slot_of(uu) + ht * 2048 + lane * 16;
        A[2 * ht + 0] = *(const f16x8*)(sb);
        A[2 * ht + 1] = *(const f16x8*)(sb + 1024);
    };
    // pair boundary: pair q + 1 has landed for every wave, the slots of pair q - 1 are free for pair q + 3
    auto sync_pair = [&](int q) __attribute__((always_inline)) {
        if (wave < 4) __builtin_amdgcn_s_waitcnt(0x0F72);              // vmcnt(2): only pair q + 2 may be in flight
        __syncthreads();
        stage_pair(q + 3);
    };
    // one unit: 3 products per hidden tile (small terms first); the fragments of the NEXT unit are requested as soon as
    // the registers are free: hidden tile 0 after the first three MFMAs, hidden tile 1 at the end
    auto unit = [&](auto sc, f32x16& U0, f32x16& U1) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value;
        if ((u & 1) == 0) sync_pair(u >> 1);
        U0 = mfma_f16(A[1], bf[0][s], U0);
        U0 = mfma_f16(A[0], bf[1][s], U0);
        U0 = mfma_f16(A[0], bf[0][s], U0);
        read_ht(u + 1, 0);
        U1 = mfma_f16(A[3], bf[0][s], U1);
        U1 = mfma_f16(A[2], bf[1][s], U1);
        U1 = mfma_f16(A[2], bf[0][s], U1);
        read_ht(u + 1, 1);
        ++u;
    };
    float d_max = 0.f, x_max = 0.f;                                    // wave maxima of |gd| and (again) |gx|

    if constexpr (SPLIT) {
        __syncthreads();                                               // the shared gx tile is complete
        HSTAMP();                                                      // 2: gx tile
        // ---- this wave's share of the units, fragments global -> registers one unit ahead
        auto uoff = [&](int k) -> int {                                // index of unit (k, 0): blocks of 16 k have kb + 1 units per k
            const int kb2 = (k - 1) >> 4;
            return 8 * kb2 * (kb2 + 1) + (k - 16 * kb2 - 1) * (kb2 + 1);
        };
        auto load_unit = [&](f16x8 (&Au)[4], int uu) __attribute__((always_inline)) {
            if (uu >= n_units) uu = n_units - 1;                       // past the end: a harmless duplicate
            const uint8_t* g = img + (int64_t)uu * HB_UNIT_BYTES + lane * 16;
#pragma unroll
            for (int f = 0; f < 4; ++f) Au[f] = *(const f16x8*)(g + f * 1024);
        };
        auto mma_unit = [&](const f16x8 (&Au)[4], auto sc, f32x16& U0, f32x16& U1) __attribute__((always_inline)) {
            constexpr int s = decltype(sc)::value;
            U0 = mfma_f16(Au[1], bf[0][s], U0);
            U0 = mfma_f16(Au[0], bf[1][s], U0);
            U0 = mfma_f16(Au[0], bf[0][s], U0);
            U1 = mfma_f16(Au[3], bf[0][s], U1);
            U1 = mfma_f16(Au[2], bf[1][s], U1);
            U1 = mfma_f16(Au[2], bf[0][s], U1);
        };
        // (round 5: THREE units ahead -- one ahead, a unit cost an L2 round trip, ~530 cycles for 192 of MFMA; the wave's units
        // are a fixed sequence -- k = 1 + wave, 9 + wave, .., the k-steps s = 0 .. (k - 1) / 16 of each -- walked by a cursor)
        f16x8 Ac[4], An[4], An2[4], An3[4];
        int nk = 1 + wave, nsx = 0;                                    // the unit to request next
        auto next_unit = [&]() __attribute__((always_inline)) -> int {
            const int uu = uoff(nk) + nsx;
            if (++nsx > ((nk - 1) >> 4)) { nsx = 0; nk += HB_WAVES; }
            return uu;
        };
        load_unit(Ac, next_unit());
        load_unit(An, next_unit());
        load_unit(An2, next_unit());
        static_for<HB_NS>([&](auto kbc) {
            constexpr int kb = decltype(kbc)::value;
            const int k_lo = 16 * kb + 1, k_hi = (16 * kb + 16 < D - 1) ? 16 * kb + 16 : D - 1;
            // the k of this wave in the block: k = 1 + wave (mod 8)
            for (int k = k_lo + ((wave - (k_lo - 1)) & (HB_WAVES - 1)); k <= k_hi; k += HB_WAVES) {
                f32x16 U0 = zero16(), U1 = zero16();
                static_for<kb + 1>([&](auto sc) {
                    load_unit(An3, next_unit());
                    mma_unit(Ac, sc, U0, U1);
#pragma unroll
                    for (int f = 0; f < 4; ++f) { Ac[f] = An[f]; An[f] = An2[f]; An2[f] = An3[f]; }
                });
                const float gk = gx_lds[k * 32 + l31];
#pragma unroll
                for (int r = 0; r < 16; ++r) { gh0[r] = fmaf(gk, U0[r], gh0[r]); gh1[r] = fmaf(gk, U1[r], gh1[r]); }
            }
        });
        HSTAMP();                                                      // 3: OFF units
        // ---- DIAG (operand gd) and LOC (operand gx) units: s = wave, wave + 8, ..
        // Wave w runs the ONE unit s = w of each section (ns <= 8 = the waves), so it needs the eight operand values of its own
        // k-step only -- sixteen loads a lane for both sections, not the 2 x 56 of a whole tile each (round 5: 20 k cycles of a
        // 128 k-cycle workgroup) -- and scales them by its own power of two: its partial sum carries its own cinv.
        const int u_sec = hb_units_off(D);
        if (wave < ns) {
            float vx_[8], vd_[8], mxv = 0.f, mdv = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = 16 * wave + 8 * half + j;
                vx_[j] = (c < D) ? gxT[(int64_t)c * nb + ic] : 0.f;
                vd_[j] = (c < D) ? gdT[(int64_t)c * nb + ic] : 0.f;
            }
            f16x8 Ad[4];
            load_unit(Ac, u_sec + wave);
            load_unit(Ad, u_sec + ns + wave);
#pragma unroll
            for (int j = 0; j < 8; ++j) { mxv = fmaxf(mxv, fabsf(vx_[j])); mdv = fmaxf(mdv, fabsf(vd_[j])); }
            x_max = wave_max_dpp(mxv);
            d_max = wave_max_dpp(mdv);
            const int sx = f16_scale_exp(x_max), sd = f16_scale_exp(d_max);
            f16x8 fxh, fxl, fdh, fdl;
            split2h_frag(vx_, ldexpf(1.0f, sx), fxh, fxl);
            split2h_frag(vd_, ldexpf(1.0f, sd), fdh, fdl);
            const float cx = w_inv * ldexpf(1.0f, -sx), cd = w_inv * ldexpf(1.0f, -sd);
            f32x16 S0 = zero16(), S1 = zero16(), T0 = zero16(), T1 = zero16();
            S0 = mfma_f16(Ac[1], fxh, S0); S0 = mfma_f16(Ac[0], fxl, S0); S0 = mfma_f16(Ac[0], fxh, S0);
            S1 = mfma_f16(Ac[3], fxh, S1); S1 = mfma_f16(Ac[2], fxl, S1); S1 = mfma_f16(Ac[2], fxh, S1);
            T0 = mfma_f16(Ad[1], fdh, T0); T0 = mfma_f16(Ad[0], fdl, T0); T0 = mfma_f16(Ad[0], fdh, T0);
            T1 = mfma_f16(Ad[3], fdh, T1); T1 = mfma_f16(Ad[2], fdl, T1); T1 = mfma_f16(Ad[2], fdh, T1);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                gh0[r] = fmaf(cx, S0[r], gh0[r]); gh1[r] = fmaf(cx, S1[r], gh1[r]);
                gh0[r] = fmaf(cd, T0[r], gh0[r]); gh1[r] = fmaf(cd, T1[r], gh1[r]);
            }
        }
        HSTAMP();                                                      // 4: section units
        // ---- sum of the eight partial tiles, fixed order; wave 0 keeps the result
        __syncthreads();                                               // every wave is done with its gx tile
        HSTAMP();                                                      // 5: barrier
        float* red = (float*)smem_hb;                                  // [8 waves][32 registers][64 lanes]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            red[((size_t)wave * 32 + r) * 64 + lane] = gh0[r];
            red[((size_t)wave * 32 + 16 + r) * 64 + lane] = gh1[r];
        }
        __syncthreads();
        if (lane == 0 && maxw) {                                       // (before the partial waves leave)
            atomic_max_raise(maxw + 0, fmaxf(g_max, x_max));
            atomic_max_raise(maxw + 1, d_max);
            atomic_max_raise(maxw + 2, e_max);
        }
#ifdef HB_STAMPS
        if (blockIdx.x == 1 && lane == 0 && wave != 0)
            printf("HSTAMPS blk 1 wave %d: epsfrag %llu gx %llu off %llu sec %llu bar %llu\n", wave, hst_[1] - hst_[0], hst_[2] - hst_[1],
                   hst_[3] - hst_[2], hst_[4] - hst_[3], hst_[5] - hst_[4]);
#endif
        // Round 5: the sum and the output are divided between the eight waves -- wave w takes the four accumulator registers
        // 4 g .. 4 g + 3 of hidden tile ht (w = 4 ht + g: the hidden units 32 ht + 8 g + 4 half + 0..3 of its lanes' persons),
        // 32 LDS reads a lane instead of wave 0's 224 with seven waves gone (63 k of a 128 k-cycle workgroup), in the same
        // fixed order w' = 0..7.
        {
            const int ht = wave >> 2, g = wave & 3;
            float a4[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float a = red[((size_t)(16 * ht + 4 * g + c)) * 64 + lane];
#pragma unroll
                for (int w = 1; w < HB_WAVES; ++w) a += red[((size_t)w * 32 + 16 * ht + 4 * g + c) * 64 + lane];
                a4[c] = a;
            }
            const int hh0 = 32 * ht + 8 * g + 4 * half;
            float p_max = 0.f;
            if (i < nb) {
                const float4 hv = *(const float4*)(h_in + i * H + hh0);
                float4 o;
                o.x = a4[0] * (1.0f - __expf(-hv.x));
                o.y = a4[1] * (1.0f - __expf(-hv.y));
                o.z = a4[2] * (1.0f - __expf(-hv.z));
                o.w = a4[3] * (1.0f - __expf(-hv.w));
                p_max = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w)));
                if (ghpreT_out) {                                      // dimension-major rows hh0 .. hh0 + 3
                    ghpreT_out[(int64_t)(hh0 + 0) * nb + i] = o.x; ghpreT_out[(int64_t)(hh0 + 1) * nb + i] = o.y;
                    ghpreT_out[(int64_t)(hh0 + 2) * nb + i] = o.z; ghpreT_out[(int64_t)(hh0 + 3) * nb + i] = o.w;
                } else {
                    *(float4*)(ghpre_out + i * H + hh0) = o;
                }
            }
            if (ghpreT_out) {
                p_max = wave_max_dpp(p_max);
                if (lane == 0 && maxw) atomic_max_raise(maxw + 3, p_max);
            }
        }
        return;
    } else {
    vx_wait_vmem();                                                    // pairs 0..2 of the ring (and nothing else)
    __syncthreads();
    read_ht(0, 0);
    read_ht(0, 1);
    // NOTE: sync_pair(0) at unit 0 waits vmcnt(3) with nothing in flight and stages pair 3

    // ---- OFF rows: k in blocks of 16; block kb has kb + 1 units per k
    static_for<HB_NS>([&](auto kbc) {
        constexpr int kb = decltype(kbc)::value;
        const int k_lo = 16 * kb + 1, k_hi = (16 * kb + 16 < D - 1) ? 16 * kb + 16 : D - 1;
        for (int k = k_lo; k <= k_hi; ++k) {
            f32x16 U0 = zero16(), U1 = zero16();
            static_for<kb + 1>([&](auto sc) { unit(sc, U0, U1); });
            const float gk = gx_lds[k * 32 + l31];
#pragma unroll
            for (int r = 0; r < 16; ++r) { gh0[r] = fmaf(gk, U0[r], gh0[r]); gh1[r] = fmaf(gk, U1[r], gh1[r]); }
        }
    });
    // ---- LOC rows (operand gx) and DIAG rows (operand gd): an accumulator pair of their own, added with their power of two
    {
        const float cinv = frags_from_T(gxT, x_max);
        f32x16 S0 = zero16(), S1 = zero16();
        static_for<HB_NS>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            if (s < ns) unit(sc, S0, S1);
        });
#pragma unroll
        for (int r = 0; r < 16; ++r) { gh0[r] = fmaf(cinv, S0[r], gh0[r]); gh1[r] = fmaf(cinv, S1[r], gh1[r]); }
    }
    {
        const float cinv = frags_from_T(gdT, d_max);
        f32x16 S0 = zero16(), S1 = zero16();
        static_for<HB_NS>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            if (s < ns) unit(sc, S0, S1);
        });
#pragma unroll
        for (int r = 0; r < 16; ++r) { gh0[r] = fmaf(cinv, S0[r], gh0[r]); gh1[r] = fmaf(cinv, S1[r], gh1[r]); }
    }
    vx_wait_vmem();                                                    // no DMA may be in flight when the LDS is released
    if (lane == 0 && maxw) {
        atomic_max_raise(maxw + 0, fmaxf(g_max, x_max));
        atomic_max_raise(maxw + 1, d_max);
        atomic_max_raise(maxw + 2, e_max);
    }
    }

    // ---- ghpre = gh * softplus'(pre) = gh * (1 - exp(-h));  C layout: rows hh = crow32(r, half), cols p
    if (ghpreT_out) {                                                  // dimension-major: 128-byte rows per half-wave
        float p_max = 0.f;
        if (i < nb) {
#pragma unroll
            for (int ht = 0; ht < 2; ++ht)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t o = (int64_t)(32 * ht + crow32(r, half)) * nb + i;
                    const float gp = (ht ? gh1 : gh0)[r] * (1.0f - __expf(-hT[o]));
                    p_max = fmaxf(p_max, fabsf(gp));
                    ghpreT_out[o] = gp;
                }
        }
        p_max = wave_max_dpp(p_max);
        if (lane == 0 && maxw) atomic_max_raise(maxw + 3, p_max);
    } else if (i < nb) {
#pragma unroll
        for (int ht = 0; ht < 2; ++ht)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int hh0 = 32 * ht + 8 * g + 4 * half;
                const float4 hv = *(const float4*)(h_in + i * H + hh0);
                float4 o;
                o.x = (ht ? gh1 : gh0)[4 * g + 0] * (1.0f - __expf(-hv.x));
                o.y = (ht ? gh1 : gh0)[4 * g + 1] * (1.0f - __expf(-hv.y));
                o.z = (ht ? gh1 : gh0)[4 * g + 2] * (1.0f - __expf(-hv.z));
                o.w = (ht ? gh1 : gh0)[4 * g + 3] * (1.0f - __expf(-hv.w));
                *(float4*)(ghpre_out + i * H + hh0) = o;
            }
    }
#ifdef HB_STAMPS
    if (SPLIT && blockIdx.x == 1 && lane == 0) {
        HSTAMP();
        printf("HSTAMPS blk 1 wave 0: epsfrag %llu gx %llu off %llu sec %llu bar %llu reduce+out %llu total %llu\n", hst_[1] - hst_[0], hst_[2] - hst_[1],
               hst_[3] - hst_[2], hst_[4] - hst_[3], hst_[5] - hst_[4], hst_[6] - hst_[5], hst_[6] - hst_[0]);
    }
#endif
}
