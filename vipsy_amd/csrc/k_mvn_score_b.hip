// Score-function operands of the amortized multivariate guide on the fp16 MFMA: u = L^-T eps for every person of a large
// batch, where the rows of L come from the encoder heads (M_jc = W22[(j, c)] . h + b22[(j, c)], never stored) -- what
// k_mvn_score_operands<0> (k_mvn_score.hip) computes one lane a person in scalar fp32 (323 k multiply-adds and as many LDS
// reads a person: measured 355 ms at 1M x 100 dims, 39 times the pathwise step).  BASELINE.json's north_star names the
// estimator; the reference's own guides are pathwise (vi.py:693: MultivariateNormal has rsample), SURVEY.md F5 / App. A.5.
//
// Back-substitution by COLUMNS of L, last column first:
//     u_c = (eps_c - sum_{j > c} L_jc u_j) / L_cc,      L_cc = exp(M_cc),      c = D - 1, D - 2, .., 0
// is the forward's head loop (k_mvn_fwd_b.hip) with the weight rows in another order and u in the place of eps: a tile of 32
// packed rows (j = jb .. jb + 31 of column c) x 32 persons is ONE chain of 13 MFMAs (f16x2: bias, lo hi, hi lo, hi hi over the
// 64 hidden units) against the person's h fragments, which a wave splits once and keeps; the accumulator layout has the person
// on the lane and 4 consecutive j in 4 consecutive registers, so the dot product with u_j is 16 FMAs against four 16-byte reads
// of the wave's own u tile in LDS.  When the column ends the two lane halves are joined, u_c is written into the tile -- the
// next column's tiles multiply by it -- and the operands of the guide-backward kernels leave for global memory:
// gxT[c][i] = w_i u_ic, gdT[c][i] = w_i (u_ic eps_ic L_cc - 1).
//
// Column image (k_pack_heads_col, once a call, ~2 MB for D = 100): column c = D - 1 .. 0, its rows from jb0 = c & ~3 in tiles of
// 32 -- rows jb0 .. c - 1 are ZERO rows (the tile base stays a multiple of four, the u reads stay 16-byte aligned, and a zero row
// times a not-yet-final u_j is zero), row c is the DIAGONAL row (c, c), rows past D - 1 are zero rows -- each tile in the
// forward's format (eight 1 KB fragments + a bias fragment, FB_IMG_BYTES), with the forward's powers of two (the scale block of
// packws: same parameters, same step).  The diagonal row rides in the first tile of its column: M_cc comes out of accumulator
// register c & 3 of lane half 0, so L_cc needs neither a tile of its own in LDS nor the forward's ldT -- and without that tile a
// workgroup is 68 KB of LDS: TWO workgroups a CU, two waves a SIMD (rules 12 / 26).  Its product with the tile's u row (still
// eps_c there) is taken back out of the dot product.
// Two forms (template RING).  RING = false, batches under 16 384 persons or when the LDS does not hold the other: four waves of
// 32 persons a workgroup, every wave streams the tiles by itself, L2 -> registers, one tile ahead (the forward's plain form).
// RING = true: ONE workgroup a CU of eight consumer waves (two a SIMD) and a ninth LOADER wave that moves tile t + 2 by LDS-DMA
// into the slot tile t has left (two 9 KB slots behind the eight u tiles: 154 KB) and waits for it before the barrier that ends
// iteration t -- a tile crosses the L2 -> CU path once a workgroup instead of eight times (60 -> 7.5 GB a launch at 1M persons).
// The loader issues nothing but transfers, so its `vmcnt(0)` counts them alone; the consumers' column stores stay in flight as
// long as they like.  Measured at 1M x 100 (bench.py --estimator score, alternating runs on one box):
//     an L tile in LDS beside the u tile (L_cc from the forward's ldT), one wave a SIMD                       3.71 ms
//     ... with the tiles through a three-slot ring whose CONSUMERS moved their own fragments (counted vmcnt:
//         the column stores, a microsecond in flight, stood in front of every tile)                           4.15
//     the diagonal row in the image, two workgroups a CU, plain form; u reads where they are used             2.74-2.78
//     ... the u reads at the head of the iteration, exp / rcp of L_cc at the column's START                   2.77-2.80
//     RING (loader wave), u reads where used                                                                  2.78
//     RING, u reads at the head, exp / rcp at the column's start  (ships for large batches)                   2.63-2.65
// The L2 was NOT what the plain form waited for (an eighth of the traffic, the same time): at two waves a SIMD a wave's in-order
// issue stream is the limit -- thirteen dependent MFMAs (416 cycles), then ~100 instructions of epilogue, column bookkeeping and
// branches during which it issues no MFMA; the matrix pipe is 50 % busy in both forms (profiles/r06_score_pmc_counters.json).
#pragma once
// (included by vx_abi.hip behind k_mvn_fwd_b.hip: the tile format FB_*, split2h_bits and the scale block are its own)

#define SB_THREADS 256
#define SB_WAVES 4
#define SB_WP 32

__host__ __device__ inline int sb_col_jb0(int c) { return c & ~3; }
__host__ __device__ inline int sb_col_tiles(int D, int c) { return (D - sb_col_jb0(c) + 31) / 32; }
__host__ __device__ inline int sb_tiles(int D) {
    int n = 0;
    for (int c = 0; c < D; ++c) n += sb_col_tiles(D, c);
    return n;
}
// u tile: [32 persons][DS] floats, DS >= D + 32 (a tile may reach 31 rows past D - 1: zero rows against zeros), DS / 4 odd (the
// 16 lanes of a 16-byte LDS read fall on 16 different slots of the bank row)
__host__ __device__ inline int sb_ds(int D) {
    int ds = (D + 32 + 3) & ~3;
    if (((ds >> 2) & 1) == 0) ds += 4;
    return ds;
}
__host__ __device__ inline size_t sb_lds_bytes(int D) { return (size_t)SB_WAVES * SB_WP * sb_ds(D) * sizeof(float); }
// RING form: eight consumer waves (u tiles) + one loader wave, the column tiles through two slots in LDS
#define SBR_CONSUMERS 8
#define SBR_THREADS (64 * (SBR_CONSUMERS + 1))
__host__ __device__ inline size_t sbr_lds_bytes(int D) {
    return (size_t)SBR_CONSUMERS * SB_WP * sb_ds(D) * sizeof(float) + 2 * (size_t)FB_IMG_BYTES;
}
__host__ __device__ inline int64_t sb_img_floats(int D) { return (int64_t)sb_tiles(D) * (FB_IMG_BYTES / 4); }

// tile t of the image: blockIdx.x = t.  (c, tile within the column) by a walk over the columns -- 100 trips of scalar
// arithmetic in a kernel of 200 blocks that runs once a call
__global__ __launch_bounds__(256) void k_pack_heads_col(int D, const float* __restrict__ W22, const float* __restrict__ b22,
                                                       const float* __restrict__ sc, uint8_t* __restrict__ img) {
    int t = blockIdx.x, c = D - 1;
    while (c > 0 && t >= sb_col_tiles(D, c)) { t -= sb_col_tiles(D, c); --c; }
    const int jb = sb_col_jb0(c) + 32 * t;
    const float w_scale = sc[2], b_scale = sc[5];
    uint8_t* out = img + (int64_t)blockIdx.x * FB_IMG_BYTES;
    const int tid = threadIdx.x;
    const int row = (tid >> 3) & 31;                                         // the thread's row for every element it makes
    const int j = jb + row;
    const bool live = j >= c && j < D;                                       // j == c: the diagonal row
    const int64_t src = (int64_t)j * (j + 1) / 2 + c;                        // row-major lower triangle (torch.tril_indices, vi.py:453)
    const float* wrow = W22 + src * 64;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = tid + 256 * i;
        const int jj = e & 7, lane = (e >> 3) & 63, sx = e >> 9;
        const int half = lane >> 5;
        v[i] = live ? wrow[16 * sx + 8 * (jj >> 2) + 4 * half + (jj & 3)] : 0.f;
    }
    const float bv = (live && (tid & 7) < 2) ? b22[src] : 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = tid + 256 * i;
        const int jj = e & 7, lane = (e >> 3) & 63, sx = e >> 9;
        uint16_t* o = (uint16_t*)(out + sx * 1024 + lane * 16) + jj;
        split2h_bits(w_scale * v[i], o[0], o[2048]);                         // + 4 fragments = 4096 bytes: the remainders
    }
    for (int e = tid; e < FB_AUX_BYTES / 2; e += 256) {                      // bias fragment: lane = row (half 0), elements 0, 1
        const int jj = e & 7;
        uint16_t w = 0;
        if (e < 256 && jj < 2) {
            uint16_t bh, bl;
            split2h_bits(b_scale * bv, bh, bl);
            w = jj == 0 ? bh : bl;
        }
        ((uint16_t*)(out + FB_A_BYTES))[e] = w;
    }
}

template <bool RING>
__global__ __launch_bounds__(RING ? SBR_THREADS : SB_THREADS, RING ? 1 : 2) void k_mvn_score_b(
    int D, int64_t nb, float scale, const int64_t* __restrict__ rows, const float* __restrict__ h /*[nb][64]*/,
    const uint8_t* __restrict__ img, const float* __restrict__ sc, const float* __restrict__ eps /*[nb][D]*/,
    const float* __restrict__ ll, const float* __restrict__ ent, float* __restrict__ baseline, float base_beta, int base_by_row,
    float* __restrict__ log_r_out, float* __restrict__ gxT /*[D][nb]*/, float* __restrict__ gdT /*[D][nb] or null*/) {
    extern __shared__ __attribute__((aligned(16))) float sb_lds[];
    const int DS = sb_ds(D);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, p = lane & 31;
    constexpr int NCW = RING ? SBR_CONSUMERS : SB_WAVES;                    // waves with persons
    const int n_tiles = sb_tiles(D);
    typedef __attribute__((address_space(3))) uint8_t lds_u8;
    lds_u8* const ring = (lds_u8*)(sb_lds + (size_t)NCW * SB_WP * DS);      // RING: two tile slots behind the u tiles
    // LDS accesses of this wave done, then the workgroup barrier (no release fence: nobody waits for the column stores)
    auto turn = [&]() __attribute__((always_inline)) {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xC07F);                                  // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    if constexpr (RING) {
        if (wave == NCW) {
            // ---- the loader wave: tile t + 2 into the slot tile t has left, landed before the barrier that ends iteration t.
            // It issues nothing but these transfers, so its vmcnt counts them alone (a consumer's column stores, in flight for
            // a microsecond, never stand in front of a tile: the three-slot ring whose consumers moved their own fragments
            // measured 4.15 ms for that reason)
            const uint32_t ring_l = lds_addr_uniform((const void*)(sb_lds + (size_t)NCW * SB_WP * DS));
            auto stage = [&](int t) __attribute__((always_inline)) {
                const int tc = t < n_tiles ? t : n_tiles - 1;                // past the end: a harmless duplicate
                const uint8_t* gb = img + (int64_t)tc * FB_IMG_BYTES + lane * 16;
                const uint32_t dst = ring_l + (uint32_t)(t & 1) * FB_IMG_BYTES;
#pragma unroll
                for (int f = 0; f < 9; ++f) dma16(gb + f * 1024, __builtin_amdgcn_readfirstlane(dst + (uint32_t)f * 1024u));
            };
            stage(0); stage(1);
            vx_wait_vmem();
            turn();                                                          // B1: tiles 0 and 1 have landed
            turn();                                                          // B2: every consumer holds tile 0
            for (int t = 0; t < n_tiles; ++t) {
                stage(t + 2);
                vx_wait_vmem();
                turn();
            }
            return;
        }
    }
    float* const u_t = sb_lds + (size_t)wave * SB_WP * DS;                   // [32][DS]: eps, then u column by column
    const int64_t i0 = ((int64_t)blockIdx.x * NCW + wave) * SB_WP;
    if constexpr (!RING)
        if (i0 >= nb) return;                                                // (no workgroup barrier in this form: a wave may leave)
    // (RING: a wave past the last person stays for the barriers, on clamped inputs, and stores nothing)
    const int64_t i = i0 + p;
    const bool live = i < nb;
    const int64_t ic = live ? i : nb - 1;                                    // absent persons: the last one, never stored
    const float h_scale = sc[3], acc_inv = sc[4];

    // ---- w_i = log_r_i - baseline_i (both lane halves hold it; half 0 owns the side effects)
    float w;
    {
        const float lr = scale * (ll[ic] + ent[ic]);
        w = lr;
        if (baseline) {
            const int64_t bi = (base_by_row && rows) ? rows[ic] : ic;
            const float bv = baseline[bi];
            w = lr - bv;
            if (live && half == 0 && base_beta >= 0.f) baseline[bi] = fmaf(base_beta, bv, (1.0f - base_beta) * lr);
        }
        if (live && half == 0 && log_r_out) log_r_out[i] = lr;
    }
    // ---- the person's h as the B fragments of the four k-steps: element j of k-step s = hidden unit 16 s + 8 (j >> 2) + 4 half + (j & 3)
    f16x8 hb[2][4];
    {
        const float* hr = h + ic * 64 + 4 * half;
        f32x4 q[8];
#pragma unroll
        for (int s = 0; s < 4; ++s) { q[2 * s] = *(const f32x4*)(hr + 16 * s); q[2 * s + 1] = *(const f32x4*)(hr + 16 * s + 8); }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float v[8] = {q[2 * s][0], q[2 * s][1], q[2 * s][2], q[2 * s][3], q[2 * s + 1][0], q[2 * s + 1][1], q[2 * s + 1][2], q[2 * s + 1][3]};
            split2h_frag(v, h_scale, hb[0][s], hb[1][s]);
        }
    }
    // ---- the wave's tile: eps, zero past D
    {
        const int c4 = D >> 2, z4 = (DS - D) >> 2;                           // D % 4 == 0
        for (int e = lane; e < SB_WP * c4; e += 64) {
            const int pp = e / c4, cq = e - pp * c4;
            int64_t ii = i0 + pp;
            ii = ii < nb ? ii : nb - 1;
            *(f32x4*)(u_t + pp * DS + 4 * cq) = *(const f32x4*)(eps + ii * D + 4 * cq);
        }
        for (int e = lane; e < SB_WP * z4; e += 64) {
            const int pp = e / z4, cq = e - pp * z4;
            *(f32x4*)(u_t + pp * DS + D + 4 * cq) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);                                      // lgkmcnt(0): the tile is written
    float* const up = u_t + p * DS;

    struct TileRegs { f16x8 a[2][4]; f16x8 bias; };
    auto pull = [&](TileRegs& R, int t) __attribute__((always_inline)) {
        if constexpr (RING) {
            lds_u8* lb = ring + (t & 1) * FB_IMG_BYTES + lane * 16;
            R.bias = *(const __attribute__((address_space(3))) f16x8*)(lb + FB_A_BYTES);
#pragma unroll
            for (int sp = 1; sp >= 0; --sp)
#pragma unroll
                for (int s = 0; s < 4; ++s) R.a[sp][s] = *(const __attribute__((address_space(3))) f16x8*)(lb + (sp * 4 + s) * 1024);
        } else {
            const int tc = t < n_tiles ? t : n_tiles - 1;                    // past the end: a harmless duplicate
            const uint8_t* gb = img + (int64_t)tc * FB_IMG_BYTES + lane * 16;
            R.bias = *(const f16x8*)(gb + FB_A_BYTES);
#pragma unroll
            for (int sp = 1; sp >= 0; --sp)
#pragma unroll
                for (int s = 0; s < 4; ++s) R.a[sp][s] = *(const f16x8*)(gb + (sp * 4 + s) * 1024);
        }
    };
    f16x8 cfrag;                                                             // the bias product's constant 2^(sw + sh - sb)
    {
        const _Float16 c16 = (_Float16)sc[6];
#pragma unroll
        for (int j = 0; j < 8; ++j) cfrag[j] = c16;
    }
    // cursor of the tile whose MFMAs run (column c, row base jb, tiles left in the column, first tile of the column?) and of the
    // tile before it, whose epilogue runs
    int c = D - 1, jb = sb_col_jb0(c), left = sb_col_tiles(D, c);
    bool fst = true;
    int cP = 0, jbP = 0;
    bool endP = false, fstP = false;
    float dot = 0.f, lcc = 1.f, rlc = 1.f, e_c = 0.f;                        // (lcc = exp(M_cc), rlc = 1 / lcc: lane half 0)
    f32x16 accP = zero16();
    const float* up_h = up + 4 * half;
    // the four 16-byte reads of u for the tile before: issued at the HEAD of an iteration, consumed between its MFMA groups (read
    // where they are used, every group put an LDS latency into the wave's in-order issue stream in front of the next MFMAs:
    // ~1 640 cycles a tile and wave for 416 of matrix pipe)
    f32x4 uq[4];
    auto epi_read = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < 4; ++g) uq[g] = *(const f32x4*)(up_h + jbP + 8 * g);
    };
    auto epi_group = [&](int g) __attribute__((always_inline)) {
        dot = fmaf(accP[4 * g + 0], uq[g][0], dot);
        dot = fmaf(accP[4 * g + 1], uq[g][1], dot);
        dot = fmaf(accP[4 * g + 2], uq[g][2], dot);
        dot = fmaf(accP[4 * g + 3], uq[g][3], dot);
    };
    // the tile before is the FIRST of its column: its row cP - jbP (0..3: accumulator register cP & 3 of lane half 0) is the
    // diagonal row -- keep M_cc, and take its product with the tile's u row (eps_c: the column is not solved yet) out of the sum
    auto column_start = [&]() __attribute__((always_inline)) {
        const int q = cP & 3;                                                // wave-uniform
        const float m = q == 0 ? accP[0] : (q == 1 ? accP[1] : (q == 2 ? accP[2] : accP[3]));
        e_c = up[cP];
        if (half == 0) {
            dot = fmaf(-m, e_c, dot);
            lcc = __expf(m * acc_inv);                                       // (two transcendentals: made HERE, tiles before the column ends)
            rlc = fast_rcp(lcc);
        }
    };
    // a column is done: join the lane halves, u_c into the tile, the operands of the backward kernels to global memory
    auto column_end = [&]() __attribute__((always_inline)) {
        const float tot = half_sum32(dot) * acc_inv;
        if (half == 0) {
            const float uc = (e_c - tot) * rlc;
            up[cP] = uc;
            if (live) {
                const float g = w * uc;
                gxT[(int64_t)cP * nb + i] = g;
                if (gdT) gdT[(int64_t)cP * nb + i] = fmaf(g * e_c, lcc, -w);
            }
        }
        dot = 0.f;
    };
    auto tile_iter = [&](TileRegs& Rc, TileRegs& Rn, int t, auto firstc) __attribute__((always_inline)) {
        constexpr bool first = decltype(firstc)::value;
        if constexpr (!first) epi_read();
        pull(Rn, t + 1);
        f32x16 a = mfma_f16(Rc.bias, cfrag, zero16());
        a = mfma_f16(Rc.a[1][0], hb[0][0], a);
        a = mfma_f16(Rc.a[1][1], hb[0][1], a);
        if constexpr (!first) epi_group(0);
        a = mfma_f16(Rc.a[1][2], hb[0][2], a);
        a = mfma_f16(Rc.a[1][3], hb[0][3], a);
        a = mfma_f16(Rc.a[0][0], hb[1][0], a);
        if constexpr (!first) epi_group(1);
        a = mfma_f16(Rc.a[0][1], hb[1][1], a);
        a = mfma_f16(Rc.a[0][2], hb[1][2], a);
        a = mfma_f16(Rc.a[0][3], hb[1][3], a);
        if constexpr (!first) epi_group(2);
        a = mfma_f16(Rc.a[0][0], hb[0][0], a);
        a = mfma_f16(Rc.a[0][1], hb[0][1], a);
        if constexpr (!first) epi_group(3);
        a = mfma_f16(Rc.a[0][2], hb[0][2], a);
        a = mfma_f16(Rc.a[0][3], hb[0][3], a);
        if constexpr (!first) {
            if (fstP) column_start();                                        // (wave-uniform branches)
            if (endP) column_end();
        }
        accP = a;
        cP = c; jbP = jb; endP = (left == 1); fstP = fst;
        if (left == 1) {                                                     // (scalar arithmetic: the cursor is wave-uniform)
            --c;
            jb = sb_col_jb0(c);
            left = c >= 0 ? sb_col_tiles(D, c) : 1;
            fst = true;
        } else {
            jb += 32;
            --left;
            fst = false;
        }
        if constexpr (RING) turn();                                          // tile t + 1 is in everybody's registers, t + 2 has landed
    };
    TileRegs RA, RB;
    if constexpr (RING) {
        turn();                                                              // B1
        pull(RA, 0);
        turn();                                                              // B2
    } else {
        pull(RA, 0);
    }
    tile_iter(RA, RB, 0, std::true_type{});
    int t = 1;
    for (; t + 1 < n_tiles; t += 2) {
        tile_iter(RB, RA, t, std::false_type{});
        tile_iter(RA, RB, t + 1, std::false_type{});
    }
    if (t < n_tiles) tile_iter(RB, RA, t, std::false_type{});
    epi_read();
    epi_group(0); epi_group(1); epi_group(2); epi_group(3);                  // the last tile: column 0 (one tile when D <= 32)
    if (fstP) column_start();
    column_end();
}
