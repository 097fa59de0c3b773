// Hidden-layer gradient of the amortized MVN guide, SIXTY-FOUR persons per wave: the same mathematics, operands and
// outputs as k_mvn_enc_bwd_h_b<false> (k_mvn_bwd_hb.hip), for batches that fill the chip with 256-person workgroups and
// D <= 112.
//
// What bounded k_mvn_enc_bwd_h_b once its products fell from ten to six MFMAs a unit (f16x2) was not the matrix pipe
// (busy 0.31): eight waves met at a barrier every two units, the fragments of a unit were read from LDS at the end of
// the unit before it (their latency exposed once a unit), and the per-k epilogue waited for the chain it scaled.  Here
//   * every unit fragment feeds TWO 32-person MFMA chains (person sets 0 and 1 of the wave): 12 MFMAs per 4 ds_read_b128;
//   * the unit images stream through a ring of 12 LDS slots in batches of FOUR units: one barrier per 48 MFMAs, and each
//     fragment of unit u + 1 is read right after the last MFMA of unit u that takes its register (>= 8 MFMAs ahead);
//   * the rows k alternate between two accumulator sets, so the 64-FMA epilogue of row k - 1 (gh += gx_k U_k) sits between
//     the MFMA pairs of the first unit of row k;
//   * one wave per SIMD (the kernel takes ~400 registers); nothing but MFMAs, LDS reads and the epilogue in the loop.
// (included by vx_abi.hip after k_mvn_bwd_hb.hip, whose unit images, scales and helpers it uses)

// NSET person sets of 32 per wave, 8 / NSET waves per workgroup (256 persons either way).  NSET = 2: four waves of 64
// persons, one per SIMD, accumulators in AGPRs (the form described above).  NSET = 1 (round 3, after the same change paid in
// k_mvn_enc_bwd_w_b): eight waves of 32 persons, two per SIMD, under 256 registers -- the MFMAs take the VGPR form, so the
// epilogue reads its accumulators directly (32 FMAs a row and wave, no AGPR reads), and one wave's epilogue / LDS waits run
// under its SIMD partner's MFMAs.
#define HB2_WAVES_OF(NSET) (8 / (NSET))
#define HB2_WP_OF(NSET) (32 * (NSET))
#define HB2_NSLOT 12                                                   // three batches of four units
#define HB2_BATCH 4

__host__ __device__ inline size_t hb2_lds_bytes(int D) {      // operand tiles [roundup8(D)][256 persons] | ring
    return (size_t)((D + 7) & ~7) * 256 * sizeof(float) + (size_t)HB2_NSLOT * HB_UNIT_BYTES;
}

template <int NS, int NSET>
__global__ __launch_bounds__(64 * HB2_WAVES_OF(NSET), 1) void k_mvn_enc_bwd_h_b2(
    EncDims dm, const uint8_t* __restrict__ img, const float* __restrict__ sc /*k_enc_scales*/, const float* __restrict__ h_in,
    const float* __restrict__ eps_in, const float* __restrict__ gxT, const float* __restrict__ gdT /*DIAG-row operand [D][nb]*/,
    float* __restrict__ ghpre_out /*[nb][64] or null*/, const float* __restrict__ hT /*[64][nb], with ghpreT_out*/,
    float* __restrict__ ghpreT_out /*[64][nb] or null*/,
    uint32_t* __restrict__ maxw /*float bits: largest |gx|, |gd|, |eps|, |ghpre| of the launch, or null*/) {
    extern __shared__ __attribute__((aligned(16))) char smem_h2[];
    constexpr int H = 64, HB2_WAVES = HB2_WAVES_OF(NSET), HB2_WP = HB2_WP_OF(NSET);
    constexpr int VM_OWN = 16 / HB2_WAVES;                             // ring transfers of one wave per batch
    constexpr int TROWS = 256 / HB2_WP;                                // operand rows per 1 KB tile transfer
    const int D = dm.D;
    const int64_t nb = dm.nb;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int DR = (D + 7) & ~7;                                       // tile rows (whole transfers)
    float* gx_lds = (float*)smem_h2 + (size_t)wave * DR * HB2_WP;      // [DR][WP] of this wave
    const char* ring = smem_h2 + (size_t)DR * 256 * sizeof(float);
    const uint32_t ring_lds = lds_addr_uniform(ring);
    const int64_t i0 = ((int64_t)blockIdx.x * HB2_WAVES + wave) * HB2_WP;
    int64_t iu[NSET], ic[NSET];
#pragma unroll
    for (int u = 0; u < NSET; ++u) {
        iu[u] = i0 + 32 * u + l31;
        ic[u] = iu[u] < nb ? iu[u] : nb - 1;                           // absent persons: a valid one, never stored
    }
    const int n_units = hb_units(D);
    const int ns = (D + 15) / 16;
    // diagnostic build (make EXTRA=-DHB2_STAMPS): cycles per phase of two workgroups, printed at the end; in the shipped
    // build no stamp executes
#ifdef HB2_STAMPS
    uint64_t st_[10]; int sn_ = 0;
#define HSTAMP() do { __builtin_amdgcn_s_waitcnt(0); st_[sn_++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define HSTAMP() do {} while (0)
#endif
    HSTAMP();

    // ---- weight ring: batch b = units 4b .. 4b + 3 -> slots 4 (b % 3) + c; wave w moves fragment w of each of the four.
    // Batches are staged in order, so the slot group and the image offset are running counters.
    int st_b3 = 0;                                                     // (next batch to stage) % 3
    int st_u = 0;                                                      // its first unit
    auto stage_next = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < VM_OWN; ++j) {                              // transfer t = (unit c, fragment f) of the batch
            const int t = wave + HB2_WAVES * j, c = t >> 2, f = t & 3;
            int u = st_u + c;
            const uint32_t dst = ring_lds + (uint32_t)(4 * st_b3 + c) * HB_UNIT_BYTES + (uint32_t)f * 1024u;
            if (u >= n_units) u = n_units - 1;                         // past the end: a harmless duplicate
            dma16s(img + (int64_t)u * HB_UNIT_BYTES, (uint32_t)(f * 1024 + lane * 16), dst);
        }
        st_u += HB2_BATCH;
        st_b3 = (st_b3 == 2) ? 0 : st_b3 + 1;
    };

    // ---- an operand tile [D][64] (lanes = persons: 256-byte rows of a dimension-major array) into this wave's region by
    // DMA, as it stands: one transfer = four rows (lane: row lane / 16, persons 4 (lane % 16) .. + 3; D % 4 == 0, nb % 4 == 0).
    auto tile_dma = [&](const float* __restrict__ srcT) __attribute__((always_inline)) {
        constexpr int LPR = HB2_WP / 4;                                // lanes per row
        int64_t ig = i0 + 4 * (lane % LPR);
        if (ig + 4 > nb) ig = nb - 4;                                  // absent persons: valid ones, never stored
        const int r0 = lane / LPR;
        const uint32_t dst = lds_addr_uniform(gx_lds);
        for (int k = 0; k < D; k += TROWS) {
            const int kr = (k + r0 < D) ? k + r0 : D - 1;              // rows past D (tile padding): a harmless duplicate
            dma16(srcT + (int64_t)kr * nb + ig, dst + (uint32_t)k * (HB2_WP * 4));
        }
    };
    // The ring's first two batches and the gx tile are requested before anything else: they land while the eps fragments
    // are made.  (The powers of two that U_k carries are folded into gx when it is read: epilogue.)
    stage_next();
    stage_next();
    tile_dma(gxT);
    // ---- B fragments of both person sets: contraction index c = 16 s + 8 half + j as two fp16 terms of eps 2^e; e from
    // the largest magnitude among the wave's 64 persons
    const float w_inv = 1.0f / sc[2];                                  // 2^-sw (a power of two: exact)
    f16x8 bf[NSET][2][NS];
    float e_max = 0.f;
    {
        float v[NSET][NS][8];
        float m = 0.f;
#pragma unroll
        for (int u = 0; u < NSET; ++u)
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const float* er = eps_in + ic[u] * D;
                const int c0 = 16 * s + 8 * half;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    f32x4 t = {0.f, 0.f, 0.f, 0.f};
                    if (c0 + 4 * q + 4 <= D) t = *(const f32x4*)(er + c0 + 4 * q);      // D % 4 == 0 on this path
                    v[u][s][4 * q + 0] = t[0]; v[u][s][4 * q + 1] = t[1]; v[u][s][4 * q + 2] = t[2]; v[u][s][4 * q + 3] = t[3];
                }
            }
#pragma unroll
        for (int u = 0; u < NSET; ++u)
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(v[u][s][j]));
        e_max = wave_max_dpp(m);
        const float e_scale = ldexpf(1.0f, f16_scale_exp(e_max));
#pragma unroll
        for (int u = 0; u < NSET; ++u)
#pragma unroll
            for (int s = 0; s < NS; ++s) split2h_frag(v[u][s], e_scale, bf[u][0][s], bf[u][1][s]);
    }
    HSTAMP();                                                          // 1: eps fragments
    const float u_inv = w_inv * ldexpf(1.0f, -f16_scale_exp(e_max));   // takes 2^(sw + se) off U_k, folded into gx[p][k]
    if (lane == 0 && maxw) atomic_max_raise(maxw + 2, e_max);
    vx_wait_vmem();                                                    // the prologue's transfers are done: from here on
    HSTAMP();                                                          // 2: gx tile               vmcnt counts the ring alone

    // fragments of the operand tile in LDS ([D][64], this wave's region: gx, later gd), scaled by the power of two of its
    // largest magnitude (returned in vmax); returns the power of two that takes the scale (and the weights') off
    auto frags_from_tile = [&](float& vmax) __attribute__((always_inline)) -> float {
        float v[NSET][NS][8];
        float m = 0.f;
#pragma unroll
        for (int u = 0; u < NSET; ++u)
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c = 16 * s + 8 * half + j;
                    v[u][s][j] = (c < D) ? gx_lds[c * HB2_WP + 32 * u + l31] : 0.f;
                    m = fmaxf(m, fabsf(v[u][s][j]));
                }
        vmax = wave_max_dpp(m);
        const int se = f16_scale_exp(vmax);
        const float scl = ldexpf(1.0f, se);
#pragma unroll
        for (int u = 0; u < NSET; ++u)
#pragma unroll
            for (int s = 0; s < NS; ++s) split2h_frag(v[u][s], scl, bf[u][0][s], bf[u][1][s]);
        return w_inv * ldexpf(1.0f, -se);
    };

    f32x16 gh[NSET][2];
#pragma unroll
    for (int u = 0; u < NSET; ++u) { gh[u][0] = zero16(); gh[u][1] = zero16(); }
    f16x8 A[4];                                                        // fragments [ht * 2 + term] of the unit in flight
    int un = 0;                                                        // the unit whose fragments are being requested (uniform)
    int slot = 0;                                                      // its ring slot
    auto read_frag = [&](int f) __attribute__((always_inline)) {
        A[f] = *(const f16x8*)(ring + (size_t)slot * HB_UNIT_BYTES + f * 1024 + lane * 16);
    };
    // before the first fragment of batch b is read: its transfers have landed for every wave, and the slots of batch b - 1
    // -- whose last fragments every wave has in registers by now -- are free for batch b + 2
    auto advance = [&]() __attribute__((always_inline)) {
        ++un;
        slot = (slot + 1 == HB2_NSLOT) ? 0 : slot + 1;
        if ((un & (HB2_BATCH - 1)) == 0) {
            __builtin_amdgcn_s_waitcnt(0x0F70 | VM_OWN);               // vmcnt(own transfers of one batch): only the batch after it may be in flight
            __syncthreads();
            stage_next();
        }
    };
    // gh += gx_k U_k, element e = (set u, hidden tile ht, register r) of the row before the current one.  (asm: plain fmaf
    // calls are SLP-packed into v_pk_fma_f32, which wait for the matrix pipe beside MFMAs; volatile keeps the pieces where
    // they are written, between the MFMAs of the current row's first unit.)
    // The accumulator is read out of its AGPR by an explicit instruction at the place of use: left to itself the compiler
    // copies all 64 values to VGPRs in front of the row's first MFMA (so that the new row can take over the registers) --
    // 64 exposed reads a row.  Every piece sits at least four MFMAs (128 cycles) behind the last MFMA that wrote what it
    // reads, beyond the wait states the hardware needs between an MFMA and a read of its result.
    float gk[NSET];
    auto epi_piece = [&](const f32x16 (&U)[NSET][2], int e0, int e1) __attribute__((always_inline)) {
        if constexpr (NSET == 1) {                                     // VGPR-form accumulators: read in place
#pragma unroll
            for (int e = e0; e < e1; ++e) {
                const int ht = (e >> 4) & 1, r = e & 15;
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(gh[0][ht][r]) : "v"(gk[0]), "v"(U[0][ht][r]));
            }
        } else {
            float t[12];                                               // the reads of a piece first, then its FMAs (a read
#pragma unroll                                                         // followed by its own use costs a wait state each)
            for (int e = e0; e < e1; ++e) {
                const int u = e >> 5, ht = (e >> 4) & 1, r = e & 15;
                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t[e - e0]) : "a"(U[u][ht][r]));
            }
#pragma unroll
            for (int e = e0; e < e1; ++e) {
                const int u = e >> 5, ht = (e >> 4) & 1, r = e & 15;
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(gh[u][ht][r]) : "v"(gk[u]), "v"(t[e - e0]));
            }
        }
    };
    // one unit: 3 products per hidden tile and person set.  ONE fragment set: a fragment is re-read (for the next unit) right
    // after the last MFMA that takes it has been issued -- the heads first (eight MFMAs each), then the remainders -- so
    // every read has at least eight MFMAs (256 cycles) before its first use.  FIRST: the unit opens its row (the accumulators
    // start from zero).  EPI: which part of the epilogue of the row before (accumulators Up, row kp) goes between its MFMA
    // pairs: 0 none, 1 person set 0, 2 person set 1, 3 both (rows of one unit).
    auto unit = [&](auto sc_, auto firstc, auto epic, f32x16 (&U)[NSET][2], const f32x16 (&Up)[NSET][2], int kp) __attribute__((always_inline)) {
        constexpr int s = decltype(sc_)::value;
        constexpr bool FIRST = decltype(firstc)::value;
        constexpr int EPI = decltype(epic)::value;
        constexpr int E0 = (EPI == 2) ? 32 : 0, EN = (EPI == 3) ? 64 : (EPI ? 32 : 0);      // first element, count
        constexpr int P0 = E0, P1 = E0 + (EN * 1 + 5) / 6, P2 = E0 + (EN * 2 + 5) / 6, P3 = E0 + (EN * 3 + 5) / 6,
                      P4 = E0 + (EN * 4 + 5) / 6, P5 = E0 + (EN * 5 + 5) / 6, P6 = E0 + EN;
        if constexpr (EPI == 1 || EPI == 3) gk[0] = gx_lds[kp * HB2_WP + l31] * u_inv;
        if constexpr (NSET == 2 && (EPI == 2 || EPI == 3)) gk[NSET - 1] = gx_lds[kp * HB2_WP + 32 + l31] * u_inv;
#pragma unroll
        for (int u = 0; u < NSET; ++u) U[u][0] = mfma_f16(A[0], bf[u][1][s], FIRST ? zero16() : U[u][0]);
        if constexpr (EPI != 0) { __builtin_amdgcn_sched_barrier(0); epi_piece(Up, P0, P1); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int u = 0; u < NSET; ++u) U[u][0] = mfma_f16(A[0], bf[u][0][s], U[u][0]);
        advance();                                                     // from here on the reads are for the next unit
        read_frag(0);
        if constexpr (EPI != 0) { __builtin_amdgcn_sched_barrier(0); epi_piece(Up, P1, P2); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int u = 0; u < NSET; ++u) U[u][1] = mfma_f16(A[2], bf[u][1][s], FIRST ? zero16() : U[u][1]);
        if constexpr (EPI != 0) { __builtin_amdgcn_sched_barrier(0); epi_piece(Up, P2, P3); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int u = 0; u < NSET; ++u) U[u][1] = mfma_f16(A[2], bf[u][0][s], U[u][1]);
        read_frag(2);
        if constexpr (EPI != 0) { __builtin_amdgcn_sched_barrier(0); epi_piece(Up, P3, P4); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int u = 0; u < NSET; ++u) U[u][0] = mfma_f16(A[1], bf[u][0][s], U[u][0]);
        read_frag(1);
        if constexpr (EPI != 0) { __builtin_amdgcn_sched_barrier(0); epi_piece(Up, P4, P5); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int u = 0; u < NSET; ++u) U[u][1] = mfma_f16(A[3], bf[u][0][s], U[u][1]);
        read_frag(3);
        if constexpr (EPI != 0) { __builtin_amdgcn_sched_barrier(0); epi_piece(Up, P5, P6); __builtin_amdgcn_sched_barrier(0); }
    };
    auto epilogue = [&](int k, const f32x16 (&U)[NSET][2]) __attribute__((always_inline)) {     // not overlapped: block ends
#pragma unroll
        for (int u = 0; u < NSET; ++u) {                               // (plain code: the compiler places the MFMA -> read wait)
            const float g = gx_lds[k * HB2_WP + 32 * u + l31] * u_inv;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                gh[u][0][r] = fmaf(g, U[u][0][r], gh[u][0][r]);
                gh[u][1][r] = fmaf(g, U[u][1][r], gh[u][1][r]);
            }
        }
    };
    constexpr std::true_type T_{};
    constexpr std::false_type F_{};
    constexpr std::integral_constant<int, 0> E_none{};

    // unit 0
    __syncthreads();                                                   // batches 0 and 1 have landed for every wave
    stage_next();
#pragma unroll
    for (int f = 0; f < 4; ++f) read_frag(f);
    HSTAMP();                                                          // 3: first batch landed

    // ---- OFF rows: k in blocks of 16; block kb has kb + 1 units per k.  The rows alternate between two accumulator sets
    // with FIXED roles in the loop body (no copies): the first row of a block goes to Ua without an epilogue, then pairs
    // (Ub with the epilogue of Ua, Ua with the epilogue of Ub), and the block ends with the epilogue of its last row.
    f32x16 Ua[NSET][2], Ub[NSET][2];
    static_for<NS>([&](auto kbc) {
        constexpr int kb = decltype(kbc)::value;
        const int k_lo = 16 * kb + 1, k_hi = (16 * kb + 16 < D - 1) ? 16 * kb + 16 : D - 1;
        // a row with the epilogue of the row before it: person set 0 beside the first unit, set 1 beside the second (rows of
        // one unit: both beside it)
        auto row = [&](int k, auto epic, f32x16 (&Uc)[NSET][2], const f32x16 (&Up)[NSET][2]) __attribute__((always_inline)) {
            constexpr bool EP = decltype(epic)::value;
            static_for<kb + 1>([&](auto sc_) {
                constexpr int s = decltype(sc_)::value;
                constexpr int part = !EP ? 0 : NSET == 1 ? (s == 0 ? 1 : 0) : (kb == 0 ? 3 : (s == 0 ? 1 : (s == 1 ? 2 : 0)));
                if constexpr (s == 0) unit(sc_, T_, std::integral_constant<int, part>{}, Uc, Up, k - 1);
                else unit(sc_, F_, std::integral_constant<int, part>{}, Uc, Up, k - 1);
            });
        };
        if (k_lo <= k_hi) {
            row(k_lo, F_, Ua, Ub);
            int k = k_lo + 1;
            for (; k + 1 <= k_hi; k += 2) {
                row(k, T_, Ub, Ua);
                row(k + 1, T_, Ua, Ub);
            }
            if (k <= k_hi) {                                           // an even number of rows in the block: one more, into Ub
                row(k, T_, Ub, Ua);
                epilogue(k, Ub);
            } else {
                epilogue(k - 1, Ua);
            }
        }
    });
    HSTAMP();                                                          // 4: OFF rows
    // ---- LOC rows (operand gx: the tile in LDS) and DIAG rows (operand gd): an accumulator set of their own, added with
    // their power of two.  The ring keeps streaming (in the image the LOC units follow the OFF units, the DIAG units
    // come last).  The gd tile takes the place of the gx tile by DMA as soon as the LOC fragments have been made, and lands
    // beside the MFMAs of the LOC units.
    float d_max = 0.f;
    auto section = [&](float cinv) __attribute__((always_inline)) {
        static_for<NS>([&](auto sc_) {
            constexpr int s = decltype(sc_)::value;
            if (s < ns) { if constexpr (s == 0) unit(sc_, T_, E_none, Ua, Ub, 0); else unit(sc_, F_, E_none, Ua, Ub, 0); }
        });
#pragma unroll
        for (int u = 0; u < NSET; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                gh[u][0][r] = fmaf(cinv, Ua[u][0][r], gh[u][0][r]);
                gh[u][1][r] = fmaf(cinv, Ua[u][1][r], gh[u][1][r]);
            }
    };
    float g_max = 0.f;
    {
        const float cinv = frags_from_tile(g_max);
        __builtin_amdgcn_s_waitcnt(0xC07F);                            // lgkmcnt(0): the tile has been read ...
        __builtin_amdgcn_wave_barrier();
        tile_dma(gdT);                                                 // ... and is replaced by the gd tile
        section(cinv);
    }
    {
        vx_wait_vmem();                                                // the gd tile (and every ring transfer) has landed
        section(frags_from_tile(d_max));
    }
    vx_wait_vmem();                                                    // no DMA may be in flight when the LDS is released
    HSTAMP();                                                          // 5: sections
    if (lane == 0 && maxw) {
        atomic_max_raise(maxw + 0, g_max);
        atomic_max_raise(maxw + 1, d_max);
    }

    // ---- ghpre = gh * softplus'(pre) = gh * (1 - exp(-h));  C layout: rows hh = crow32(r, half), cols p
    float p_max = 0.f;
#pragma unroll
    for (int u = 0; u < NSET; ++u) {
        const int64_t i = iu[u];
        if (ghpreT_out) {                                              // dimension-major: 128-byte rows per half-wave
            if (i < nb) {
                // h from the person-major copy (eight 16-byte loads a person set; hT would be 64 strided 4-byte loads)
                f32x4 hv[2][4];
#pragma unroll
                for (int ht = 0; ht < 2; ++ht)
#pragma unroll
                    for (int g = 0; g < 4; ++g) hv[ht][g] = *(const f32x4*)(h_in + i * H + 32 * ht + 8 * g + 4 * half);
#pragma unroll
                for (int ht = 0; ht < 2; ++ht)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int64_t o = (int64_t)(32 * ht + crow32(r, half)) * nb + i;
                        const float gp = gh[u][ht][r] * (1.0f - __expf(-hv[ht][r >> 2][r & 3]));
                        p_max = fmaxf(p_max, fabsf(gp));
                        ghpreT_out[o] = gp;
                    }
            }
        } else if (i < nb) {
#pragma unroll
            for (int ht = 0; ht < 2; ++ht)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int hh0 = 32 * ht + 8 * g + 4 * half;
                    const float4 hv = *(const float4*)(h_in + i * H + hh0);
                    float4 o;
                    o.x = gh[u][ht][4 * g + 0] * (1.0f - __expf(-hv.x));
                    o.y = gh[u][ht][4 * g + 1] * (1.0f - __expf(-hv.y));
                    o.z = gh[u][ht][4 * g + 2] * (1.0f - __expf(-hv.z));
                    o.w = gh[u][ht][4 * g + 3] * (1.0f - __expf(-hv.w));
                    *(float4*)(ghpre_out + i * H + hh0) = o;
                }
        }
    }
    if (ghpreT_out) {
        p_max = wave_max_dpp(p_max);
        if (lane == 0 && maxw) atomic_max_raise(maxw + 3, p_max);
    }
#ifdef HB2_STAMPS
    HSTAMP();                                                          // 6: output
    if ((blockIdx.x == 100 || blockIdx.x == 2000) && lane == 0 && (wave == 0 || wave == 3))
        printf("HB2 STAMPS blk %d wave %d: eps %llu gx %llu first %llu off %llu sec %llu out %llu total %llu\n", (int)blockIdx.x, wave,
               st_[1] - st_[0], st_[2] - st_[1], st_[3] - st_[2], st_[4] - st_[3], st_[5] - st_[4], st_[6] - st_[5], st_[6] - st_[0]);
#endif
}
