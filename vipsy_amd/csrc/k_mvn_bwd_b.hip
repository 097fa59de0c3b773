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
//
// Round 5 ("without its per-tile barrier").  Per-wave private tiles -- every wave staging its own G / E
// rows AND h through a ring of its own -- put h through the L2 -> LDS path sixteen times a tile (8 KB x 16 waves x 31 250
// tiles x 11 slabs = 44 GB a launch on top of the 5 GB now, the traffic docs/HARDWARE.md rule 30 measured as binding in the
// forward), so the structure built instead keeps the shared tile and removes what the barrier was thought to wait for:
//   * NBUF = 3: a COMPACT tile image (only the transfers a slab needs, in rank order -- 14-25 KB of the map's 50 KB for all
//     slabs but the last; the per-lane LDS addresses are computed once, so the re-mapping costs the prologue only) makes three
//     52 KB buffers fit, the transfers of tile s + 3 go out behind the barrier of tile s -- two tile periods to land instead
//     of one -- and the wait in front of a barrier is counted so that the youngest tile's transfers stay in flight;
//   * scalar-lean staging: running scalar source pointers (two adds a tile instead of a 64-bit shift and add per transfer),
//     precomputed LDS destinations, 32-bit tile indices, the ragged last tile peeled off: ~45 scalar instructions a tile and
//     wave instead of ~75;
//   * no register spills (116 registers with two buffers, 128 with three: the remainders of h single-buffered, 32-bit lane
//     offsets against scalar bases instead of 64-bit per-lane pointers).
// Every variant produces the round-4 kernel's slab words bit for bit (tools/bwb_bench.hip: one checksum over all of them, at
// 1 000 000 / 999 968 / 4 008 / 96 persons), and none is faster: 2.60-2.70 ms (two buffers, lean staging), 2.68-2.78 (three
// buffers), 2.63-2.71 (round 4's file) in alternating runs on one box.  The kernel waits neither for its transfers' latency
// nor for the scalar unit; the library ships NBUF = 2 (the cleaner code: no spills), NBUF = 3 stays a template argument for the
// harness.
#ifndef BWB_WAVES
#define BWB_WAVES 16
#endif
#define BWB_RT (16 / BWB_WAVES)
#define BWB_THREADS (64 * BWB_WAVES)
#define BB_BUF3 53248                                                   // 52 KB: a compact image at D <= 112 (14 G + 14 E + 8 H + 14 GD + 1 C units)
__host__ __device__ inline bool bb_three(int D) { return bt_dr(D) <= 112; }
__host__ __device__ inline size_t bb_lds_bytes_n(int D) { return bb_three(D) ? 3 * (size_t)BB_BUF3 : bb_lds_bytes(D); }
// transfer (1 KB = the row units r0 .. r0 + 3 and r0 + 8 .. r0 + 11) that holds row unit R of the tile map, and R's offset in it
__host__ __device__ inline int bb_xfer_of(int R) { return 2 * (R >> 4) + ((R >> 2) & 1); }

template <int NBUF>
__global__ __launch_bounds__(BWB_THREADS, 1) void k_mvn_enc_bwd_w_b(
    EncDims dm, const uint16_t* __restrict__ hs /*[2][64][nb] fp16 terms of h 2^sh*/, const float* __restrict__ epsT,
    const float* __restrict__ gdT, const float* __restrict__ gxT, const uint32_t* __restrict__ gtab,
    const float* __restrict__ sc /*k_enc_scales*/, const uint32_t* __restrict__ maxw /*float bits: max |gx|, |gd|, |eps|*/,
    float* __restrict__ slabs, int64_t slab_len) {
    extern __shared__ __attribute__((aligned(16))) char smem_bb[];
    static_assert(NBUF == 2 || NBUF == 3, "two full-size buffers or three compact ones");
    const int D = dm.D;
    const int64_t nb = dm.nb;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    constexpr uint32_t BUF = NBUF == 3 ? BB_BUF3 : BB_BUF;
    const int Rp = pk_rows(D);
    int rb_, blk_pr;
    bt_decode(vgrid_launch(), rb_, blk_pr);
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
    __shared__ uint32_t need_lo, need_hi;                              // bit d: transfer d is staged by this workgroup
    if (tid < 16) row_used[tid] = 0u;
    __syncthreads();
    // ---- the operand rows of this lane's packed rows
    int gRow[BWB_RT], eRow[BWB_RT];
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
        gRow[t] = g; eRow[t] = e;
        // which row units of the tile this workgroup's 512 rows read at all (see "only what the slab reads" below)
        if (half == 0) { atomicOr(&row_used[g >> 5], 1u << (g & 31)); atomicOr(&row_used[e >> 5], 1u << (e & 31)); }
    }
    // DMA transfers of 1 KB (see k_mvn_bwd_t.hip): d < rH / 8: G / E rows; the next 8: the H area (4 per term); then GD rows.
    const int dH0 = rH / 8, dGD0 = rGD / 8;
    const int n_dma = need_gd ? rOnes / 8 : dGD0;                      // <= 56
    // Only what the slab reads is staged.  A slab of 512 packed rows reads the G rows of its own k (a dozen of the D), the E
    // rows below its largest k, h, and GD only in the slab with the diagonal rows: 14-25 KB a tile of 32 persons instead of the
    // whole map's 50 KB for every slab (17.7 GB a launch through the L2 -> LDS path before).  A transfer (8 row units: 4 and
    // the 4 eight further on) is issued if one of its units is in row_used.
    __syncthreads();                                                   // row_used is complete
    if (wave == 0) {
        const int d = lane;
        bool need = false;
        if (d >= dH0 && d < dGD0) need = true;
        else if (d < n_dma) {
            const int r0 = 16 * (d >> 1) + 4 * (d & 1);                // units r0 .. r0 + 3 and r0 + 8 .. r0 + 11
            const uint64_t w2 = ((uint64_t)row_used[(r0 >> 5) + 1] << 32) | row_used[r0 >> 5];
            need = ((w2 >> (r0 & 31)) & 0x0F0Full) != 0;
        }
        const uint64_t m = __ballot(need);
        if (lane == 0) { need_lo = (uint32_t)m; need_hi = (uint32_t)(m >> 32); }
    }
    __syncthreads();
    const uint64_t need_mask = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)need_hi) << 32) |
                               (uint32_t)__builtin_amdgcn_readfirstlane((int)need_lo);
    const int n_need = __popcll(need_mask);
    // LDS byte offset (inside a buffer) of chunk c of row unit R: the full map's place, or -- three buffers -- the place in the
    // compact image: needed transfers in order, the constant rows (ones, zeros) in the unit behind them
    auto caddr = [&](int R, int c) -> uint32_t {
        if constexpr (NBUF == 2) return bt_addr(R, c);
        const int d = bb_xfer_of(R);
        const uint32_t rank = d >= n_dma ? (uint32_t)n_need : (uint32_t)__popcll(need_mask & ((1ull << d) - 1ull));
        return rank * 1024u + (bt_addr(R, c) - (uint32_t)d * 1024u);
    };
    // ---- per-lane LDS addresses: fp32 rows, persons 16 c + 8 half + 0..7 = chunks 4c + 2 half and + 1
    uint32_t aG[BWB_RT][2], aE[BWB_RT][2], aHf[2][2][2];                 // [..][chunk c]; H: [term][hidden tile][chunk]
#pragma unroll
    for (int t = 0; t < BWB_RT; ++t)
#pragma unroll
        for (int c = 0; c < 2; ++c) { aG[t][c] = caddr(gRow[t], 4 * c + 2 * half); aE[t][c] = caddr(eRow[t], 4 * c + 2 * half); }
    // second 16-byte piece of a row's pair: chunk index + 1 = slot XOR 1 (4c + 2 half is even)
    {
        const uint32_t hbase = NBUF == 2 ? (uint32_t)rH * 128u : (uint32_t)__popcll(need_mask & ((1ull << dH0) - 1ull)) * 1024u;
#pragma unroll
        for (int s3 = 0; s3 < 2; ++s3)
#pragma unroll
            for (int ht = 0; ht < 2; ++ht)
#pragma unroll
                for (int c = 0; c < 2; ++c) aHf[s3][ht][c] = hbase + bb_haddr(s3, 32 * ht + l31, 2 * c + half);
    }
    for (int b = 0; b < NBUF; ++b) {
        // the "ones" row (and the row of zeros behind it)
        if (tid < 64) *(float*)(smem_bb + b * BUF + caddr(rOnes + (tid >> 5), (tid & 31) >> 2) + 4 * (tid & 3)) = (tid < 32) ? 1.0f : 0.f;
    }
    f32x16 acc[BWB_RT][2];
    float bsum[BWB_RT];
#pragma unroll
    for (int t = 0; t < BWB_RT; ++t) { bsum[t] = 0.f; acc[t][0] = zero16(); acc[t][1] = zero16(); }

    const int n_ptiles = (int)((nb + BT_P - 1) / BT_P);               // (nb < 2^23: tile indices are 32-bit scalars)
    constexpr int BB_MAXD = (56 + BWB_WAVES - 1) / BWB_WAVES;
    // source of transfer u: a wave-uniform region base (the array the transfer's rows belong to) + this lane's 32-bit byte
    // offset at person 0 (row and 16-byte piece: fixed for the whole launch; nb < 2^23 keeps it under 2^32) + the tile's person
    // offset, which is uniform and goes into the scalar base: global_load_lds takes (scalar base, 32-bit lane offset), so a
    // transfer costs no vector address arithmetic and its lane offset ONE register (64-bit per-lane pointers: two, and the
    // three-buffer form then spilled them inside the loop -- a scratch reload before every transfer is a vmcnt(0) in front of
    // transfers that are meant to stay in flight)
    const char* sbase[BB_MAXD];                                        // wave-uniform
    uint32_t voff[BB_MAXD];
    uint32_t vsh[BB_MAXD];                                             // person index -> bytes: fp16 planes 1, fp32 rows 2
    uint32_t ldst[BB_MAXD];                                            // wave-uniform: the transfer's place in a buffer
    uint32_t need_u = 0u;                                              // bit u: transfer wave + BWB_WAVES u is issued
#pragma unroll
    for (int u = 0; u < BB_MAXD; ++u) {
        const int d = wave + BWB_WAVES * u;
        if (d < n_dma && ((need_mask >> d) & 1ull)) need_u |= 1u << u;
        ldst[u] = NBUF == 2 ? (uint32_t)d * 1024u : (uint32_t)__popcll(need_mask & ((1ull << (d < 63 ? d : 63)) - 1ull)) * 1024u;
        if (d >= dH0 && d < dGD0) {                                    // H area: 16 hidden rows of one term
            const int dd = d - dH0, s3 = dd >> 2;
            const int hh = 16 * (dd & 3) + 4 * (lane >> 4) + ((lane & 15) >> 2);
            const int s = (lane & 3) ^ ((hh >> 2) & 3);
#ifdef BWB_TILE_MAJOR                                                  // (harness experiment: every operand in 32-person blocks)
            voff[u] = (uint32_t)((((int64_t)s3 * 64 + hh) * 32 + 8 * s) * 2);
#else
            voff[u] = (uint32_t)((((int64_t)s3 * 64 + hh) * nb + 8 * s) * 2);
#endif
        } else {
            const int i = 4 * (d & 1) + (lane >> 4), beta = (lane >> 3) & 1;
            const int R = 16 * (d >> 1) + 8 * beta + i;
            const int c = (lane & 7) ^ i;
            int rl = R >= rGD ? R - rGD : R >= rE ? R - rE : R;
            if (rl >= D) rl = D - 1;                                   // padding rows of a region: a harmless duplicate
#ifdef BWB_TILE_MAJOR
            voff[u] = (uint32_t)(((int64_t)rl * 32 + 4 * c) * 4);
#else
            voff[u] = (uint32_t)(((int64_t)rl * nb + 4 * c) * 4);
#endif
        }
        const bool isH = d >= dH0 && d < dGD0;
        sbase[u] = d >= dGD0 ? (const char*)gdT : isH ? (const char*)hs : d >= rE / 8 ? (const char*)epsT : (const char*)gxT;
        vsh[u] = isH ? 1u : 2u;
    }
    need_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)need_u);
    const int n_own = __popc(need_u);                                  // transfers this wave issues for a whole tile (<= BB_MAXD)
    const uint32_t img_bytes = NBUF == 2 ? (uint32_t)bb_rows(D) * 128u : (uint32_t)n_need * 1024u;   // what the transfers fill
    // Staging is called once per tile of this workgroup, in order (t0, t0 + GS, ...), so the scalar source of a transfer is a
    // RUNNING pointer: two scalar adds a tile instead of a 64-bit shift and add from the tile index, and its LDS destination
    // (buffer and place) a precomputed scalar per buffer.  The fast path -- a whole tile, every tile but possibly the last --
    // is then seven scalar instructions a transfer; the ragged last tile (nb % 32 != 0) takes the slow path with its clearing
    // pass and per-lane person checks.  (Round 5: the scalar unit issues one instruction per SIMD turn like every other
    // unit; ~75 scalar instructions a tile and wave of staging and 64-bit loop control were ~1 200 of a tile's ~3 400 cycles.)
    const int GS = (int)gridDim.y;
    uint64_t sptr[BB_MAXD];                                            // wave-uniform: source of transfer u for the next tile to stage
    uint32_t sdel[BB_MAXD], lm0[NBUF][BB_MAXD];
    {
        const uint32_t lb0 = lds_addr_uniform(smem_bb);
#pragma unroll
        for (int u = 0; u < BB_MAXD; ++u) {
#ifdef BWB_TILE_MAJOR
            const uint32_t tstride = vsh[u] == 1u ? 2u * 64u * 32u * 2u : (uint32_t)D * 128u;
            sptr[u] = (uint64_t)sbase[u] + (uint64_t)blk_pr * tstride;
            sdel[u] = (uint32_t)GS * tstride;
#else
            sptr[u] = (uint64_t)sbase[u] + ((uint64_t)((int64_t)blk_pr * BT_P) << vsh[u]);
            sdel[u] = (uint32_t)(GS * BT_P) << vsh[u];
#endif
#pragma unroll
            for (int b2 = 0; b2 < NBUF; ++b2) lm0[b2][u] = lb0 + (uint32_t)b2 * BUF + ldst[u];
        }
    }
    const bool ragged_end = (nb % BT_P) != 0;
    auto stage = [&](int tile, auto bc) __attribute__((always_inline)) {
        constexpr int b = decltype(bc)::value;
        if (ragged_end && tile == n_ptiles - 1) {                      // uniform, once a launch at most
            const int64_t i0 = (int64_t)tile * BT_P;
            const int pv = (int)(nb - i0);
            if constexpr (NBUF == 2) {
                for (int e = tid; e < bb_rows(D) * 32; e += BWB_THREADS) {
                    const int row = e >> 5;
                    if (row != rOnes) ((float*)(smem_bb + b * BUF))[e] = 0.f;
                }
            } else {
                for (uint32_t e = tid; e < img_bytes / 4; e += BWB_THREADS) ((float*)(smem_bb + b * BUF))[e] = 0.f;   // (the C unit lies behind)
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < BB_MAXD; ++u) {
                const int d = wave + BWB_WAVES * u;
                if ((need_u >> u) & 1u) {
                    const bool isH = d >= dH0 && d < dGD0;
                    int p0, ln = lane;
                    asm volatile("" : "+v"(ln));                       // (computed HERE, on the one ragged tile: hoisted out of the
                    if (isH) {                                         // loop it holds a register per transfer)
                        const int hh = 16 * ((d - dH0) & 3) + 4 * (ln >> 4) + ((ln & 15) >> 2);
                        p0 = 8 * ((ln & 3) ^ ((hh >> 2) & 3));         // persons of this lane's 16 bytes: 8 (H) or 4 (fp32)
                    } else {
                        p0 = 4 * ((ln & 7) ^ (4 * (d & 1) + (ln >> 4)));
                    }
                    if (p0 < pv) dma16s((const void*)sptr[u], voff[u], lm0[b][u]);
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < BB_MAXD; ++u)
                if ((need_u >> u) & 1u) dma16s((const void*)sptr[u], voff[u], lm0[b][u]);
        }
#pragma unroll
        for (int u = 0; u < BB_MAXD; ++u) sptr[u] += sdel[u];
    };
    // this wave's own transfers of the tile after next stay in flight (three buffers); vmcnt takes an immediate
    auto wait_leaving = [&](int n) __attribute__((always_inline)) {
        switch (n) {
            case 1: __builtin_amdgcn_s_waitcnt(0x0F70 | 1); break;
            case 2: __builtin_amdgcn_s_waitcnt(0x0F70 | 2); break;
            case 3: __builtin_amdgcn_s_waitcnt(0x0F70 | 3); break;
            case 4: __builtin_amdgcn_s_waitcnt(0x0F70 | 4); break;
            default: vx_wait_vmem(); break;
        }
    };
    static_assert(BB_MAXD <= 4, "wait_leaving covers four transfers a wave");

    // ---- compute: 2 chunks x BWB_RT row tiles = 2 BWB_RT groups per tile and wave, 6 MFMAs per group.  The fragments of group g + 1
    // (4 LDS reads, 8 products, their scaling, 4 x (split of an element pair), the bias sum) are made in the shadow of the
    // MFMAs of group g, a few vector instructions after each MFMA; every slice is a pinned scheduling region.  The barrier
    // of a tile sits before its LAST group, whose operands are in registers already: behind it the buffer is free for the
    // DMA of a later tile, and the prefetch of that group reads the next tile from the next buffer.
    // Vector instructions in the shadow of an MFMA (tools/slice_ubench.hip): about five single-pass ones are free;
    // v_pk_*_f32 are NOT (they wait for the matrix pipe).
    // Every instruction of a slice is a volatile asm statement: the optimizer otherwise re-vectorizes the scalar
    // arithmetic into v_pk_*_f32 across slices and sinks whole slices out of the MFMA shadow.  A slice alternates pieces
    // of two element pairs: a vector instruction that depends on the one issued just before it waits for it (measured
    // 1.66 x the issue time).
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    uint32_t fq[2][2][4];                                              // [parity of the group][hi, lo][element pair]
    // h fragments [hidden tile]: the heads (term 0) feed the first two and the last two MFMAs of a group, so the next chunk's
    // are read into a second set while this chunk's are in use; the remainders (term 1) feed MFMAs three and four only and
    // are read for the next chunk right behind the fourth -- one set: eight registers less than two sets of both, which is
    // what keeps the three-buffer form inside 128 registers (a reload from scratch in front of a transfer would be a
    // vmcnt(0) in front of transfers that are meant to stay in flight)
    f16x8 h0f[2][2], h1f[2];                                           // [parity of the chunk][hidden tile]; [hidden tile]
    f32x4 rg0, rg1, re0, re1;
    float pv_[8], s0, s1, s2, s3;
    auto amul = [](float x, float y) -> float { float d; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y)); return d; };
    auto aadd = [](float x, float y) -> float { float d; asm volatile("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y)); return d; };
    auto read_raw = [&](const char* rb, uint32_t ag, uint32_t ae) __attribute__((always_inline)) {
        rg0 = *(const f32x4*)(rb + ag); rg1 = *(const f32x4*)(rb + (ag ^ 16u));
        re0 = *(const f32x4*)(rb + ae); re1 = *(const f32x4*)(rb + (ae ^ 16u));
    };
    auto products0 = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) pv_[j] = amul(rg0[j], re0[j]);
        s0 = aadd(pv_[0], pv_[1]); s1 = aadd(pv_[2], pv_[3]);
    };
    auto products1 = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) pv_[4 + j] = amul(rg1[j], re1[j]);
        s2 = aadd(pv_[4], pv_[5]); s3 = aadd(pv_[6], pv_[7]);
    };
    // The split of an element pair V 2^sv -> two fp16 terms in FOUR instructions (round 4; eight before: two scalings, head
    // conversion, two conversions back, two subtractions, remainder conversion).  v_fma_mixlo_f16 / v_fma_mixhi_f16 compute an
    // fp32 fma of operands that are fp32 or one half of a register read as fp16, and write the result, rounded to fp16, into one
    // half of the destination: head = rn16(v * 2^sv + 0), remainder = rn16(v * 2^sv - head).  Both fmas are exact in fp32 (a
    // power-of-two scaling; the difference of a value and its own fp16 rounding), so each result is rounded once, to fp16,
    // exactly as in the longer sequence: the same bits, except that a product of -0 splits into (+0, -0) instead of (-0, +0)
    // (tools/bwb_bench.hip: the checksum of every slab word is the same).
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
    // one tile from buffer b.  has_next: a next tile exists (its first group is prefetched from the next buffer in the last
    // slice); stage_tile: the tile whose transfers go into THIS buffer behind the barrier (-1: none); leave: how many of this
    // wave's transfers may stay in flight at the barrier (three buffers: those of the tile after next, when it is a whole one)
    auto tile_body = [&](auto bc, bool has_next, int stage_tile, int leave) {
        constexpr int b = decltype(bc)::value, bn = (b + 1) % NBUF;
        static_for<2 * BWB_RT>([&](auto gc) {
            constexpr int gi = decltype(gc)::value, c = gi / BWB_RT, t = gi % BWB_RT, cur = gi & 1, nxt = cur ^ 1;
            constexpr int gn = (gi + 1) % (2 * BWB_RT), cn = gn / BWB_RT, tn = gn % BWB_RT;
            constexpr bool last = gi == 2 * BWB_RT - 1;
            if constexpr (last) {
                wait_leaving(leave);
                __syncthreads();                                       // next tile landed; this tile's buffer is free
                if (stage_tile >= 0) stage(stage_tile, bc);
            }
            const char* rb = smem_bb + (last ? bn : b) * BUF;
            constexpr std::integral_constant<int, cur> curc{};
            constexpr std::integral_constant<int, nxt> nxtc{};
            constexpr std::integral_constant<int, 0> I0{};
            constexpr std::integral_constant<int, 1> I1{};
            constexpr std::integral_constant<int, 2> I2{};
            constexpr std::integral_constant<int, 3> I3{};
            const f16x8 vh = frag(curc, I0), vl = frag(curc, I1);
            __builtin_amdgcn_sched_barrier(0);
            acc[t][0] = mfma_f16(vl, h0f[c][0], acc[t][0]);
            __builtin_amdgcn_sched_barrier(0);
            read_raw(rb, aG[tn][cn], aE[tn][cn]);
            if constexpr (t == BWB_RT - 1) {
#pragma unroll
                for (int ht = 0; ht < 2; ++ht) h0f[cn][ht] = *(const f16x8*)(rb + aHf[0][ht][cn]);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[t][1] = mfma_f16(vl, h0f[c][1], acc[t][1]);
            __builtin_amdgcn_sched_barrier(0);
            products0();
            __builtin_amdgcn_sched_barrier(0);
            acc[t][0] = mfma_f16(vh, h1f[0], acc[t][0]);
            __builtin_amdgcn_sched_barrier(0);
            products1();
            __builtin_amdgcn_sched_barrier(0);
            acc[t][1] = mfma_f16(vh, h1f[1], acc[t][1]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (t == BWB_RT - 1) {                           // the remainders of the next chunk: their registers are free now
#pragma unroll
                for (int ht = 0; ht < 2; ++ht) h1f[ht] = *(const f16x8*)(rb + aHf[1][ht][cn]);
            }
            H1(I0, nxtc); H1(I1, nxtc); s0 = aadd(s0, s1); H2(I0, nxtc); H2(I1, nxtc); H1(I2, nxtc); H1(I3, nxtc);
            s2 = aadd(s2, s3); H2(I2, nxtc); H2(I3, nxtc); s0 = aadd(s0, s2);
            __builtin_amdgcn_sched_barrier(0);
            acc[t][0] = mfma_f16(vh, h0f[c][0], acc[t][0]);
            __builtin_amdgcn_sched_barrier(0);
            L1(I0, nxtc); L1(I1, nxtc); L2(I0, nxtc); L2(I1, nxtc);
            if constexpr (last) bsum[tn] = aadd(bsum[tn], has_next ? s0 : 0.f); else bsum[tn] = aadd(bsum[tn], s0);
            L1(I2, nxtc); L1(I3, nxtc);
            __builtin_amdgcn_sched_barrier(0);
            acc[t][1] = mfma_f16(vh, h0f[c][1], acc[t][1]);
            __builtin_amdgcn_sched_barrier(0);
            L2(I2, nxtc); L2(I3, nxtc);
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    const int t0 = blk_pr;
    if (t0 < n_ptiles) {
        // the first NBUF tiles before anything else; the first barrier publishes all of them
        stage(t0, std::integral_constant<int, 0>{});
        if (t0 + GS < n_ptiles) stage(t0 + GS, std::integral_constant<int, 1>{});
        if constexpr (NBUF == 3) { if (t0 + 2 * GS < n_ptiles) stage(t0 + 2 * GS, std::integral_constant<int, 2>{}); }
        vx_wait_vmem();
        __syncthreads();
        // fragments of the first group, outside the pipeline
        read_raw(smem_bb, aG[0][0], aE[0][0]);
#pragma unroll
        for (int ht = 0; ht < 2; ++ht) { h0f[0][ht] = *(const f16x8*)(smem_bb + aHf[0][ht][0]); h1f[ht] = *(const f16x8*)(smem_bb + aHf[1][ht][0]); }
        products0();
        products1();
        {
            constexpr std::integral_constant<int, 0> z{};
            constexpr std::integral_constant<int, 1> o1{};
            constexpr std::integral_constant<int, 2> o2{};
            constexpr std::integral_constant<int, 3> o3{};
            H1(z, z); H2(z, z); L1(z, z); L2(z, z);
            H1(o1, z); H2(o1, z); L1(o1, z); L2(o1, z);
            H1(o2, z); H2(o2, z); L1(o2, z); L2(o2, z);
            H1(o3, z); H2(o3, z); L1(o3, z); L2(o3, z);
        }
        bsum[0] += (s0 + s1) + (s2 + s3);
        int tile = t0;
        // tile s runs from buffer s % NBUF; behind its barrier the transfers of tile s + NBUF go into that buffer; at the barrier
        // the transfers of tile s + 1 must have landed and (three buffers) those of tile s + 2 -- a whole tile -- may still fly
        auto step = [&](auto bc) __attribute__((always_inline)) {
            const int tn = tile + NBUF * GS;
            int leave = 0;
            if constexpr (NBUF == 3) {
                const int t2 = tile + 2 * GS;
                if (t2 < n_ptiles && !(ragged_end && t2 == n_ptiles - 1)) leave = n_own;
            }
            tile_body(bc, tile + GS < n_ptiles, tn < n_ptiles ? tn : -1, leave);
            tile += GS;
        };
        while (tile < n_ptiles) {
            step(std::integral_constant<int, 0>{});
            if (tile < n_ptiles) step(std::integral_constant<int, 1>{});
            if constexpr (NBUF == 3) { if (tile < n_ptiles) step(std::integral_constant<int, 2>{}); }
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
        const float bt = half_sum32(bsum[t]);                           // (2^-sv: taken off by V's own scaling -- the sums are of unscaled V)
        const int64_t row = rbase + 32 * t + l31;
        if (half == 0 && row < Rp) slab[(int64_t)Rp * 64 + row] = bt;
    }
}
