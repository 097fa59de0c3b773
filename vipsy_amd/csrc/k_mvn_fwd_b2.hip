// Amortized MVN guide forward, SIXTY-FOUR persons per wave: the same mathematics and outputs as k_mvn_enc_fwd_b
// (k_mvn_fwd_b.hip), for batches large enough to fill the chip with 256-person workgroups.
//
// What bounds k_mvn_enc_fwd_b is the instruction budget of a wave per head tile: 13 fragment loads, the group-table read,
// four eps reads and ~120 vector instructions beside 25 MFMAs (measured: a second wave per SIMD buys nothing, the loads
// alone cost 0.8 ms of 4.2).  Here every fragment of a head tile feeds TWO 32-person MFMA chains (person sets 0 and 1 of
// the wave), so loads, table reads and k-end branches are spent once per 50 MFMAs; fc1 shares its W1 fragments the same
// way.  Two person sets do not fit the LDS with an x tile each (832 B a person), so x is not kept in LDS at all: the OFF
// rows' sums go to global memory as each k ends (write only), and the DIAG / LOC tiles of a 32-row block are taken
// TOGETHER at the end: x = x_off (read back once) + exp(M_kk) eps_k + loc_k.  The staging area is the wave's own
// 64 x D region of eps_out, used as [D][64] so that a k-end is one full 128-byte line per person set (4-byte stores
// into the x rows left partly written lines to the L2: +1.4 GB of HBM writes a step); eps_out itself is written last,
// from the LDS tile.  A wave with fewer than 64 persons stages in its x rows instead.
// (included by vx_abi.hip after k_mvn_fwd_b.hip, whose images, tables and helpers it uses)

#define FB2_THREADS 256
#define FB2_WAVES 4
// NS person sets of 32 per wave.  NS = 2 (64 persons, one workgroup a CU: 135 KB of LDS at the headline shape) is the form
// described above.  NS = 1 (round 3) is the same code with one set: 66 KB of LDS and under 256 registers, so TWO workgroups
// share a CU -- two waves per SIMD, from different workgroups and therefore out of step: the serial phases of one (y
// staging, fc1 outputs, eps, sections, x image: 40 % of the NS = 2 kernel) run under the head loop of the other, and the
// MFMAs take the VGPR form (no accumulator reads from AGPRs in the epilogue).
#define FB2_WP_OF(NS) (32 * (NS))

__host__ __device__ inline size_t fb2_wave_floats(int D, int J, int NS = 2) {
    const size_t a = (size_t)FB2_WP_OF(NS) * ef_ys(J) / 4, b = (size_t)FB2_WP_OF(NS) * pk_dse(D);   // response bytes | eps tile
    return ((a > b ? a : b) + 3) & ~(size_t)3;
}
__host__ __device__ inline size_t fb2_lds_bytes(int D, int J, int NS = 2) {
    return FB2_WAVES * fb2_wave_floats(D, J, NS) * sizeof(float) + (size_t)(pk_off_total(D) / 8 + 4) / 4 * 16;
}

// SH (NS = 1 only; round 4): the OFF tiles come to the four waves of a workgroup through ONE ring in LDS instead of four
// register streams.  Every wave of the plain form pulls the whole 1.5 MB image of the packed heads out of the L2 for its 32
// persons: 49 GB a launch at the headline shape, 17.6 TB/s over the kernel -- the rate at which this chip serves a table
// shared by every workgroup out of its L2s (MI355X_MICROARCH.md, indexed rows: 16.8-18.8 TB/s), and the reason why neither
// a third wave a SIMD nor a split of the roles over the waves moved the kernel (docs/NOTEBOOK.md).  Shared, a tile crosses
// the L2 -> CU path once a workgroup: a quarter of the bytes; the 9 fragment reads a tile and wave then come from LDS
// (256 B / clock / CU).
//   ring   three slots of one tile image (9 KB); tile t lives in slot t % 3.  Wave w transfers fragments w and w + 4 (wave 0
//          also the bias fragment) of a tile by LDS-DMA.
//   turn   at the end of iteration t (tile t in the MFMAs, tile t + 1 read into the other register set): wait for the
//          reads of tile t + 1 and for the own transfers of tile t + 2 (counted: those of tile t + 3 stay in flight),
//          barrier, then transfer tile t + 4 into the slot of tile t + 1.  A transfer has two iterations to land.
// Two workgroups a CU as before, so the ring has to fit beside them: the eps tile loses its row padding ([32][D] + 8 floats:
// D / 4 odd keeps the 16-byte reads of 16 lanes on distinct banks as well as the padded stride does) and the response bytes
// are staged in QUARTERS of 128 items (three 4 KB slots a wave, chunks XOR-swizzled over the row, the fourth quarter staged
// while the second runs) instead of whole rows (16 KB): 81,680 bytes a workgroup.  Same arithmetic in the same order: every
// output bit-identical to the plain form (tools/fwd2_bench.hip compares checksums of all nine).
#define FB2_RING_SLOTS 3
__host__ __device__ inline size_t fb2_tbl_bytes(int D) { return (size_t)(pk_off_total(D) / 8 + 4) / 4 * 16; }
__host__ __device__ inline size_t fb2s_wave_floats(int D) {       // eps tile | 3 quarter slots | the hT / hs stages
    size_t f = (size_t)32 * D + 8;
    if (f < 3072) f = 3072;
    return (f + 3) & ~(size_t)3;
}
__host__ __device__ inline size_t fb2s_lds_bytes(int D) {
    return FB2_WAVES * fb2s_wave_floats(D) * sizeof(float) + fb2_tbl_bytes(D) + (size_t)FB2_RING_SLOTS * FB_IMG_BYTES;
}
__host__ __device__ inline bool fb2s_shape_ok(int D, int J) {
    return D % 4 == 0 && ((D >> 2) & 1) && D >= 80 && J % 4 == 0 && J <= 512 && J > 384 && pk_off_total(D) / 32 >= 4 &&
           2 * ((fb2s_lds_bytes(D) + 1279) / 1280 * 1280) <= 160 * 1024;
}
template <int NS, bool SH = false>
__global__ __launch_bounds__(FB2_THREADS, (NS == 1 ? 2 : 1)) void k_mvn_enc_fwd_b2(
    EncDims dm, const uint8_t* __restrict__ y, const int64_t* __restrict__ rows, int64_t gid0,
    const uint8_t* __restrict__ w1img, const float* __restrict__ b1, const uint8_t* __restrict__ img,
    const uint32_t* __restrict__ gt2, const float* __restrict__ sc /*k_enc_scales*/, const float* __restrict__ eps_in,
    uint64_t seed, uint32_t step, const uint32_t* __restrict__ step_dev, uint32_t stream, float* __restrict__ h_out, float* __restrict__ x_out, float* __restrict__ eps_out, float* __restrict__ ldT,
    float* __restrict__ ent_out, float* __restrict__ hT_out, float* __restrict__ epsT_out, uint8_t* __restrict__ ximg_out,
    uint16_t* __restrict__ hs_out) {
    if (step_dev) step = *step_dev;                              // captured step: the Philox step lives in device memory
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef uint32_t u32x4w __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x2w __attribute__((ext_vector_type(2)));
    constexpr int H = 64, FB2_WP = FB2_WP_OF(NS);
    const int D = dm.D, J = dm.J;
    static_assert(!SH || NS == 1, "shared ring: one person set a wave");
    const int DS = SH ? D : pk_dse(D);
    const int YS = ef_ys(J);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const size_t WF = SH ? fb2s_wave_floats(D) : fb2_wave_floats(D, J, NS);
    float* R1 = smem + wave * WF;
    int8_t* Yi = (int8_t*)R1;                                 // phase A: [64][ysr] response bytes (SH: three quarter slots)
    float* eps_lds = R1;                                      // phase B: [64][DS]
    uint32_t* gt_lds = (uint32_t*)(smem + FB2_WAVES * WF);
    uint8_t* ring = (uint8_t*)gt_lds + fb2_tbl_bytes(D);       // SH: FB2_RING_SLOTS tile images
    const int64_t i0 = ((int64_t)blockIdx.x * FB2_WAVES + wave) * FB2_WP;
    const int p = l31;
    int64_t iu[NS];                                           // this lane's person of set u
#pragma unroll
    for (int u = 0; u < NS; ++u) iu[u] = i0 + 32 * u + p;
    // no early exit: every wave takes part in the barrier that publishes the group table; persons past the end compute on
    // clamped inputs and store nothing

    // diagnostic build (make EXTRA=-DFB2_STAMPS): cycles per phase of two workgroups, printed at the end; in the shipped
    // build no stamp executes
#ifdef FB2_STAMPS
    uint64_t st_[10]; int sn_ = 0;
    uint64_t w_lds_ = 0, w_vm_ = 0, w_bar_ = 0;                // SH: cycles of the turn's three waits, summed over the tiles
#define STAMP() do { __builtin_amdgcn_s_waitcnt(0); st_[sn_++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP() do {} while (0)
#endif
    STAMP();
    const int n_off = pk_off_total(D) / 32;                   // even
    const int n_sec = pk_sec(D) / 32;
    const int t_end = n_off + 2 * n_sec;
    for (int e = tid; e < 4 * n_off; e += FB2_THREADS) gt_lds[e] = gt2[e];
    const float w1_inv = sc[1], h_scale = sc[3], acc_inv = sc[4];
    // SH: this wave's fragments of tile t -> slot (tiles past the end: a harmless reload of the last one, so that the counted
    // wait of a turn always has the transfers of one tile behind the ones it waits for)
    const uint32_t ring_lb = SH ? lds_addr_uniform(ring) : 0u;
    auto ring_dma = [&](int t, int slot) __attribute__((always_inline)) {
        const int tc = t < t_end ? t : t_end - 1;
        const uint8_t* gb = img + (int64_t)tc * FB_IMG_BYTES + lane * 16;
        const uint32_t lb = ring_lb + (uint32_t)slot * FB_IMG_BYTES;
        dma16(gb + wave * 1024, lb + (uint32_t)wave * 1024u);
        dma16(gb + (wave + 4) * 1024, lb + (uint32_t)(wave + 4) * 1024u);
        if (wave == 0) dma16(gb + 8 * 1024, lb + 8u * 1024u);
    };
    if constexpr (SH) { ring_dma(0, 0); ring_dma(1, 1); ring_dma(2, 2); }      // land under fc1

    // ---------------------------------------------------------------- response rows of the wave's 64 persons
    // SH: quarter q (items 128 q .. + 127) of the 32 persons -> slot: [32 rows][8 chunks of 16 bytes], chunk c of row r at
    // position c ^ ((r ^ (r >> 3)) & 7); by DMA where every byte of the quarter rows lies inside y, word by word otherwise
    const bool qdma = SH && !rows && i0 + FB2_WP <= dm.nb && (i0 + FB2_WP - 1) * (int64_t)J + 512 <= dm.nb * (int64_t)J;
    auto stage_q = [&](int q, int slot) __attribute__((always_inline)) {
        if (qdma) {
            const uint32_t lb = lds_addr_uniform(R1) + (uint32_t)slot * 4096u;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const int row = 8 * d + (lane >> 3), pos = lane & 7, c = pos ^ ((row ^ (row >> 3)) & 7);
                dma16(y + (i0 + row) * J + 128 * q + 16 * c, lb + (uint32_t)d * 1024u);
            }
        } else {
            uint32_t* Yw = (uint32_t*)R1 + slot * 1024;
            for (int e = lane; e < 32 * 32; e += 64) {
                const int row = e >> 5, w = e & 31, byte = 128 * q + 4 * w;
                const int64_t ii = i0 + row;
                uint32_t v = 0u;
                if (byte < J && ii < dm.nb) {
                    const int64_t rr = rows ? rows[ii] : ii;
                    v = *(const uint32_t*)(y + rr * J + byte);
                }
                Yw[row * 32 + 4 * ((w >> 2) ^ ((row ^ (row >> 3)) & 7)) + (w & 3)] = v;
            }
        }
    };
    const int n_ydma = (FB2_WP * J + 1023) / 1024;
    const bool ydense = !rows && ((J >> 2) & 1) && i0 + FB2_WP <= dm.nb && (i0 * J + (int64_t)n_ydma * 1024 <= dm.nb * (int64_t)J) &&
                        (size_t)n_ydma * 1024 <= fb2_wave_floats(D, J, NS) * sizeof(float);
    const int ysr = ydense ? J : YS;
    if constexpr (SH) {
        stage_q(0, 0); stage_q(1, 1); stage_q(2, 2);
        if (qdma) vx_wait_vmem();
    } else if (ydense) {
        const uint8_t* src = y + i0 * J + 16 * lane;
        const uint32_t lb = lds_addr_uniform(R1);
        for (int d = 0; d < n_ydma; ++d) dma16(src + d * 1024, lb + (uint32_t)d * 1024u);
        vx_wait_vmem();
    } else {
        const int YW = YS / 4, JW = J / 4;
        uint32_t* Yw = (uint32_t*)R1;
        for (int e = lane; e < FB2_WP * YW; e += 64) {
            const int pp = e / YW, wq = e - pp * YW;
            const int64_t ii = i0 + pp;
            uint32_t v = 0u;
            if (wq < JW && ii < dm.nb) {
                const int64_t row = rows ? rows[ii] : ii;
                v = *(const uint32_t*)(y + row * J + 4 * wq);              // bytes 0/1/255 == int8 0/1/-1 (vi.py:689-691)
            }
            Yw[e] = v;
        }
    }
    __builtin_amdgcn_wave_barrier();
    STAMP();                                                  // 1: y staged
    // ---------------------------------------------------------------- phase A: fc1 (+ softplus) of both person sets
    f16x8 hb[NS][2][4];                                       // [set][term][k-step]: B fragments of every head tile
    {
        f32x16 acc[NS][2];
#pragma unroll
        for (int u = 0; u < NS; ++u) { acc[u][0] = zero16(); acc[u][1] = zero16(); }
        const int n_ks = (J + 15) / 16;
        auto loadA = [&](f16x8 (&Af)[4], int ks) __attribute__((always_inline)) {
            ks = ks < n_ks ? ks : n_ks - 1;
            const uint8_t* src = w1img + (int64_t)ks * FB_W1_KS + lane * 16;
#pragma unroll
            for (int f = 0; f < 4; ++f) Af[f] = *(const f16x8*)(src + f * 1024);
        };
        auto compute = [&](const f16x8 (&Af)[4], int ks) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                const int8_t* yp = SH ? Yi + ((ks >> 3) == 3 ? 0 : (ks >> 3)) * 4096 + p * 128 + 16 * ((ks & 7) ^ ((p ^ (p >> 3)) & 7)) + 8 * half
                                      : Yi + (32 * u + p) * ysr + 16 * ks + 8 * half;
                const f16x8 yb = fb_y_frag((const uint32_t*)yp);
                acc[u][0] = mfma_f16(Af[1], yb, acc[u][0]); acc[u][1] = mfma_f16(Af[3], yb, acc[u][1]);
                acc[u][0] = mfma_f16(Af[0], yb, acc[u][0]); acc[u][1] = mfma_f16(Af[2], yb, acc[u][1]);
            }
        };
        {
            // W1 fragments seven k-steps ahead (a k-step is ~300 cycles, an L2 round trip under load ~2 000)
            constexpr int RG = 8;
            f16x8 A[RG][4];
#pragma unroll
            for (int u = 0; u < RG - 1; ++u) loadA(A[u], u);
            for (int c = 0; c < n_ks; c += RG) {
                if constexpr (SH) {
                    if (c == 8) {                              // quarter 0 has been read: its slot takes quarter 3
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_wave_barrier();
                        stage_q(3, 0);
                    }
                    if (c == 24) {                             // quarter 3 has landed (the W1 loads in flight with it)
                        if (qdma) vx_wait_vmem();
                        __builtin_amdgcn_wave_barrier();
                    }
                }
#pragma unroll
                for (int u = 0; u < RG; ++u) {
                    loadA(A[(u + RG - 1) % RG], c + u + RG - 1);
                    if (c + u < n_ks) compute(A[u], c + u);
                }
            }
        }
        STAMP();                                              // 2: fc1 MFMA loop
        // Dimension-major outputs from the C layout are one 128-byte (hT) or 64-byte (hs) row piece per store instruction,
        // 128 of them per person set, and the phase is bound by store issue.  A whole wave on 16-byte aligned rows
        // transposes through the LDS region the response bytes have left instead: 16 bytes per lane, 20 instructions.
        constexpr int ST_T = 36, ST_S = 40;                   // row strides of the stages: [64][36] f32 | [2 * 64][40] u16
        // (NS = 1: the region holds one stage at a time -- the hs stage takes the place of the hT stage once that is stored)
        constexpr bool SEQ = NS == 1;
        const bool coal = i0 + FB2_WP <= dm.nb && (dm.nb & 7) == 0 && hT_out && hs_out &&
                          (((uintptr_t)hT_out | (uintptr_t)hs_out) & 15) == 0 &&
                          WF * sizeof(float) >= (SEQ ? 128 * ST_S * 2 : 64 * ST_T * 4 + 128 * ST_S * 2);   // wave-uniform
        float* stT = R1;
        uint16_t* stS = SEQ ? (uint16_t*)R1 : (uint16_t*)(R1 + 64 * ST_T);
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const int64_t i = iu[u];
            f32x16 hreg[2];
#pragma unroll
            for (int ht = 0; ht < 2; ++ht) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int hh0 = 32 * ht + 8 * g + 4 * half;
                    const float4 bb = *(const float4*)(b1 + hh0);
                    float4 hv;
                    hv.x = softplusf_(fmaf(acc[u][ht][4 * g + 0], w1_inv, bb.x));   // vi.py:449
                    hv.y = softplusf_(fmaf(acc[u][ht][4 * g + 1], w1_inv, bb.y));
                    hv.z = softplusf_(fmaf(acc[u][ht][4 * g + 2], w1_inv, bb.z));
                    hv.w = softplusf_(fmaf(acc[u][ht][4 * g + 3], w1_inv, bb.w));
                    hreg[ht][4 * g + 0] = hv.x; hreg[ht][4 * g + 1] = hv.y;
                    hreg[ht][4 * g + 2] = hv.z; hreg[ht][4 * g + 3] = hv.w;
                    if (i < dm.nb) *(float4*)(h_out + i * H + hh0) = hv;
                }
            }
            // the B fragments of every head tile: h 2^sh as two fp16 terms
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = hreg[s >> 1][8 * (s & 1) + j];
                split2h_frag(v, h_scale, hb[u][0][s], hb[u][1][s]);
            }
            if (coal) {
                __builtin_amdgcn_wave_barrier();                  // (second set: the copies of the first have been read)
#pragma unroll
                for (int ht = 0; ht < 2; ++ht)
#pragma unroll
                    for (int r = 0; r < 16; ++r) stT[(32 * ht + crow32(r, half)) * ST_T + p] = hreg[ht][r];
                const int64_t c0 = i0 + 32 * u;
                if constexpr (SEQ) {
                    __builtin_amdgcn_wave_barrier();
#ifndef FB2_NO_HT                                                 // (tools/fwd2_bench.hip -DFB2_NO_HT: what the fp32 hT copy costs)
#pragma unroll
                    for (int it = 0; it < 8; ++it) {              // hT: 64 rows x 8 pieces of 4 persons
                        const int e = lane + 64 * it, hh = e >> 3, g = e & 7;
                        *(f32x4*)(hT_out + (int64_t)hh * dm.nb + c0 + 4 * g) = *(const f32x4*)(stT + hh * ST_T + 4 * g);
                    }
#endif
                    __builtin_amdgcn_wave_barrier();
                }
                // the hs planes (operand of k_mvn_enc_bwd_w_b) take the fragments' two fp16 terms of h 2^sh
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const u32x4w w = __builtin_bit_cast(u32x4w, hb[u][t2][s]);
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int hh = 32 * (s >> 1) + crow32(8 * (s & 1) + j, half);
                            stS[(64 * t2 + hh) * ST_S + p] = (uint16_t)((j & 1) ? (w[j >> 1] >> 16) : (w[j >> 1] & 0xFFFFu));
                        }
                    }
                __builtin_amdgcn_wave_barrier();
                if constexpr (!SEQ) {
#pragma unroll
                    for (int it = 0; it < 8; ++it) {              // hT: 64 rows x 8 pieces of 4 persons
                        const int e = lane + 64 * it, hh = e >> 3, g = e & 7;
                        *(f32x4*)(hT_out + (int64_t)hh * dm.nb + c0 + 4 * g) = *(const f32x4*)(stT + hh * ST_T + 4 * g);
                    }
                }
#pragma unroll
                for (int it = 0; it < 8; ++it) {                  // hs: 2 x 64 rows x 4 pieces of 8 persons
                    const int e = lane + 64 * it, row = e >> 2, g = e & 3;
                    *(u32x4w*)(hs_out + (int64_t)row * dm.nb + c0 + 8 * g) = *(const u32x4w*)(stS + row * ST_S + 8 * g);
                }
            } else if (hT_out && i < dm.nb) {
#pragma unroll
                for (int ht = 0; ht < 2; ++ht)
#pragma unroll
                    for (int r = 0; r < 16; ++r) hT_out[(int64_t)(32 * ht + crow32(r, half)) * dm.nb + i] = hreg[ht][r];
            }
            if (!coal && hs_out && i < dm.nb) {                   // the two fp16 terms of hT 2^sh (as k_split2_f16)
                const int64_t plane = (int64_t)64 * dm.nb;
#pragma unroll
                for (int ht = 0; ht < 2; ++ht)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int64_t o = (int64_t)(32 * ht + crow32(r, half)) * dm.nb + i;
                        split2h_bits(hreg[ht][r] * h_scale, hs_out[o], hs_out[plane + o]);
                    }
            }
        }
    }
    STAMP();                                                  // 3: softplus, h outputs, fragments
    __builtin_amdgcn_wave_barrier();                          // response bytes no longer needed
    // ---------------------------------------------------------------- eps of the 64 persons (zero padded to DS) -> LDS, global
    const int nblk = D >> 2;                                  // D % 4 == 0 on this path
    for (int e = lane; e < FB2_WP * DS / 4 + (SH ? 2 : 0); e += 64) *(f32x4*)(R1 + 4 * e) = f32x4{0.f, 0.f, 0.f, 0.f};   // DS % 4 == 0 (SH: + the 8 floats behind the last row)
    __builtin_amdgcn_wave_barrier();
    // (unrolled: one Philox call is a dependent chain of ~60 instructions, and this wave is alone on its SIMD)
#pragma unroll 5
    for (int e = lane; e < FB2_WP * nblk; e += 64) {
        const int pp = e / nblk, blk = e - pp * nblk;
        int64_t ii = i0 + pp;
        const bool live = ii < dm.nb;
        if (!live) ii = dm.nb - 1;                             // absent persons: any finite values, never stored
        f32x4 z;
        if (eps_in) {
            z = *(const f32x4*)(eps_in + ii * D + 4 * blk);
        } else {
            const int64_t row = rows ? rows[ii] : ii;
            z = philox_normal4(seed, step, stream, gid0 + row, (uint32_t)blk);
        }
        *(f32x4*)(eps_lds + pp * DS + 4 * blk) = z;             // (eps_out is written at the end: its region stages x first)
    }
    __builtin_amdgcn_wave_barrier();
    // dimension-major copy of eps: a whole wave writes 16 bytes (4 persons of one k) per lane, D / 4 store instructions
    // instead of D
    const bool coalE = i0 + FB2_WP <= dm.nb && (dm.nb & 3) == 0 && epsT_out && ((uintptr_t)epsT_out & 15) == 0;
    if (coalE) {
        constexpr int PG = FB2_WP / 4;                         // groups of four persons
        for (int e = lane; e < D * PG; e += 64) {
            const int k = e / PG, g = e - k * PG;
            const float* ec = eps_lds + 4 * g * DS + k;
            *(f32x4*)(epsT_out + (int64_t)k * dm.nb + i0 + 4 * g) = f32x4{ec[0], ec[DS], ec[2 * DS], ec[3 * DS]};
        }
    }
    float eps2[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        const float* er = eps_lds + (32 * u + p) * DS;
        if (!coalE && epsT_out && iu[u] < dm.nb) {
#pragma unroll 4
            for (int k = half; k < D; k += 2) epsT_out[(int64_t)k * dm.nb + iu[u]] = er[k];
        }
        float s2 = 0.f;
        for (int k = 0; k < D; ++k) s2 = fmaf(er[k], er[k], s2);
        eps2[u] = s2;
    }
    STAMP();                                                  // 3: eps
    // ---------------------------------------------------------------- phase B: packed head rows, 32 per tile
    struct TileRegs { f16x8 a[2][4]; f16x8 bias; };
    auto pull = [&](TileRegs& R, int t) __attribute__((always_inline)) {
        const int tc = t < t_end ? t : t_end - 1;
        const uint8_t* gb = img + (int64_t)tc * FB_IMG_BYTES + lane * 16;
        R.bias = *(const f16x8*)(gb + FB_A_BYTES);
#pragma unroll
        for (int sp = 1; sp >= 0; --sp)
#pragma unroll
            for (int s = 0; s < 4; ++s) R.a[sp][s] = *(const f16x8*)(gb + (sp * 4 + s) * 1024);
    };
    f16x8 cfrag;                                              // the bias product's constant 2^(sw + sh - sb)
    {
        const _Float16 c16 = (_Float16)sc[6];
#pragma unroll
        for (int j = 0; j < 8; ++j) cfrag[j] = c16;
    }
    auto mma_all = [&](const TileRegs& R, int u) __attribute__((always_inline)) -> f32x16 {
        f32x16 a = mfma_f16(R.bias, cfrag, zero16());
#pragma unroll
        for (int s = 0; s < 4; ++s) a = mfma_f16(R.a[1][s], hb[u][0][s], a);
#pragma unroll
        for (int s = 0; s < 4; ++s) a = mfma_f16(R.a[0][s], hb[u][1][s], a);
#pragma unroll
        for (int s = 0; s < 4; ++s) a = mfma_f16(R.a[0][s], hb[u][0][s], a);
        return a;
    };
    // ---- OFF section: the partial sum of the current k stays in a register per set and goes to x_out when the k ends
    float cur_part[NS];
    const char* ep_h[NS];
    float* xrow[NS];
    float* xst[NS];                                           // staging of the OFF sums: element k at xst + (k << st_sh)
    const bool whole = i0 + FB2_WP <= dm.nb;                  // wave-uniform
    const int st_sh = whole ? (NS == 2 ? 6 : 5) : 0;
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        cur_part[u] = 0.f;
        ep_h[u] = (const char*)(eps_lds + (32 * u + p) * DS + 4 * half);
        xrow[u] = x_out + (iu[u] < dm.nb ? iu[u] : dm.nb - 1) * D;
        xst[u] = whole ? eps_out + i0 * D + 32 * u + p : xrow[u];
    }
    struct EpiOps { float4 e4[NS][4]; };
    auto epi_read = [&](EpiOps& E, const uint4& c) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            E.e4[u][0] = *(const float4*)(ep_h[u] + (c.x & 0xFFFu));
            E.e4[u][1] = *(const float4*)(ep_h[u] + (c.y & 0xFFFu));
            E.e4[u][2] = *(const float4*)(ep_h[u] + (c.z & 0xFFFu));
            E.e4[u][3] = *(const float4*)(ep_h[u] + (c.w & 0xFFFu));
        }
    };
    auto epi_group = [&](const f32x16 (&a)[NS], const EpiOps& E, uint32_t code, int g) __attribute__((always_inline)) {
        // (asm: plain fmaf calls are SLP-packed into v_pk_fma_f32 -- with v_mov to pair the operands -- and packed f32
        // instructions wait for the matrix pipe beside MFMAs; the chains of the two person sets alternate)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                const float ev = j == 0 ? E.e4[u][g].x : j == 1 ? E.e4[u][g].y : j == 2 ? E.e4[u][g].z : E.e4[u][g].w;
                float av = a[u][4 * g + j];
                // The hazard recognizer does not look into asm statements, and the wait states between an MFMA and a vector
                // read of its result are software's to insert: the FIRST read of a finished chain goes through an instruction
                // the compiler sees (an identity DPP move of one register -- all sixteen are written by the chain's last
                // MFMA, and volatile keeps the other reads behind this one).  Found with the three-workgroup form, whose
                // prologue had the first FMA four instructions behind the MFMA and read the accumulator of the tile before.
                if (g == 0 && j == 0) av = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, av), 0xE4, 0xF, 0xF, true));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(cur_part[u]) : "v"(av), "v"(ev));
            }
        if (__builtin_expect((int)code < 0, 0)) {                                 // wave-uniform, rare: the k ends here
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                const float tot = half_sum32(cur_part[u]) * acc_inv;               // the powers of two come off once per k
                if (half == 0 && iu[u] < dm.nb) *(float*)((char*)xst[u] + (((code >> 12) & 0xFFFu) << st_sh)) = tot;
                cur_part[u] = 0.f;
            }
        }
    };

    if constexpr (SH) vx_wait_vmem();                          // this wave's transfers of tiles 0 .. 2
    __syncthreads();                                           // the group table in LDS is complete (SH: and the first three tiles)
    // (two tiles ahead with a third register set: no change, tools/fwd2_bench.hip)
    TileRegs RA, RB;
    int s_rd = 1;                                              // SH: the slot of tile t + 1
    auto ring_read = [&](TileRegs& R, int slot) __attribute__((always_inline)) {
        const uint8_t* lb = ring + slot * FB_IMG_BYTES + lane * 16;
        R.bias = *(const f16x8*)(lb + FB_A_BYTES);
#pragma unroll
        for (int sp = 1; sp >= 0; --sp)
#pragma unroll
            for (int s = 0; s < 4; ++s) R.a[sp][s] = *(const f16x8*)(lb + (sp * 4 + s) * 1024);
    };
    f32x16 accP[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) accP[u] = zero16();
    uint4 codeP = make_uint4(0, 0, 0, 0);
    auto off_iter = [&](TileRegs& Rc, TileRegs& Rn, int t, auto firstc) __attribute__((always_inline)) {
        constexpr bool first = decltype(firstc)::value;
        EpiOps E;
        if constexpr (!first) epi_read(E, codeP);
        if constexpr (SH) ring_read(Rn, s_rd); else pull(Rn, t + 1);
        // two chains per tile (one per person set) on the same fragments; the epilogue of the previous tile between the
        // product groups
        f32x16 a[NS];
#pragma unroll
        for (int u = 0; u < NS; ++u) a[u] = mfma_f16(Rc.bias, cfrag, zero16());
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int u = 0; u < NS; ++u) a[u] = mfma_f16(Rc.a[1][s], hb[u][0][s], a[u]);
        if constexpr (!first) epi_group(accP, E, codeP.x, 0);
#pragma unroll
        for (int s = 2; s < 4; ++s)
#pragma unroll
            for (int u = 0; u < NS; ++u) a[u] = mfma_f16(Rc.a[1][s], hb[u][0][s], a[u]);
#pragma unroll
        for (int u = 0; u < NS; ++u) a[u] = mfma_f16(Rc.a[0][0], hb[u][1][0], a[u]);
        if constexpr (!first) epi_group(accP, E, codeP.y, 1);
#pragma unroll
        for (int s = 1; s < 4; ++s)
#pragma unroll
            for (int u = 0; u < NS; ++u) a[u] = mfma_f16(Rc.a[0][s], hb[u][1][s], a[u]);
        if constexpr (!first) epi_group(accP, E, codeP.z, 2);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int u = 0; u < NS; ++u) a[u] = mfma_f16(Rc.a[0][s], hb[u][0][s], a[u]);
        if constexpr (!first) epi_group(accP, E, codeP.w, 3);
#pragma unroll
        for (int s = 2; s < 4; ++s)
#pragma unroll
            for (int u = 0; u < NS; ++u) a[u] = mfma_f16(Rc.a[0][s], hb[u][0][s], a[u]);
#pragma unroll
        for (int u = 0; u < NS; ++u) accP[u] = a[u];
        const uint4 cv = *(const uint4*)(gt_lds + 4 * t);
        codeP.x = __builtin_amdgcn_readfirstlane(cv.x); codeP.y = __builtin_amdgcn_readfirstlane(cv.y);
        codeP.z = __builtin_amdgcn_readfirstlane(cv.z); codeP.w = __builtin_amdgcn_readfirstlane(cv.w);
        if constexpr (SH) {                                    // the turn of the ring (see the head of the kernel)
#ifdef FB2_STAMPS
            const uint64_t tw0_ = __builtin_amdgcn_s_memtime();
#endif
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef FB2_STAMPS
            const uint64_t tw1_ = __builtin_amdgcn_s_memtime();
#endif
            if (wave == 0) __builtin_amdgcn_s_waitcnt(0x0F70 | 3); else __builtin_amdgcn_s_waitcnt(0x0F70 | 2);
#ifdef FB2_STAMPS
            const uint64_t tw2_ = __builtin_amdgcn_s_memtime();
#endif
            __builtin_amdgcn_s_barrier();
#ifdef FB2_STAMPS
            { const uint64_t tw3_ = __builtin_amdgcn_s_memtime(); w_lds_ += tw1_ - tw0_; w_vm_ += tw2_ - tw1_; w_bar_ += tw3_ - tw2_; }
#endif
            ring_dma(t + 4, s_rd);
            s_rd = s_rd == FB2_RING_SLOTS - 1 ? 0 : s_rd + 1;
        }
    };
    if constexpr (SH) {                                        // (tiles 0 .. 2 have landed: the wait in front of the barrier above)
        ring_read(RA, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        ring_dma(3, 0);
    } else {
        pull(RA, 0);
    }
    off_iter(RA, RB, 0, std::true_type{});
    off_iter(RB, RA, 1, std::false_type{});
    for (int t = 2; t < n_off; t += 2) {
        off_iter(RA, RB, t, std::false_type{});
        off_iter(RB, RA, t + 1, std::false_type{});
    }
    {
        EpiOps E;
        epi_read(E, codeP);
        epi_group(accP, E, codeP.x, 0);
        epi_group(accP, E, codeP.y, 1);
        epi_group(accP, E, codeP.z, 2);
        epi_group(accP, E, codeP.w, 3);
    }
    vx_wait_vmem();                                            // the OFF sums of this wave have reached memory: the sections read them back
    STAMP();                                                  // 4: OFF loop
    // ---- sections: DIAG and LOC tile of each 32-row block together: x = x_off + exp(M_kk) eps_k + loc_k
    float ent_acc[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) ent_acc[u] = 0.f;
    // the tiles of block kb + 1 are requested as soon as the MFMAs of block kb have read theirs (tile indices past the end are
    // clamped by pull: a harmless extra load)
    pull(RA, n_off);
    pull(RB, n_off + n_sec);
    for (int kb = 0; kb < n_sec; ++kb) {
        const int k0 = 32 * kb;
        if (k0 >= D) break;                                    // padding blocks of the sections
        f32x4 xo[NS][4];
#pragma unroll
        for (int u = 0; u < NS; ++u) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int kk = k0 + 8 * g + 4 * half;
#pragma unroll
                for (int j = 0; j < 4; ++j) xo[u][g][j] = (kk < D) ? xst[u][(int64_t)(kk + j) << st_sh] : 0.f;
            }
            if (k0 == 0 && half == 0) xo[u][0][0] = 0.f;       // k = 0 has no OFF rows: nothing was stored there
        }
        f32x16 aD[NS], aL[NS];
#pragma unroll
        for (int u = 0; u < NS; ++u) aD[u] = mma_all(RA, u);
        pull(RA, n_off + kb + 1);
#pragma unroll
        for (int u = 0; u < NS; ++u) aL[u] = mma_all(RB, u);
        pull(RB, n_off + n_sec + kb + 1);
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const bool live = iu[u] < dm.nb;
            const float* er = eps_lds + (32 * u + p) * DS;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int kk = k0 + 8 * g + 4 * half;
                if (kk < D) {
                    const f32x4 ev = *(const f32x4*)(er + kk);
                    f32x4 xn;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float mkk = aD[u][4 * g + j] * acc_inv;
                        const float ld = __expf(mkk);                // exp(diag M) eps_k, entropy, ldT  (vi.py:686)
                        xn[j] = fmaf(aL[u][4 * g + j], acc_inv, fmaf(ld, ev[j], xo[u][g][j]));
                        ent_acc[u] += mkk;
                        if (live) ldT[(int64_t)(kk + j) * dm.nb + iu[u]] = ld;
                    }
                    if (live) *(f32x4*)(xrow[u] + kk) = xn;
                }
            }
        }
    }
    vx_wait_vmem();                                            // every staged sum has been read: the eps_out region is free
    for (int e = lane; e < FB2_WP * nblk; e += 64) {
        const int pp = e / nblk, blk = e - pp * nblk;
        if (i0 + pp < dm.nb) *(f32x4*)(eps_out + (i0 + pp) * D + 4 * blk) = *(const f32x4*)(eps_lds + pp * DS + 4 * blk);
    }
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        ent_acc[u] += __shfl_xor(ent_acc[u], 32, 64);
        if (half == 0 && iu[u] < dm.nb) ent_out[iu[u]] = 0.5f * eps2[u] + ent_acc[u];   // -log q + const = 0.5|eps|^2 + sum_k M_kk
    }
    STAMP();                                                  // 5: sections + eps_out
    // ---------------------------------------------------------------- the likelihood kernel's operand image of x
    if (ximg_out && i0 < (dm.nb + 63) / 64 * 64) {
        const int pbase = (int)(i0 & 63);                      // NS = 1: this wave's 32 persons are one half of a 64-person tile
        vx_wait_vmem();                                        // x of this wave is complete in memory
        __builtin_amdgcn_wave_barrier();
        // x_aug = [x, 1, 0..] 2^LH_XEXP as two fp16 terms in the LDS tile order of k_irt_lik_h (lb_xoff): the wave's 64 persons are
        // one whole tile; absent persons: all-zero rows
        const int pvi = (int)((dm.nb - i0) < FB2_WP ? (dm.nb - i0) : FB2_WP);
        uint8_t* out = ximg_out + (i0 >> 6) * LH_XT_BYTES;
        // seven work items per lane in flight (loads of all, then splits and stores): one at a time is a chain of L2
        // latencies
        constexpr int XU = 7;
        static_assert((FB2_WP * 2 * LB_NKS) % (64 * XU) == 0, "x image work items");
        for (int e0 = lane; e0 < FB2_WP * 2 * LB_NKS; e0 += 64 * XU) {
            f32x4 q[XU][2];
            uint32_t off[XU];
#pragma unroll
            for (int w = 0; w < XU; ++w) {
                // per 32 persons: 32 consecutive lanes fill one 512-byte subtile (8 persons x 4 chunks), then the 256-byte half
                // subtiles (8 persons x 2 chunks)
                const int e = e0 + 64 * w;
                const int hs2 = e / (32 * 2 * LB_NKS), e2 = e - hs2 * (32 * 2 * LB_NKS);
                int pp, ch;
                if (e2 < 4 * 3 * 32) {
                    const int blk = e2 >> 5, r = e2 & 31;
                    pp = 8 * (blk / 3) + (r >> 2);
                    ch = 4 * (blk % 3) + (r & 3);
                } else {
                    const int r = e2 - 4 * 3 * 32;
                    pp = 8 * (r >> 4) + ((r >> 1) & 7);
                    ch = 12 + (r & 1);
                }
                pp += 32 * hs2;
                off[w] = lb_xoff(pbase + pp, ch);
                const float* xr = x_out + (i0 + (pp < pvi ? pp : 0)) * D;
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {               // D % 4 == 0: a quad is inside the row, or at k == D, or past it
                    const int k0 = 8 * ch + 4 * h2;
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (pp < pvi) {
                        if (k0 < D) v = *(const f32x4*)(xr + k0);
                        else if (k0 == D) v[0] = 1.0f;
                    }
                    q[w][h2] = v;
                }
            }
#pragma unroll
            for (int w = 0; w < XU; ++w) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = q[w][j >> 2][j & 3];
                f16x8 fh, fl;
                if (lh_split_x(v, fh, fl)) atomicOr((uint32_t*)(ximg_out + (dm.nb + 63) / 64 * (int64_t)LH_XT_BYTES), 1u);
                *(f16x8*)(out + off[w]) = fh;
                *(f16x8*)(out + LB_PLANE + off[w]) = fl;
            }
        }
    }
#ifdef FB2_STAMPS
    STAMP();                                                  // 6: x image
    if ((blockIdx.x == 100 || blockIdx.x == 2000 || blockIdx.x == 5000) && lane == 0 && (wave == 0 || wave == 3))
        printf("STAMPS blk %d wave %d: y %llu fc1 %llu hout %llu eps %llu off %llu sec %llu ximg %llu total %llu | turn waits: lds %llu dma %llu barrier %llu\n",
               (int)blockIdx.x, wave, st_[1] - st_[0], st_[2] - st_[1], st_[3] - st_[2], st_[4] - st_[3], st_[5] - st_[4], st_[6] - st_[5],
               st_[7] - st_[6], st_[7] - st_[0], w_lds_, w_vm_, w_bar_);
#endif
}
