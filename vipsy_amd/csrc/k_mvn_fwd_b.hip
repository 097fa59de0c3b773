// Amortized MVN guide forward with the head GEMM on the fp16 MFMA (two-term operand splitting "f16x2", vx_common.h; fp32
// accumulate): the same mathematics and outputs as k_mvn_enc_fwd_p (k_mvn_packed.hip; vi.py:448-455,686-693).
//   M[p, r] = bias[r] + sum_hh Wp[r][hh] h[p][hh]  is computed as three fp16 products per 16-deep k-step
//   (lo*hi, hi*lo, hi*hi of the scaled operands) on v_mfma_f32_32x32x16_f16: 13 MFMAs of 32 cycles per 32x32 tile
//   (round 2: six bf16 products, 25 MFMAs; round 1: 33 fp32 MFMAs of 64 cycles), at the accuracy of the fp32 chain.
// Both operands are reused, so the splitting costs nothing in the loop:
//   Wp : scaled and split once per step by k_pack_heads_b into a per-tile IMAGE that is exactly the order the fragments
//        are read in (8 KB of fp16 fragments + a bias fragment per 32-row tile);
//   h  : scaled and split once per 32-person wave tile, in registers; the C layout of the fc1 MFMA already is the B
//        fragment order (k-step s, lane half, element j  <->  hidden unit 16 s + 8 (j >> 2) + 4 half + (j & 3)).
// The powers of two (k_enc_scales) come off the accumulator where it is consumed: the OFF sums once per k, the DIAG /
// LOC values per element.
// Every wave streams the tile images by itself, global (L2) -> registers, one tile ahead: the 9 fragment loads of tile
// t + 1 are issued at the head of tile t and consumed a tile later, so no LDS ring, no DMA and no workgroup barrier sit
// in the head loop.  The epsilon epilogue of tile t - 1 goes between the MFMA groups of tile t (the MFMA leaves the
// vector port free for 24 of its 32 cycles).
// (included by vx_abi.hip after k_mvn_packed.hip and k_mvn_bwd_b.hip)

#define FB_THREADS 256
#define FB_WAVES 4                                                   // one per SIMD: 5 or 6 (LDS allows 6) load the SIMDs unevenly -- measured 10-20 % slower

#define FB_WP 32
#define FB_A_BYTES 8192                                              // 2 terms x 4 k-steps x 1 KB
#define FB_AUX_BYTES 1024                                            // the bias fragment
#define FB_IMG_BYTES (FB_A_BYTES + FB_AUX_BYTES)                     // tile image in global memory

__host__ __device__ inline int fb_tiles(int D) { return (pk_off_total(D) + 2 * pk_sec(D)) / 32; }
__host__ __device__ inline int64_t fb_img_floats(int D) {               // tile images + the OFF group table
    return (int64_t)fb_tiles(D) * (FB_IMG_BYTES / 4) + (pk_off_total(D) / 8 + 8 + 3) / 4 * 4;
}
// A wave's region: the packed encoder's, and never less than the 32 x 64 floats of the SPLIT form's fc1 exchange
// (each wave hands its partial pre-activations, 32 registers of 64 lanes, to the other three).
__host__ __device__ inline size_t fb_wave_floats(int D, int J) {
    const size_t w = enc_p_wave_floats(D, J);
    return w < 2048 ? 2048 : w;
}
__host__ __device__ inline size_t fb_lds_bytes(int D, int J) {
    return FB_WAVES * fb_wave_floats(D, J) * sizeof(float) +
           (size_t)(pk_off_total(D) / 8 + 4) / 4 * 16 + 512;            // wave regions | OFF group table | SPLIT: entropy parts
}

// ---- powers of two of the f16x2 operands (vx_common.h), recomputed from the parameters every step by ONE small block.
// sc[]: 0 W1 scale | 1 its inverse | 2 head-weight scale 2^sw | 3 h scale 2^sh | 4 2^-(sw + sh) | 5 bias scale 2^sb |
//       6 the bias product's constant 2^(sw + sh - sb) | 7 bound of h | 8 max |W21, W22| | 9 max |b21, b22| | 10 max |W1|
// h = softplus(W1 y + b1) with y in {-1, 0, 1} is bounded by softplus(max_u (|W1[u, :]|_1 + |b1[u]|)): a scale from that
// bound cannot overflow, and a bound a few binades above the values costs nothing (fp16 pairs keep 2^-22 relative down
// to 2^-3 and 2^-25 absolute below it).
#define FB_SC_BLOCKS 64
#define FB_SC_PART 16                                                // sc[FB_SC_PART + 4 b + i]: block b's maxima (the fused pack launches)
#define FB_NSCALES (FB_SC_PART + 4 * FB_SC_BLOCKS)
// two launches: FB_SC_BLOCKS blocks take the maxima (integer atomicMax on the bit patterns of non-negative floats:
// order-independent) into sc[11..14] (cleared by k_pack_heads, which runs first), one thread turns them into the powers of
// two.  (One block over the 0.36 M parameters took 72 us a step.)
#define FB_SC_BLOCKS 64
// the maxima one block of FB_SC_BLOCKS sees (blk = its index; 256 threads): lane 0 of every wave holds the wave's four values
__device__ __forceinline__ void enc_scales_block_max(int blk, int D, int J, const float* __restrict__ W1, const float* __restrict__ b1,
                                                     const float* __restrict__ W21, const float* __restrict__ b21,
                                                     const float* __restrict__ W22, const float* __restrict__ b22,
                                                     float& mw, float& mb, float& m1, float& l1) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = D * (D + 1) / 2;
    const int gt = blk * 256 + tid, gn = FB_SC_BLOCKS * 256;
    mw = 0.f; mb = 0.f; m1 = 0.f; l1 = 0.f;
    // (maxima: any order.  16-byte loads, four of them in flight: the loop over single floats was 20 dependent trips of an
    // L2 latency each, most of the 9 us of this launch)
    auto max4 = [&](const float* __restrict__ w, int n) __attribute__((always_inline)) {
        if (((uintptr_t)w & 15) == 0 && (n & 3) == 0) {
            const f32x4* w4 = (const f32x4*)w;
            const int n4 = n >> 2;
            for (int e = gt; e < n4; e += 4 * gn) {
                f32x4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = (e + q * gn < n4) ? w4[e + q * gn] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    mw = fmaxf(fmaxf(mw, fmaxf(fabsf(v[q][0]), fabsf(v[q][1]))), fmaxf(fabsf(v[q][2]), fabsf(v[q][3])));
            }
        } else {
            for (int e = gt; e < n; e += gn) mw = fmaxf(mw, fabsf(w[e]));
        }
    };
    max4(W21, D * 64);
    max4(W22, T * 64);
    for (int e = gt; e < D; e += gn) mb = fmaxf(mb, fabsf(b21[e]));
    for (int e = gt; e < T; e += gn) mb = fmaxf(mb, fabsf(b22[e]));
    for (int u = blk * 4 + wave; u < 64; u += FB_SC_BLOCKS * 4) {          // hidden unit u: |W1[u, :]|_1 + |b1[u]|
        float sacc = 0.f;
        const float* wu = W1 + (int64_t)u * J;
        for (int j0 = 0; j0 < J; j0 += 512) {                              // eight loads in flight; the sum in the order of j
            float w[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) w[q] = (j0 + 64 * q + lane < J) ? fabsf(wu[j0 + 64 * q + lane]) : 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (j0 + 64 * q + lane < J) { sacc += w[q]; m1 = fmaxf(m1, w[q]); }
        }
        sacc = wave_sum(sacc) + fabsf(b1[u]);
        l1 = fmaxf(l1, sacc);
    }
    mw = wave_max_dpp(mw); mb = wave_max_dpp(mb); m1 = wave_max_dpp(m1);
}
__global__ __launch_bounds__(256) void k_enc_scales_max(int D, int J, const float* __restrict__ W1, const float* __restrict__ b1,
                                                        const float* __restrict__ W21, const float* __restrict__ b21,
                                                        const float* __restrict__ W22, const float* __restrict__ b22,
                                                        float* __restrict__ sc) {
    float mw, mb, m1, l1;
    enc_scales_block_max(blockIdx.x, D, J, W1, b1, W21, b21, W22, b22, mw, mb, m1, l1);
    if ((threadIdx.x & 63) == 0) {
        uint32_t* w = (uint32_t*)(sc + 11);
        atomicMax(w + 0, __builtin_bit_cast(uint32_t, mw)); atomicMax(w + 1, __builtin_bit_cast(uint32_t, mb));
        atomicMax(w + 2, __builtin_bit_cast(uint32_t, m1)); atomicMax(w + 3, __builtin_bit_cast(uint32_t, l1));
    }
}
// the maxima -> the eleven scale words (one thread's arithmetic)
__device__ __forceinline__ void enc_scales_from_max(float mw, float mb, float m1, float l1, float* __restrict__ sc) {
    const float hbound = 1.001f * (fmaxf(l1, 0.f) + log1pf(expf(-fabsf(l1)))) + 1e-30f;     // softplus(l1), a hair over
    const int sw1 = f16_scale_exp(m1), sh = f16_scale_exp(hbound);
    int sw = f16_scale_exp(mw), sb = f16_scale_exp(mb), eb = sw + sh - sb;
    // the bias enters the accumulator chain as one more product, (b 2^sb) x 2^eb: the constant must be an fp16 normal
    if (eb > 15) { sw -= eb - 15; eb = 15; }                // a bias far above |W| |h|: the weights give up headroom
    if (eb < -14) { sb = sw + sh + 14; eb = -14; }          // a bias far below: it sits lower in the fp16 range
    sc[0] = ldexpf(1.f, sw1); sc[1] = ldexpf(1.f, -sw1);
    sc[2] = ldexpf(1.f, sw); sc[3] = ldexpf(1.f, sh); sc[4] = ldexpf(1.f, -(sw + sh));
    sc[5] = ldexpf(1.f, sb); sc[6] = ldexpf(1.f, eb);
    sc[7] = hbound; sc[8] = mw; sc[9] = mb; sc[10] = m1;
}
__global__ void k_enc_scales(float* __restrict__ sc) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    enc_scales_from_max(sc[11], sc[12], sc[13], sc[14], sc);
}

// tile image: fragment (term sp, k-step s) at byte (sp * 4 + s) * 1024 + lane * 16, lane = 32 half + row; sp = 0: heads, 1:
// remainders of Wp 2^sw; element j of it = Wp[32 T + row][16 s + 8 (j >> 2) + 4 half + (j & 3)];  a 9th fragment at
// FB_A_BYTES carries the bias
// gt2[group] (OFF groups only): byte offset of eps[l0] | byte offset of x[k] << 12 | (last group of its k) << 31
// DIRECT: the rows come straight from W21 / W22 / b21 / b22 through pk_decode (what k_pack_heads would have put into Wp / bp)
// and the block also writes its four words of gtab -- the fused pack then has no packed copy to wait for.
template <bool DIRECT = false>
__device__ __forceinline__ void pack_heads_b_tile(int T, int n_off_groups, const float* __restrict__ Wp, const float* __restrict__ bp,
                                                  const uint32_t* __restrict__ gtab, float w_scale, float b_scale,
                                                  uint8_t* __restrict__ img, uint32_t* __restrict__ gt2, int D = 0,
                                                  const float* __restrict__ W21 = nullptr, const float* __restrict__ b21 = nullptr,
                                                  const float* __restrict__ W22 = nullptr, const float* __restrict__ b22 = nullptr,
                                                  uint32_t* __restrict__ gtab_out = nullptr) {
    uint8_t* out = img + (int64_t)T * FB_IMG_BYTES;
    const int tid = threadIdx.x;
    const int Tt = D * (D + 1) / 2;
    if constexpr (DIRECT) {
        if (tid < 4) {
            const int G = 4 * T + tid;
            int src;
            uint32_t c, cn = 0u;
            pk_decode(8 * G, D, Tt, src, c);
            gtab_out[G] = c;
            if (G < n_off_groups) {
                if (G + 1 < n_off_groups) pk_decode(8 * (G + 1), D, Tt, src, cn);
                const uint32_t k = (c >> 12) & 0xFFFFu, l0 = c & 0xFFFu;
                const bool last = (G + 1 == n_off_groups) || (((cn >> 12) & 0xFFFFu) != k);
                gt2[G] = (4u * l0) | ((4u * k) << 12) | (last ? 0x80000000u : 0u);
            }
        }
    } else if (tid < 4 && 4 * T + tid < n_off_groups) {
        const int G = 4 * T + tid;
        const uint32_t c = gtab[G], k = (c >> 12) & 0xFFFFu, l0 = c & 0xFFFu;
        const bool last = (G + 1 == n_off_groups) || (((gtab[G + 1] >> 12) & 0xFFFFu) != k);
        gt2[G] = (4u * l0) | ((4u * k) << 12) | (last ? 0x80000000u : 0u);
    }
    // (s, lane, j) = e = tid + 256 i: row = lane & 31 is the thread's own for every i; all eight loads first (one L2 latency,
    // not eight)
    const int row = (tid >> 3) & 31;
    const float* wrow;
    int bsrc = -1;
    if constexpr (DIRECT) {
        uint32_t gc;
        pk_decode(T * 32 + row, D, Tt, bsrc, gc);
        wrow = bsrc < 0 ? nullptr : (bsrc < Tt ? W22 + (int64_t)bsrc * 64 : W21 + (int64_t)(bsrc - Tt) * 64);
    } else {
        wrow = Wp + ((int64_t)T * 32 + row) * 64;
    }
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = tid + 256 * i;
        const int j = e & 7, lane = (e >> 3) & 63, sx = e >> 9;
        const int half = lane >> 5;
        v[i] = wrow ? wrow[16 * sx + 8 * (j >> 2) + 4 * half + (j & 3)] : 0.f;
    }
    float bv = 0.f;                                                       // bias fragment: lane = row (half 0), thread 8 lane + j, j < 2
    const bool bias_thread = tid < 256 && (tid & 7) < 2;
    if (bias_thread) {
        if constexpr (DIRECT) {
            // (thread 8 r + j holds row r = (tid >> 3) & 31 already)
            bv = bsrc < 0 ? 0.f : (bsrc < Tt ? b22[bsrc] : b21[bsrc - Tt]);
        } else {
            bv = bp[T * 32 + (tid >> 3)];
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = tid + 256 * i;
        const int j = e & 7, lane = (e >> 3) & 63, sx = e >> 9;
        uint16_t* o = (uint16_t*)(out + sx * 1024 + lane * 16) + j;
        split2h_bits(w_scale * v[i], o[0], o[2048]);                      // + 4 fragments = 4096 bytes
    }
    // bias fragment (9th): lane = row (half 0), elements 0, 1 = the two fp16 terms of the row's bias 2^sb, the rest zero; one
    // MFMA against a fragment of the constant 2^(sw + sh - sb) starts the accumulator chain from the (scaled) bias
    for (int e = tid; e < FB_AUX_BYTES / 2; e += 256) {
        const int j = e & 7;
        uint16_t w = 0;
        if (e < 256 && j < 2) {
            uint16_t bh, bl;
            split2h_bits(b_scale * bv, bh, bl);
            w = j == 0 ? bh : bl;
        }
        ((uint16_t*)(out + FB_A_BYTES))[e] = w;
    }
}
__global__ __launch_bounds__(256) void k_pack_heads_b(int n_tiles, int n_off_groups, const float* __restrict__ Wp, const float* __restrict__ bp,
                               const uint32_t* __restrict__ gtab, const float* __restrict__ sc, uint8_t* __restrict__ img,
                               uint32_t* __restrict__ gt2) {
    if ((int)blockIdx.x >= n_tiles) return;
    pack_heads_b_tile(blockIdx.x, n_off_groups, Wp, bp, gtab, sc[2], sc[5], img, gt2);
}

// fc1 weights as k-step images: k-step ks (items 16 ks .. + 15), fragment (hidden tile ht, term sp) at byte
// ks * FB_W1_KS + (ht * 2 + sp) * 1024 + lane * 16, lane = 32 half + row; element j = W1[32 ht + row][16 ks + 8 half + j] 2^sw1
// (zero past J)
#define FB_W1_KS 4096
__host__ __device__ inline int64_t fb_w1img_floats(int J) { return (int64_t)((J + 15) / 16) * (FB_W1_KS / 4); }
__device__ __forceinline__ void pack_w1_b_kstep(int ks, int J, const float* __restrict__ W1, float w1_scale, uint8_t* __restrict__ w1img) {
    uint8_t* out = w1img + (int64_t)ks * FB_W1_KS;
    if (blockDim.x != 256) {                                           // (any other block size: element by element)
        for (int e = threadIdx.x; e < 2 * 64 * 8; e += blockDim.x) {
            const int j = e & 7, lane = (e >> 3) & 63, ht = e >> 9;
            const int half = lane >> 5, hh = 32 * ht + (lane & 31);
            const int it = 16 * ks + 8 * half + j;
            const float w = it < J ? w1_scale * W1[(int64_t)hh * J + it] : 0.f;
            uint16_t* o = (uint16_t*)(out + (ht * 2) * 1024 + lane * 16) + j;
            split2h_bits(w, o[0], o[512]);
        }
        return;
    }
    float v[4];                                                        // (256 threads: the loads of all four trips first)
#pragma unroll
    for (int i = 0; i < 4; ++i) {                                      // (ht, lane, j)
        const int e = threadIdx.x + 256 * i;
        const int j = e & 7, lane = (e >> 3) & 63, ht = e >> 9;
        const int half = lane >> 5, hh = 32 * ht + (lane & 31);
        const int it = 16 * ks + 8 * half + j;
        v[i] = it < J ? w1_scale * W1[(int64_t)hh * J + it] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = threadIdx.x + 256 * i;
        const int j = e & 7, lane = (e >> 3) & 63, ht = e >> 9;
        uint16_t* o = (uint16_t*)(out + (ht * 2) * 1024 + lane * 16) + j;
        split2h_bits(v[i], o[0], o[512]);
    }
}
__global__ void k_pack_w1_b(int J, const float* __restrict__ W1, const float* __restrict__ sc, uint8_t* __restrict__ w1img) {
    pack_w1_b_kstep(blockIdx.x, J, W1, sc[0], w1img);
}

// the response bytes of one B fragment (items 16 ks + 8 half + 0..7 of a person) as fp16: byte b in {0, 1, 255} ->
// {0, 1, -1} = (b & 1) * 0x3C00 | (b & 0x80) << 8, two bytes per dword
__device__ __forceinline__ f16x8 fb_y_frag(const uint32_t* yw) {
    typedef uint32_t u32x4y __attribute__((ext_vector_type(4)));
    const uint32_t w0 = yw[0], w1 = yw[1];
    u32x4y q;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const uint32_t src = (d >> 1) ? w1 : w0;
        const uint32_t t = (d & 1) ? __builtin_amdgcn_perm(0u, src, 0x0c030c02u) : __builtin_amdgcn_perm(0u, src, 0x0c010c00u);
        q[d] = (t & 0x00010001u) * 0x3C00u | ((t & 0x00800080u) << 8);
    }
    return __builtin_bit_cast(f16x8, q);
}

// eight fp32 values -> three bf16 fragments by truncation: v = hi + mid + lo exactly (8 + 8 + 8 significand bits); the
// fc1 weight-gradient kernel (k_fc1_bwd_b.hip) splits ghpre with it
__device__ __forceinline__ void fb_split8(const float* v, bf16x8& fh, bf16x8& fm, bf16x8& fl) {
    typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
    u32x4v ph, pm, pl;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t hb[2], mb[2], lb[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float x = v[2 * q + e];
            hb[e] = __builtin_bit_cast(uint32_t, x) & 0xffff0000u;
            const float r1 = x - __builtin_bit_cast(float, hb[e]);
            mb[e] = __builtin_bit_cast(uint32_t, r1) & 0xffff0000u;
            const float r2 = r1 - __builtin_bit_cast(float, mb[e]);
            lb[e] = __builtin_bit_cast(uint32_t, r2) & 0xffff0000u;
        }
        ph[q] = hb[1] | (hb[0] >> 16);
        pm[q] = mb[1] | (mb[0] >> 16);
        pl[q] = lb[1] | (lb[0] >> 16);
    }
    fh = __builtin_bit_cast(bf16x8, ph);
    fm = __builtin_bit_cast(bf16x8, pm);
    fl = __builtin_bit_cast(bf16x8, pl);
}

// SPLIT (small batches: at most FB_SPLIT_MAX persons): a workgroup takes ONE 32-person tile and its four waves share the
// 176 head tiles (quarter ranges of the OFF tiles, one 32-row block of the DIAG / LOC sections each) instead of each
// walking all of them for its own persons -- the per-wave chain of a step is what a small batch waits for.  Every wave
// makes fc1 for the same persons (it needs h in registers); wave 0 owns the outputs and the eps / x tile in LDS; a k
// whose rows straddle two ranges gets its two partial sums by commutative LDS float adds (two addends: order-free).
#define FB_SPLIT_MAX 8192                                             // one round of workgroups on 256 CUs; beyond it the plain form wins (measured at 10 000)
template <bool SPLIT>
__global__ __launch_bounds__(FB_THREADS, 1) void k_mvn_enc_fwd_b(
    EncDims dm, const uint8_t* __restrict__ y, const int64_t* __restrict__ rows, int64_t gid0,
    const uint8_t* __restrict__ w1img, const float* __restrict__ b1, const uint8_t* __restrict__ img,
    const uint32_t* __restrict__ gt2, const float* __restrict__ sc /*k_enc_scales*/, const float* __restrict__ eps_in, uint64_t seed,
    uint32_t step, const uint32_t* __restrict__ step_dev, uint32_t stream, float* __restrict__ h_out,
    float* __restrict__ x_out, float* __restrict__ eps_out, float* __restrict__ ldT, float* __restrict__ ent_out,
    float* __restrict__ hT_out /*[64][nb] or null*/, float* __restrict__ epsT_out /*[D][nb] or null*/,
    uint8_t* __restrict__ ximg_out /*f16x2 tile images of x for k_irt_lik_h (k_irt_lik_h.hip), or null*/,
    uint16_t* __restrict__ hs_out /*[2][64][nb] fp16 terms of h 2^sh for k_mvn_enc_bwd_w_b (what k_split2_f16 makes), or null*/,
    int64_t i_base = 0 /*first person of this launch (a multiple of 64): the persons before it belong to another launch*/) {
    if (step_dev) step = *step_dev;                              // captured step: the Philox step lives in device memory
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef uint32_t u32x4w __attribute__((ext_vector_type(4)));
    constexpr int H = 64;
    const int D = dm.D, J = dm.J;
    const int DS = pk_dse(D), DX = (D + 3) & ~3;
    const int YS = ef_ys(J);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    float* R1 = smem + wave * fb_wave_floats(D, J);
    int8_t* Yi = (int8_t*)R1;                                 // phase A
    float* S1 = SPLIT ? smem : R1;                            // SPLIT: the tile of wave 0 serves the workgroup
    float* eps_lds = S1;                                      // phase B  [32][DS]
    float* x_lds = S1 + FB_WP * DS;                           //          [32][DX]
    uint32_t* gt_lds = (uint32_t*)(smem + FB_WAVES * fb_wave_floats(D, J));
    float* ent_s = (float*)(gt_lds + ((pk_off_total(D) / 8 + 4) / 4 * 4));   // SPLIT: [4][32] partial entropy sums
    const int64_t i0 = i_base + (SPLIT ? (int64_t)blockIdx.x * FB_WP : ((int64_t)blockIdx.x * FB_WAVES + wave) * FB_WP);
    const int p = l31;
    const int64_t i = i0 + p;
    // NOTE: no early exit -- every wave takes part in the workgroup barrier that publishes the group table; waves (and
    // lanes) past the last person compute on clamped inputs and store nothing.
    const bool wave_live = i0 < dm.nb;
    // diagnostic build (tools/fwd2_bench.hip -DFB_STAMPS): cycles per phase of one workgroup, printed at the end
#ifdef FB_STAMPS
    uint64_t fst_[10]; int fsn_ = 0;
#define FSTAMP() do { __builtin_amdgcn_s_waitcnt(0); fst_[fsn_++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FSTAMP() do {} while (0)
#endif
    FSTAMP();

    const int n_off = pk_off_total(D) / 32;                   // multiple of 6
    const int n_sec = pk_sec(D) / 32;
    const int t_end = n_off + 2 * n_sec;
    // the OFF group table: fetched NOW into registers, written to LDS behind the response staging (published by the barrier
    // before the OFF loop) -- a load + LDS store here made the kernel wait one memory latency before it asked for anything else
    constexpr int FB_GQ = 5;                                  // 4 n_off <= 1 080 words (D <= 128) over 256 threads
    uint32_t gtv[FB_GQ];
#pragma unroll
    for (int q = 0; q < FB_GQ; ++q) { const int e = tid + q * FB_THREADS; gtv[q] = e < 4 * n_off ? gt2[e] : 0u; }
    const float w1_inv = sc[1], h_scale = sc[3], acc_inv = sc[4];

    // ---------------------------------------------------------------- stage this wave's response rows (bytes)
    const int n_ydma = (32 * J + 1023) / 1024;
    const bool ydense = !rows && ((J >> 2) & 1) && i0 + FB_WP <= dm.nb && (i0 * J + (int64_t)n_ydma * 1024 <= dm.nb * (int64_t)J) &&
                        (size_t)n_ydma * 1024 <= fb_wave_floats(D, J) * sizeof(float);
    const int ysr = ydense ? J : YS;                          // LDS row stride of the response bytes
    // the normals of this wave's 32 persons (D / 4 Philox blocks each) are drawn into registers while the response
    // rows are in flight: the draw needs no memory, the DMA latency at the head of the workgroup is otherwise exposed
    constexpr int FB_EQ = 16;                                 // 32 * (D / 4) / 64 <= 16 for D <= 128
    f32x4 zq[FB_EQ];
    const int nblk = D >> 2;                                  // D % 4 == 0 on this path
    // (SPLIT: the four waves have the same 32 persons and each draws a quarter of the slots, q = wave mod 4: drawn by every
    // wave in full the normals were 45 k of the kernel's 141 k cycles at B = 100, tools/fwd2_bench.hip -DFB_STAMPS)
    auto draw_eps = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < FB_EQ; ++q) {
            const int e = lane + 64 * q;
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            if ((!SPLIT || (q & (FB_WAVES - 1)) == wave) && e < FB_WP * nblk) {
                const int pp = e / nblk, blk = e - pp * nblk;
                int64_t ii = i0 + pp;
                if (ii >= dm.nb) ii = dm.nb - 1;              // absent persons: any finite values, never stored
                if (eps_in) {
                    z = *(const f32x4*)(eps_in + ii * D + 4 * blk);
                } else {
                    const int64_t row = rows ? rows[ii] : ii;
                    z = philox_normal4(seed, step, stream, gid0 + row, (uint32_t)blk);
                }
            }
            zq[q] = z;
        }
    };
    if (ydense) {
        const uint8_t* src = y + i0 * J + 16 * lane;
        const uint32_t lb = lds_addr_uniform(R1);
        for (int d = 0; d < n_ydma; ++d) dma16(src + d * 1024, lb + (uint32_t)d * 1024u);
        draw_eps();
        vx_wait_vmem();
    } else {
        // gathered rows, person by person: the row's byte offset is wave-uniform (read once per lane = person, handed out by
        // v_readlane), so a load is a scalar base plus 4 * lane -- no per-element (person, word) arithmetic, which with its two
        // integer divisions was 33 k of this phase's 50 k cycles at B = 100 (tools/fwd2_bench.hip -DFB_STAMPS; the loads
        // themselves were not: all of a batch in flight made no difference).  16 persons a batch, the normals drawn under
        // the first batch's latency.
        constexpr int PB = 16, MAXLD = 4;                       // J <= 1024: at most four 64-word pieces a row
        const int YW = YS / 4, JW = J / 4;
        const int nld = (JW + 63) >> 6, nch = (YW + 63) >> 6;   // pieces with response words | pieces of the LDS row (padding)
        uint32_t* Yw = (uint32_t*)R1;
        int64_t rowb = 0;
        {
            const int64_t ii = i0 + l31;
            if (ii < dm.nb) rowb = (rows ? rows[ii] : ii) * J;
        }
        const int rowb_lo = (int)(uint32_t)rowb, rowb_hi = (int)(uint32_t)((uint64_t)rowb >> 32);
        if constexpr (SPLIT) {
            // Round 5: the four waves DIVIDE fc1 (k-steps [ks_lo, ks_hi) each, partial sums joined below), so a wave stages the
            // response words of ITS items only -- 4 (ks_hi - ks_lo) words a person, two persons a load instruction -- instead of all
            // four staging all 32 rows whole (22 k of the workgroup's 92 k cycles at B = 100, tools/fwd2_bench.hip -DFB_STAMPS).
            const int n_ks_all = (J + 15) / 16, ks_per = (n_ks_all + FB_WAVES - 1) / FB_WAVES;
            const int w_lo = 4 * ks_per * wave;                    // first word of this wave's items; 4 ks_per words a person
            const int wpp = 4 * ks_per;                            // <= 64 (J <= 1024)
            const int ppl = 64 / wpp > 0 ? 64 / wpp : 1;           // persons a load instruction
            const int pl = lane / wpp, wl = lane - pl * wpp;       // this lane's person within the instruction, word within the range
            draw_eps();                                            // (no memory: under the latency of the row offsets)
            for (int p0 = 0; p0 < FB_WP; p0 += ppl * 8) {
                uint32_t v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int pp = p0 + ppl * u + pl;
                    const int ppc = pp < FB_WP ? pp : FB_WP - 1;
                    const uint64_t rb = ((uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute(4 * ppc, rowb_hi) << 32) |
                                        (uint32_t)__builtin_amdgcn_ds_bpermute(4 * ppc, rowb_lo);
                    v[u] = 0u;
                    if (pl < ppl && pp < FB_WP && i0 + pp < dm.nb && w_lo + wl < JW) v[u] = ((const uint32_t*)(y + (int64_t)rb))[w_lo + wl];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int pp = p0 + ppl * u + pl;
                    if (pl < ppl && pp < FB_WP && w_lo + wl < YW) Yw[pp * YW + w_lo + wl] = v[u];
                }
            }
        } else
        for (int p0 = 0; p0 < FB_WP; p0 += PB) {
            uint32_t v[PB][MAXLD];
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const int pp = p0 + u;
                const uint64_t rb = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(rowb_hi, pp) << 32) |
                                    (uint32_t)__builtin_amdgcn_readlane(rowb_lo, pp);
                const uint32_t* src = (const uint32_t*)(y + (int64_t)rb) + lane;
                const bool live = i0 + pp < dm.nb;              // wave-uniform
#pragma unroll
                for (int c = 0; c < MAXLD; ++c) {
                    v[u][c] = 0u;
                    if (c < nld && live && lane + 64 * c < JW) v[u][c] = src[64 * c];   // bytes 0/1/255 == int8 0/1/-1 (vi.py:689-691)
                }
            }
            if (p0 == 0) draw_eps();
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                uint32_t* dst = Yw + (p0 + u) * YW + lane;
#pragma unroll
                for (int c = 0; c < MAXLD + 1; ++c)
                    if (c < nch && lane + 64 * c < YW) dst[64 * c] = c < MAXLD ? v[u][c < MAXLD ? c : 0] : 0u;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < FB_GQ; ++q) { const int e = tid + q * FB_THREADS; if (e < 4 * n_off) gt_lds[e] = gtv[q]; }
    __builtin_amdgcn_wave_barrier();
    FSTAMP();                                                 // 1: responses staged, normals drawn
    // ---------------------------------------------------------------- phase A: fc1 (+ softplus), both hidden tiles
    f16x8 hb[2][4];                                           // [term][k-step]: B fragments of every head tile
    {
        f32x16 hreg[2];
        f32x16 acc0 = zero16(), acc1 = zero16();
        // pre[hh][p] = sum_j W1[hh][j] yin[p][j] on the fp16 MFMA: the response bytes (-1 / 0 / 1) are exact in fp16, so
        // two products per 16-item k-step and hidden tile (W1 2^sw1 hi, lo -- split once per step into w1img by
        // k_pack_w1_b); the four 16-byte fragments of a k-step go global -> registers three k-steps ahead.
        const int n_ks = (J + 15) / 16;
        auto loadA = [&](f16x8 (&Af)[4], int ks) __attribute__((always_inline)) {
            ks = ks < n_ks ? ks : n_ks - 1;                   // past the end: reload the last k-step (never used)
            const uint8_t* src = w1img + (int64_t)ks * FB_W1_KS + lane * 16;
#pragma unroll
            for (int f = 0; f < 4; ++f) Af[f] = *(const f16x8*)(src + f * 1024);
        };
        auto compute = [&](const f16x8 (&Af)[4], int ks) __attribute__((always_inline)) {
            const f16x8 yb = fb_y_frag((const uint32_t*)(Yi + p * ysr + 16 * ks + 8 * half));   // rows are 4-byte aligned
            acc0 = mfma_f16(Af[1], yb, acc0); acc1 = mfma_f16(Af[3], yb, acc1);
            acc0 = mfma_f16(Af[0], yb, acc0); acc1 = mfma_f16(Af[2], yb, acc1);
        };
        if constexpr (SPLIT) {
            // this wave's quarter of the k-steps (its items: what it staged above; the dense staging has every row whole)
            const int ks_per = (n_ks + FB_WAVES - 1) / FB_WAVES;
            const int ks_lo = ks_per * wave, ks_hi = (ks_lo + ks_per < n_ks) ? ks_lo + ks_per : n_ks;
            f16x8 A[4][4];
            loadA(A[0], ks_lo); loadA(A[1], ks_lo + 1); loadA(A[2], ks_lo + 2);
            for (int c = ks_lo; c < ks_hi; c += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    loadA(A[(u + 3) & 3], c + u + 3);
                    if (c + u < ks_hi) compute(A[u], c + u);
                }
            }
            // the four partial pre-activations meet in LDS -- every wave needs all of h in registers -- and are added in the
            // fixed order of the waves: [wave][32 registers][64 lanes] floats, each wave's slab in its own region (the response
            // bytes there have been consumed)
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 16; ++r) { R1[r * 64 + lane] = acc0[r]; R1[(16 + r) * 64 + lane] = acc1[r]; }
            __syncthreads();
            const size_t wstride = fb_wave_floats(D, J);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float a0 = smem[r * 64 + lane], a1 = smem[(16 + r) * 64 + lane];
#pragma unroll
                for (int w = 1; w < FB_WAVES; ++w) { a0 += smem[w * wstride + r * 64 + lane]; a1 += smem[w * wstride + (16 + r) * 64 + lane]; }
                acc0[r] = a0; acc1[r] = a1;
            }
            __syncthreads();                                       // (the regions are written again below: eps / x tile of the workgroup)
        } else {
            f16x8 A[4][4];
            loadA(A[0], 0); loadA(A[1], 1); loadA(A[2], 2);
            for (int c = 0; c < n_ks; c += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    loadA(A[(u + 3) & 3], c + u + 3);
                    if (c + u < n_ks) compute(A[u], c + u);
                }
            }
        }
#pragma unroll
        for (int ht = 0; ht < 2; ++ht) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int hh0 = 32 * ht + 8 * g + 4 * half;
                const float4 bb = *(const float4*)(b1 + hh0);
                float4 hv;
                hv.x = softplusf_(fmaf((ht ? acc1 : acc0)[4 * g + 0], w1_inv, bb.x));   // vi.py:449
                hv.y = softplusf_(fmaf((ht ? acc1 : acc0)[4 * g + 1], w1_inv, bb.y));
                hv.z = softplusf_(fmaf((ht ? acc1 : acc0)[4 * g + 2], w1_inv, bb.z));
                hv.w = softplusf_(fmaf((ht ? acc1 : acc0)[4 * g + 3], w1_inv, bb.w));
                hreg[ht][4 * g + 0] = hv.x; hreg[ht][4 * g + 1] = hv.y;
                hreg[ht][4 * g + 2] = hv.z; hreg[ht][4 * g + 3] = hv.w;
                if ((SPLIT ? wave == ht : true) && i < dm.nb) *(float4*)(h_out + i * H + hh0) = hv;   // SPLIT: wave ht writes tile ht
            }
        }
        // (SPLIT: every wave has h of the same persons: waves 2 and 3 write the two tiles of hT, waves 0 and 1 those of hs)
        if (hT_out && i < dm.nb) {                            // dimension-major copy for the weight-gradient kernel
#pragma unroll
            for (int ht = 0; ht < 2; ++ht)
                if (SPLIT ? wave == 2 + ht : true)
#pragma unroll
                    for (int r = 0; r < 16; ++r) hT_out[(int64_t)(32 * ht + crow32(r, half)) * dm.nb + i] = hreg[ht][r];
        }
        if (hs_out && i < dm.nb) {                            // ... and the two fp16 terms of h 2^sh (as k_split2_f16)
            const int64_t plane = (int64_t)64 * dm.nb;
#pragma unroll
            for (int ht = 0; ht < 2; ++ht)
                if (SPLIT ? wave == ht : true)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int64_t o = (int64_t)(32 * ht + crow32(r, half)) * dm.nb + i;
                        split2h_bits(hreg[ht][r] * h_scale, hs_out[o], hs_out[plane + o]);
                    }
        }
        // k-step s of the head GEMM takes accumulator registers 8 (s & 1) .. + 7 of hidden tile s >> 1
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = hreg[s >> 1][8 * (s & 1) + j];
            split2h_frag(v, h_scale, hb[0][s], hb[1][s]);
        }
    }
    __builtin_amdgcn_wave_barrier();                          // response bytes no longer needed
    FSTAMP();                                                 // 2: fc1, h outputs
    // ---------------------------------------------------------------- eps (zero padded to DS), x := 0
    if constexpr (SPLIT) {
        __syncthreads();                                       // wave 0 has read its response bytes: its region takes the tile
        // x := 0 and the padding columns of eps (disjoint from the values written below: no barrier between them)
        for (int e = tid; e < FB_WP * DX / 4; e += FB_THREADS) *(f32x4*)(x_lds + 4 * e) = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int e = tid; e < FB_WP * (DS - D) / 4; e += FB_THREADS) {
            const int pp = e / ((DS - D) / 4), c = e - pp * ((DS - D) / 4);
            *(f32x4*)(eps_lds + pp * DS + D + 4 * c) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int q = 0; q < FB_EQ; ++q) {
            const int e = lane + 64 * q;
            if ((q & (FB_WAVES - 1)) == wave && e < FB_WP * nblk) {       // the slots this wave drew
                const int pp = e / nblk, blk = e - pp * nblk;
                *(f32x4*)(eps_lds + pp * DS + 4 * blk) = zq[q];
                if (i0 + pp < dm.nb) *(f32x4*)(eps_out + (i0 + pp) * D + 4 * blk) = zq[q];
            }
        }
    } else {
        for (int e = lane; e < FB_WP * (DS + DX) / 4; e += 64) *(f32x4*)(S1 + 4 * e) = f32x4{0.f, 0.f, 0.f, 0.f};   // DS, DX % 4 == 0
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < FB_EQ; ++q) {
            const int e = lane + 64 * q;
            if (e < FB_WP * nblk) {
                const int pp = e / nblk, blk = e - pp * nblk;
                *(f32x4*)(eps_lds + pp * DS + 4 * blk) = zq[q];
                if (i0 + pp < dm.nb) *(f32x4*)(eps_out + (i0 + pp) * D + 4 * blk) = zq[q];
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (epsT_out && i < dm.nb) {                          // dimension-major copy: 128-byte rows per half-wave
#pragma unroll 4
            for (int k = half; k < D; k += 2) epsT_out[(int64_t)k * dm.nb + i] = eps_lds[p * DS + k];
        }
    }
    // ---------------------------------------------------------------- phase B: packed head rows, 32 per tile
    float ent_acc = 0.f;
    const float* ep = eps_lds + p * DS;
    float* xp = x_lds + p * DX;
    struct TileRegs { f16x8 a[2][4]; f16x8 bias; };
    // the fragments of tile t come from the image into registers one tile ahead
    auto pull = [&](TileRegs& R, int t) __attribute__((always_inline)) {
        const int tc = t < t_end ? t : t_end - 1;              // past the end: a harmless duplicate of the last tile
        const uint8_t* gb = img + (int64_t)tc * FB_IMG_BYTES + lane * 16;
        R.bias = *(const f16x8*)(gb + FB_A_BYTES);
#pragma unroll
        for (int sp = 1; sp >= 0; --sp)
#pragma unroll
            for (int s = 0; s < 4; ++s) R.a[sp][s] = *(const f16x8*)(gb + (sp * 4 + s) * 1024);
    };
    // the bias product's constant 2^(sw + sh - sb) in every element (the bias fragment is zero past its two terms)
    f16x8 cfrag;
    {
        const _Float16 c16 = (_Float16)sc[6];
#pragma unroll
        for (int j = 0; j < 8; ++j) cfrag[j] = c16;
    }
    // the chain starts from the bias (one MFMA against the constant); products in order of increasing magnitude
    auto mma_all = [&](const TileRegs& R) __attribute__((always_inline)) -> f32x16 {
        f32x16 a = mfma_f16(R.bias, cfrag, zero16());
#pragma unroll
        for (int s = 0; s < 4; ++s) a = mfma_f16(R.a[1][s], hb[0][s], a);
#pragma unroll
        for (int s = 0; s < 4; ++s) a = mfma_f16(R.a[0][s], hb[1][s], a);
#pragma unroll
        for (int s = 0; s < 4; ++s) a = mfma_f16(R.a[0][s], hb[0][s], a);
        return a;
    };
    // ---- OFF section: x[p][k] += sum_l M[p,(k,l)] eps[p,l]; the partial sum of the current k stays in a register and
    // is stored when the group table says the k ends (4 group words per tile from a copy of the table in LDS).
    // One wave per SIMD issues one instruction every ~4 cycles whatever its kind, so the loop is written for a small
    // instruction count per tile (~150 beside the 25 MFMAs), not only for few vector instructions.
    float cur_part = 0.f;
    const char* ep_h = (const char*)(ep + 4 * half);
    char* xp_b = (char*)xp;
    struct EpiOps { float4 e4[4]; };
    auto epi_read = [&](EpiOps& E, const uint4& c) __attribute__((always_inline)) {
        E.e4[0] = *(const float4*)(ep_h + (c.x & 0xFFFu));
        E.e4[1] = *(const float4*)(ep_h + (c.y & 0xFFFu));
        E.e4[2] = *(const float4*)(ep_h + (c.z & 0xFFFu));
        E.e4[3] = *(const float4*)(ep_h + (c.w & 0xFFFu));
    };
    auto epi_group = [&](const f32x16& a, const float4& e, uint32_t code, int g) __attribute__((always_inline)) {
        // rows (k, l0 + 4half + j) live in a[4g + j]
        cur_part = fmaf(a[4 * g + 0], e.x, cur_part);
        cur_part = fmaf(a[4 * g + 1], e.y, cur_part);
        cur_part = fmaf(a[4 * g + 2], e.z, cur_part);
        cur_part = fmaf(a[4 * g + 3], e.w, cur_part);
        if (__builtin_expect((int)code < 0, 0)) {                                 // wave-uniform, rare: the k ends here
            const float tot = half_sum32(cur_part) * acc_inv;                     // the powers of two come off once per k
            if (half == 0) {
                if (SPLIT) atomicAdd((float*)(xp_b + ((code >> 12) & 0xFFFu)), tot);   // two ranges may share this k
                else *(float*)(xp_b + ((code >> 12) & 0xFFFu)) = tot;
            }
            cur_part = 0.f;
        }
    };

    FSTAMP();                                                 // 3: eps tile, epsT
    __syncthreads();                                           // the group table in LDS is complete
    FSTAMP();                                                 // 4: barrier
    if constexpr (SPLIT) {                                     // dimension-major copy of eps, the rows shared out over the waves
        if (epsT_out && i < dm.nb) {
#pragma unroll 4
            for (int k = 2 * wave + half; k < D; k += 2 * FB_WAVES) epsT_out[(int64_t)k * dm.nb + i] = eps_lds[p * DS + k];
        }
    }
    // OFF tiles of this wave: all of them, or (SPLIT) a quarter -- even counts, the last range takes the remainder
    int t_lo = 0, t_hi = n_off;
    if (SPLIT) {
        const int nr = (n_off / 2 < FB_WAVES) ? n_off / 2 : FB_WAVES;      // ranges in use (few tiles: fewer waves)
        const int qn = (n_off / nr) & ~1;
        t_lo = wave < nr ? wave * qn : n_off;
        t_hi = (wave >= nr - 1) ? n_off : t_lo + qn;
    }
    TileRegs RA, RB;
    pull(RA, t_lo);
    f32x16 accP = zero16();                                    // accumulator of the tile before the current one
    uint4 codeP = make_uint4(0, 0, 0, 0);                      // its group words

    auto off_iter = [&](TileRegs& Rc, TileRegs& Rn, int t, auto firstc) __attribute__((always_inline)) {
        constexpr bool first = decltype(firstc)::value;
        EpiOps E;
        if constexpr (!first) epi_read(E, codeP);
        pull(Rn, t + 1);
        // the chain starts from the bias (one MFMA against the constant); products in order of increasing magnitude; the
        // epilogue of the previous tile goes between its parts
        f32x16 a = mfma_f16(Rc.bias, cfrag, zero16());
        a = mfma_f16(Rc.a[1][0], hb[0][0], a);
        a = mfma_f16(Rc.a[1][1], hb[0][1], a);
        if constexpr (!first) epi_group(accP, E.e4[0], codeP.x, 0);
        a = mfma_f16(Rc.a[1][2], hb[0][2], a);
        a = mfma_f16(Rc.a[1][3], hb[0][3], a);
        a = mfma_f16(Rc.a[0][0], hb[1][0], a);
        if constexpr (!first) epi_group(accP, E.e4[1], codeP.y, 1);
        a = mfma_f16(Rc.a[0][1], hb[1][1], a);
        a = mfma_f16(Rc.a[0][2], hb[1][2], a);
        a = mfma_f16(Rc.a[0][3], hb[1][3], a);
        if constexpr (!first) epi_group(accP, E.e4[2], codeP.z, 2);
        a = mfma_f16(Rc.a[0][0], hb[0][0], a);
        a = mfma_f16(Rc.a[0][1], hb[0][1], a);
        if constexpr (!first) epi_group(accP, E.e4[3], codeP.w, 3);
        a = mfma_f16(Rc.a[0][2], hb[0][2], a);
        a = mfma_f16(Rc.a[0][3], hb[0][3], a);
        accP = a;
        // the group words of this tile, for its epilogue in the next iteration, from the table in LDS (a scalar load
        // would share lgkmcnt with the LDS reads and return out of order: while one is outstanding every LDS wait
        // becomes lgkmcnt(0))
        const uint4 cv = *(const uint4*)(gt_lds + 4 * t);
        codeP.x = __builtin_amdgcn_readfirstlane(cv.x); codeP.y = __builtin_amdgcn_readfirstlane(cv.y);
        codeP.z = __builtin_amdgcn_readfirstlane(cv.z); codeP.w = __builtin_amdgcn_readfirstlane(cv.w);
    };
    if (!SPLIT || t_lo < t_hi) {                               // wave-uniform
        off_iter(RA, RB, t_lo, std::true_type{});
        off_iter(RB, RA, t_lo + 1, std::false_type{});
        for (int t = t_lo + 2; t < t_hi; t += 2) {
            off_iter(RA, RB, t, std::false_type{});
            off_iter(RB, RA, t + 1, std::false_type{});
        }
        {
            EpiOps E;
            epi_read(E, codeP);
            epi_group(accP, E.e4[0], codeP.x, 0);
            epi_group(accP, E.e4[1], codeP.y, 1);
            epi_group(accP, E.e4[2], codeP.z, 2);
            epi_group(accP, E.e4[3], codeP.w, 3);
        }
        if (SPLIT) {
            // the range may end inside a k: its partial sum goes to x now (the range that finishes the k adds the rest)
            const float tot = half_sum32(cur_part) * acc_inv;
            if (half == 0) atomicAdd((float*)(xp_b + ((codeP.w >> 12) & 0xFFFu)), tot);
            cur_part = 0.f;
        }
    }
    FSTAMP();                                                 // 5: OFF tiles
    if (SPLIT) __syncthreads();                                // every OFF row is in x: the sections update it in place
    FSTAMP();                                                 // 6: barrier
    // ---- DIAG section (exp(M_kk) eps_k, entropy, ldT) and LOC section (the loc head): 2 * n_sec tiles; RA holds the
    // first of them.  The 16 x entries a lane updates are read together, updated and written together.
    auto tile_sec = [&](const f32x16& a, int t2) __attribute__((always_inline)) {
        const bool is_diag = t2 < n_off + n_sec;
        const int k0 = 32 * (t2 - (is_diag ? n_off : n_off + n_sec));
        // accumulator registers 4g .. 4g + 3 of a lane are the four consecutive k = k0 + 8g + 4half + 0..3: 16-byte LDS
        // accesses; D % 4 == 0, so a group of four is inside D or outside it as a whole (wave-uniform per lane half)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int kk = k0 + 8 * g + 4 * half;
            if (kk < D) {
                f32x4 xo = *(const f32x4*)(xp + kk);
                if (is_diag) {                                 // exp(diag M) eps_k, entropy, ldT  (vi.py:686)
                    const f32x4 ev = *(const f32x4*)(ep + kk);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float mkk = a[4 * g + j] * acc_inv;
                        const float ld = __expf(mkk);
                        xo[j] = fmaf(ld, ev[j], xo[j]);
                        ent_acc += mkk;
                        if (i < dm.nb) ldT[(int64_t)(kk + j) * dm.nb + i] = ld;
                    }
                } else {                                       // loc head (vi.py:450)
#pragma unroll
                    for (int j = 0; j < 4; ++j) xo[j] = fmaf(a[4 * g + j], acc_inv, xo[j]);
                }
                *(f32x4*)(xp + kk) = xo;
            }
        }
    };
    auto sec_iter = [&](TileRegs& Rc, TileRegs& Rn, int t) __attribute__((always_inline)) {
        if (t + 1 < t_end) pull(Rn, t + 1);
        const f32x16 a = mma_all(Rc);
        tile_sec(a, t);
    };
    if (SPLIT) {
        // wave w takes the 32-row blocks w, w + 4, .. of both sections: the DIAG and the LOC tile of a block update the
        // same 32 entries of x, no other wave touches them
        for (int ts = wave; ts < n_sec; ts += FB_WAVES) {
            pull(RA, n_off + ts);
            tile_sec(mma_all(RA), n_off + ts);
            pull(RA, n_off + n_sec + ts);
            tile_sec(mma_all(RA), n_off + n_sec + ts);
        }
        const float ea = ent_acc + __shfl_xor(ent_acc, 32, 64);
        if (half == 0) ent_s[wave * FB_WP + p] = ea;
        __syncthreads();                                       // x and the entropy parts are complete
        ent_acc = (ent_s[p] + ent_s[FB_WP + p]) + (ent_s[2 * FB_WP + p] + ent_s[3 * FB_WP + p]);
    } else {
        for (int t = n_off; t < t_end; t += 2) {               // n_off and t_end are even
            sec_iter(RA, RB, t);
            sec_iter(RB, RA, t + 1);
        }
    }
    vx_wait_vmem();
    __builtin_amdgcn_wave_barrier();
    FSTAMP();                                                 // 7: sections
    // ---------------------------------------------------------------- write x, entropy part
    // (SPLIT: the tile is the workgroup's and complete: all four waves write the outputs)
    const int e_lo = SPLIT ? tid : lane, e_st = SPLIT ? FB_THREADS : 64;
    if (ximg_out && i0 < (dm.nb + 63) / 64 * 64) {
        // the likelihood kernel's operand: x_aug = [x, 1, 0..] 2^LH_XEXP as two fp16 terms (k_irt_lik_h.hip), in its LDS tile order (lb_xoff): this
        // wave's 32 persons are one half of a 64-person tile (absent persons: all-zero rows); 14 chunks of 8 columns each
        const int pvi = (int)((dm.nb - i0) < FB_WP ? (dm.nb - i0) : FB_WP);       // may be <= 0
        uint8_t* out = ximg_out + (i0 >> 6) * LH_XT_BYTES;
        const int pbase = (int)(i0 & 63);
        for (int e = e_lo; e < FB_WP * 2 * LB_NKS; e += e_st) {
            // order of the image bytes: 32 consecutive lanes fill one 512-byte subtile (8 persons x 4 chunks), then the
            // 256-byte half subtiles (8 persons x 2 chunks): whole contiguous runs per store instruction
            int pp, ch;
            if (e < 4 * 3 * 32) {
                const int blk = e >> 5, r = e & 31;                   // blk = person group * 3 + subtile
                pp = 8 * (blk / 3) + (r >> 2);
                ch = 4 * (blk % 3) + (r & 3);
            } else {
                const int r = e - 4 * 3 * 32;                         // 4 person groups x 8 persons x 2 chunks
                pp = 8 * (r >> 4) + ((r >> 1) & 7);
                ch = 12 + (r & 1);
            }
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 8 * ch + j;
                v[j] = (pp < pvi) ? (k < D ? x_lds[pp * DX + k] : (k == D ? 1.0f : 0.f)) : 0.f;
            }
            f16x8 fh, fl;
            if (lh_split_x(v, fh, fl)) atomicOr((uint32_t*)(ximg_out + (dm.nb + 63) / 64 * (int64_t)LH_XT_BYTES), 1u);
            const uint32_t o = lb_xoff(pbase + pp, ch);
            *(f16x8*)(out + o) = fh;
            *(f16x8*)(out + LB_PLANE + o) = fl;
        }
    }
    if (wave_live) {
        const int pv = (int)((dm.nb - i0) < FB_WP ? (dm.nb - i0) : FB_WP);
        const int c4 = D >> 2;
        for (int e = e_lo; e < pv * c4; e += e_st) {
            const int pp = e / c4, c = e - pp * c4;
            *(f32x4*)(x_out + (i0 + pp) * D + 4 * c) = *(const f32x4*)(x_lds + pp * DX + 4 * c);
        }
        if (!SPLIT) ent_acc += __shfl_xor(ent_acc, 32, 64);
        if ((SPLIT ? wave == FB_WAVES - 1 : true) && half == 0 && i < dm.nb) {
            float s = 0.f;
            for (int k = 0; k < D; k += 4) {                   // (D % 4 == 0; the additions in the order k = 0, 1, 2, ..)
                const f32x4 e = *(const f32x4*)(eps_lds + p * DS + k);
                s += e[0] * e[0]; s += e[1] * e[1]; s += e[2] * e[2]; s += e[3] * e[3];
            }
            ent_out[i] = 0.5f * s + ent_acc;                  // -log q + const = 0.5|eps|^2 + sum_k M_kk
        }
    }
#ifdef FB_STAMPS
    FSTAMP();                                                 // 8: x image, x, entropy
    if (blockIdx.x == 1 && lane == 0)
        printf("FSTAMPS blk %d wave %d: y+draw %llu fc1 %llu eps %llu bar %llu off %llu bar %llu sec %llu out %llu total %llu\n", (int)blockIdx.x, wave,
               fst_[1] - fst_[0], fst_[2] - fst_[1], fst_[3] - fst_[2], fst_[4] - fst_[3], fst_[5] - fst_[4], fst_[6] - fst_[5], fst_[7] - fst_[6],
               fst_[8] - fst_[7], fst_[8] - fst_[0]);
#endif
}


// ---------------------------------------------------------------------------------------------
// NormEncoder forward (vi.py:417-435) for hidden_dim == 64, J % 4 == 0 on the bf16 MFMA: the same outputs as
// k_norm_enc_fwd_fast (k_mvn_packed.hip).  The response bytes are exact in bf16, so fc1 is THREE products per 16-item
// k-step (W1 in three bf16 terms).  k_norm_enc_fwd_fast had every wave stream all of W1 from L2 (4 GB a step at 1M x 500:
// the kernel ran at L2 speed, not MFMA speed); here the four waves of a workgroup share one copy: the 256 threads fetch a
// 64 x 16 slice of W1, split it and lay the six fragments down in LDS (double buffered, one barrier per k-step), so W1
// crosses the L2 once per 128 persons.
// ---------------------------------------------------------------------------------------------
#define NB_THREADS 256
#define NB_WAVES 4
__host__ __device__ inline size_t nb_lds_bytes(int J) {
    return (size_t)NB_WAVES * norm_fast_wave_floats(J) * sizeof(float) + 2 * 6 * 1024;      // response tiles | 2 x 6 fragments
}

__global__ __launch_bounds__(NB_THREADS, 2) void k_norm_enc_fwd_b(
    EncDims dm, const uint8_t* __restrict__ y, const int64_t* __restrict__ rows, const float* __restrict__ W1,
    const float* __restrict__ b1, const float* __restrict__ W21, const float* __restrict__ b21,
    const float* __restrict__ W22, const float* __restrict__ b22, float* __restrict__ h_out,
    float* __restrict__ loc_out, float* __restrict__ raw_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef uint32_t u32x4w __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x2w __attribute__((ext_vector_type(2)));
    constexpr int H = 64;
    const int J = dm.J, YS = ef_ys(J);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    float* R1 = smem + wave * norm_fast_wave_floats(J);
    const int8_t* Yi = (const int8_t*)R1;
    char* frag = (char*)(smem + NB_WAVES * norm_fast_wave_floats(J));      // [2][6][64 lanes][16 bytes]
    const int64_t i0 = ((int64_t)blockIdx.x * NB_WAVES + wave) * EP_WP;    // waves past the end still serve the barriers
    const int p = l31;
    const int64_t i = i0 + p;
    // ---- this wave's response rows (as in k_norm_enc_fwd_fast)
    const int n_ydma = (32 * J + 1023) / 1024;
    const bool ydense = !rows && ((J >> 2) & 1) && i0 + EP_WP <= dm.nb && (i0 * J + (int64_t)n_ydma * 1024 <= dm.nb * (int64_t)J);
    const int ysr = ydense ? J : YS;
    if (ydense) {
        const uint8_t* src = y + i0 * J + 16 * lane;
        const uint32_t lb = lds_addr_uniform(R1);
        for (int d = 0; d < n_ydma; ++d) dma16(src + d * 1024, lb + (uint32_t)d * 1024u);
    } else {
        // gathered (or ragged) rows, word by word: the person's row index is read ONCE (lane p of the wave) and handed round,
        // so the word loads of a person depend on nothing and many are in flight -- with the index looked up per word
        // (rows[ii], then y[row]: two latencies in a row, 125 rounds a wave at 1000 items) this staging was 90 of the
        // kernel's 97 us at the reference's 100 rows a step (Irt2PL.test_ai's shape; tools/ref_usage_times.py)
        const int YW = YS / 4, JW = J / 4;
        uint32_t* Yw = (uint32_t*)R1;
        const int64_t ip = i0 + l31;
        const int64_t myrow = ip < dm.nb ? (rows ? rows[ip] : ip) : (int64_t)-1;
        const int row_lo = (int)(uint32_t)myrow, row_hi = (int)(uint32_t)((uint64_t)myrow >> 32);
        if (YW >= 64) {                                                   // a row fills the wave: person by person (118.7 us a step
#pragma unroll 4                                                          // at 1000 items against 127.8 the other way round)
            for (int pp = 0; pp < EP_WP; ++pp) {
                const int64_t row = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane(row_hi, pp) << 32) |
                                              (uint32_t)__builtin_amdgcn_readlane(row_lo, pp));
                for (int wq = lane; wq < YW; wq += 64) {
                    uint32_t v = 0u;
                    if (wq < JW && row >= 0) v = *(const uint32_t*)(y + row * J + 4 * wq);   // bytes 0/1/255 == int8 0/1/-1 (vi.py:680-682)
                    Yw[pp * YW + wq] = v;
                }
            }
        } else {                                                          // short rows: the lanes cover several persons at once
#pragma unroll 4
            for (int e = lane; e < EP_WP * YW; e += 64) {
                const int pp = e / YW, wq = e - pp * YW;
                const int64_t row = (int64_t)(((uint64_t)(uint32_t)__shfl(row_hi, pp, 64) << 32) | (uint32_t)__shfl(row_lo, pp, 64));
                uint32_t v = 0u;
                if (wq < JW && row >= 0) v = *(const uint32_t*)(y + row * J + 4 * wq);
                Yw[e] = v;
            }
        }
    }
    // ---- W1 slices: thread = (hidden unit hh, quarter q of the 16 items of a k-step)
    const int n_ks = (J + 15) / 16;
    const int hh = tid >> 2, q = tid & 3;
    const float* wrow = W1 + (int64_t)hh * J + 4 * q;
    auto fetch_w = [&](int ks) -> float4 {
        const int j0 = 16 * ks + 4 * q;
        return (ks < n_ks && j0 + 4 <= J) ? *(const float4*)(wrow + 16 * ks) : make_float4(0.f, 0.f, 0.f, 0.f);   // J % 4 == 0
    };
    // fragment (hidden tile ht, term sp): lane (row = hh & 31, half) holds W1[32 ht + row][16 ks + 8 half + j], j = 0..7;
    // this thread owns j = 4 (q & 1) .. + 3 of lane 32 (q >> 1) + (hh & 31): 8 bytes per term
    char* wdst = frag + ((hh >> 5) * 3) * 1024 + (32 * (q >> 1) + (hh & 31)) * 16 + 8 * (q & 1);
    auto put_w = [&](const float4& w, int buf) {
        const float v[4] = {w.x, w.y, w.z, w.w};
        uint16_t t[3][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const __bf16 a = (__bf16)v[e];
            const float r1 = v[e] - (float)a;
            const __bf16 m = (__bf16)r1;
            const __bf16 l = (__bf16)(r1 - (float)m);
            t[0][e] = __builtin_bit_cast(uint16_t, a); t[1][e] = __builtin_bit_cast(uint16_t, m); t[2][e] = __builtin_bit_cast(uint16_t, l);
        }
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) {
            const u32x2w pk = {(uint32_t)t[sp][0] | ((uint32_t)t[sp][1] << 16), (uint32_t)t[sp][2] | ((uint32_t)t[sp][3] << 16)};
            *(u32x2w*)(wdst + buf * 6144 + sp * 1024) = pk;
        }
    };
    put_w(fetch_w(0), 0);
    // W1 slices NB_PF k-steps ahead, in registers: a minibatch is ONE workgroup on one CU, and with the slice of k-step ks + 1
    // requested only an iteration before it is laid down every iteration waited for that load (the reference's Irt2PL.test_ai
    // shape, 1000 items x 100 rows: 63 k-steps of 1.6 us = 100 us of a 146 us step; four ahead: tools/ref_usage_times.py)
    constexpr int NB_PF = 4;
    float4 wq[NB_PF];
#pragma unroll
    for (int u = 0; u < NB_PF; ++u) wq[u] = fetch_w(1 + u);
    vx_wait_vmem();                                                       // the response rows (DMA) have landed
    __syncthreads();
    f32x16 acc0 = zero16(), acc1 = zero16();
    for (int ks0 = 0; ks0 < n_ks; ks0 += NB_PF) {
#pragma unroll
      for (int u = 0; u < NB_PF; ++u) {
        const int ks = ks0 + u;
        if (ks >= n_ks) break;                                            // (uniform)
        const int buf = ks & 1;
        const float4 wcur = wq[u];                                        // slice ks + 1, requested NB_PF iterations ago
        wq[u] = fetch_w(ks + 1 + NB_PF);
        const char* fb = frag + buf * 6144 + lane * 16;
        bf16x8 A[6];
#pragma unroll
        for (int f = 0; f < 6; ++f) A[f] = *(const bf16x8*)(fb + f * 1024);
        // B fragment: items 16 ks + 8 half + 0..7 of person p; byte b in {0, 1, 255} -> bf16 {0, 1, -1}
        const uint32_t* yw = (const uint32_t*)(Yi + p * ysr + 16 * ks + 8 * half);
        const u32x2w w = {yw[0], yw[1]};
        u32x4w qv;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const uint32_t src = w[d >> 1];
            const uint32_t t = (d & 1) ? __builtin_amdgcn_perm(0u, src, 0x0c030c02u) : __builtin_amdgcn_perm(0u, src, 0x0c010c00u);
            qv[d] = (t & 0x00010001u) * 0x3F80u | ((t & 0x00800080u) << 8);
        }
        const bf16x8 yb = __builtin_bit_cast(bf16x8, qv);
        acc0 = mfma_bf16(A[2], yb, acc0); acc1 = mfma_bf16(A[5], yb, acc1);
        acc0 = mfma_bf16(A[1], yb, acc0); acc1 = mfma_bf16(A[4], yb, acc1);
        acc0 = mfma_bf16(A[0], yb, acc0); acc1 = mfma_bf16(A[3], yb, acc1);
        if (ks + 1 < n_ks) put_w(wcur, buf ^ 1);                          // the other buffer: read last in iteration ks - 1
        __syncthreads();
      }
    }
    if (i0 >= dm.nb) return;
    // ---- softplus, the two 1-row heads as per-lane dot products over the 32 hidden units a lane holds
    float sl = 0.f, sr = 0.f;
#pragma unroll
    for (int ht = 0; ht < 2; ++ht)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int hh0 = 32 * ht + 8 * g + 4 * half;
            const float4 bb = *(const float4*)(b1 + hh0);
            const float4 w21 = make_float4(W21[hh0], W21[hh0 + 1], W21[hh0 + 2], W21[hh0 + 3]);
            const float4 w22 = make_float4(W22[hh0], W22[hh0 + 1], W22[hh0 + 2], W22[hh0 + 3]);
            float4 hv;
            hv.x = softplusf_((ht ? acc1 : acc0)[4 * g + 0] + bb.x);               // vi.py:432
            hv.y = softplusf_((ht ? acc1 : acc0)[4 * g + 1] + bb.y);
            hv.z = softplusf_((ht ? acc1 : acc0)[4 * g + 2] + bb.z);
            hv.w = softplusf_((ht ? acc1 : acc0)[4 * g + 3] + bb.w);
            sl += hv.x * w21.x + hv.y * w21.y + hv.z * w21.z + hv.w * w21.w;
            sr += hv.x * w22.x + hv.y * w22.y + hv.z * w22.z + hv.w * w22.w;
            if (i < dm.nb) *(float4*)(h_out + i * H + hh0) = hv;
        }
    sl = half_sum32(sl);
    sr = half_sum32(sr);
    if (half == 0 && i < dm.nb) {
        loc_out[i] = sl + b21[0];
        raw_out[i] = sr + b22[0];
    }
}


// ---------------------------------------------------------------------------------------------
// The same forward with fc1 as the multivariate guide runs it (round 5): W1 2^sw1 split ONCE a step into two fp16 terms, laid
// down as ready k-step fragments (pack_w1_b_kstep; 4 KB a k-step, L2-resident) that every wave pulls straight into registers
// three k-steps ahead -- FOUR products of 32 cycles per k-step instead of six, no split inside the loop, no LDS round trip
// for the weights and no workgroup barrier per k-step (k_norm_enc_fwd_b above re-splits its W1 slice in every workgroup and
// meets at a barrier 32 times: 0.37 ms at 1M x 500 against an MFMA floor of 0.06).  Two launches make the images; small
// batches keep k_norm_enc_fwd_b (vx_norm_enc_forward).
//   packws: [w1img: n_ks x FB_W1_KS bytes | 16 floats: 2^sw1, 2^-sw1 | NH_MAX_BLOCKS partial maxima]
// ---------------------------------------------------------------------------------------------
#define NH_MAX_BLOCKS 64
__host__ __device__ inline int64_t nh_pack_floats(int J) { return fb_w1img_floats(J) + 16 + NH_MAX_BLOCKS; }

__global__ __launch_bounds__(256) void k_norm_pack_max(int J, const float* __restrict__ W1, float* __restrict__ part) {
    __shared__ float red[4];
    float m = 0.f;
    const int n4 = 64 * J / 4;                                            // (J % 4 == 0, W1 16-byte aligned)
    const f32x4* w4 = (const f32x4*)W1;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n4; e += NH_MAX_BLOCKS * 256) {
        const f32x4 v = w4[e];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    m = wave_max_dpp(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__global__ __launch_bounds__(256) void k_norm_pack_w1(int J, const float* __restrict__ W1, const float* __restrict__ part,
                                                      float* __restrict__ sc, uint8_t* __restrict__ w1img) {
    __shared__ float scl;
    if (threadIdx.x < 64) {
        const float m = wave_max_dpp(part[threadIdx.x]);                  // NH_MAX_BLOCKS == 64: one block's maximum per lane
        if (threadIdx.x == 0) scl = ldexpf(1.f, f16_scale_exp(m));
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) { sc[0] = scl; sc[1] = 1.0f / scl; }
    pack_w1_b_kstep(blockIdx.x, J, W1, scl, w1img);
}

// NP person tiles of 32 a wave (against the same weight fragments: with NP = 2 they cross the L2 once per 64 persons), 4 / NP
// waves a workgroup; the fragments come PF k-steps ahead of their use.  Measured at 1M x 500, 90 % missing (tools/nenc_bench.hip,
// one box): k_norm_enc_fwd_b 407 us; this kernel NP = 1: PF = 3 / 5 / 7: 294 / 306 / 295 us; NP = 2 (one wave a SIMD): PF = 3 / 7 /
// 11: 299 / 305 / 314 -- neither the depth of the prefetch nor half the L2 traffic moves it, and neither did four independent
// accumulator chains with the next response fragment made under this k-step's products (fc1 15 k instead of 20 k cycles of a
// wave's 34 k by s_memtime stamps, the launch unchanged): what bounds it is a wave's LIFE (stage 8 k | fc1 15 k | softplus and
// outputs 10 k cycles, one after the other) times sixteen rounds at eight waves a CU, which the 16.5 KB of LDS a wave hold
// there.  A form without LDS (the response bytes through registers: 16 bytes of its own row per lane and pair of k-steps, two
// v_permlane32_swap to serve both lane halves) is bit-identical and SLOWER at 500 items (436 us: 32 row pieces a load
// instruction) and the fastest at 72 items (14.5 against 15.5 us for 70 k persons); not kept.  The library ships NP = 1, PF = 3
// for J >= 256 and batches from 4 096 persons on.
template <int NP>
__host__ __device__ inline size_t nh_wave_floats(int J) {
    const size_t a = (size_t)(32 * NP) * ef_ys(J) / 4, b = (size_t)(((32 * NP * J + 1023) / 1024) * 256);
    return ((a > b ? a : b) + 3) & ~(size_t)3;
}
template <int NP>
__host__ __device__ inline size_t nh_lds_bytes(int J) { return (4 / NP) * nh_wave_floats<NP>(J) * sizeof(float); }

template <int NP, int PF>
__global__ __launch_bounds__(64 * (4 / NP)) void k_norm_enc_fwd_h(
    EncDims dm, const uint8_t* __restrict__ y, const int64_t* __restrict__ rows, const uint8_t* __restrict__ w1img,
    const float* __restrict__ sc, const float* __restrict__ b1, const float* __restrict__ W21, const float* __restrict__ b21,
    const float* __restrict__ W22, const float* __restrict__ b22, float* __restrict__ h_out,
    float* __restrict__ loc_out, float* __restrict__ raw_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int H = 64, WP = 32 * NP, NW = 4 / NP, RING = PF + 1;
    const int J = dm.J, YS = ef_ys(J);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    float* R1 = smem + wave * nh_wave_floats<NP>(J);
    const int8_t* Yi = (const int8_t*)R1;
    const int64_t i0 = ((int64_t)blockIdx.x * NW + wave) * WP;
    if (i0 >= dm.nb) return;                                              // (no workgroup barrier in this kernel)
#ifdef NH_STAMPS
    unsigned long long nst_[5]; int nsn_ = 0;
#define NSTAMP() do { __builtin_amdgcn_s_waitcnt(0); nst_[nsn_++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define NSTAMP() do {} while (0)
#endif
    NSTAMP();
    // ---- this wave's response rows: WP consecutive rows by DMA, or gathered word by word
    const int n_ydma = (WP * J + 1023) / 1024;
    const bool ydense = !rows && ((J >> 2) & 1) && i0 + WP <= dm.nb && (i0 * J + (int64_t)n_ydma * 1024 <= dm.nb * (int64_t)J);
    const int ysr = ydense ? J : YS;
    const int n_ks = (J + 15) / 16;
    auto loadA = [&](f16x8 (&Af)[4], int ks) __attribute__((always_inline)) {
        ks = ks < n_ks ? ks : n_ks - 1;                                   // past the end: reload the last k-step (never used)
        const uint8_t* src = w1img + (int64_t)ks * FB_W1_KS + lane * 16;
#pragma unroll
        for (int f = 0; f < 4; ++f) Af[f] = *(const f16x8*)(src + f * 1024);
    };
    f16x8 A[RING][4];
    if (ydense) {
        const uint8_t* src = y + i0 * J + 16 * lane;
        const uint32_t lb = lds_addr_uniform(R1);
        for (int d = 0; d < n_ydma; ++d) dma16(src + d * 1024, lb + (uint32_t)d * 1024u);
#pragma unroll
        for (int u = 0; u < PF; ++u) loadA(A[u], u);
        vx_wait_vmem();                                                   // the response rows (DMA) and the first fragments
    } else {
        const int YW = YS / 4, JW = J / 4;
        uint32_t* Yw = (uint32_t*)R1;
        for (int e = lane; e < WP * YW; e += 64) {
            const int pp = e / YW, wq = e - pp * YW;
            const int64_t ii = i0 + pp;
            uint32_t v = 0u;
            if (wq < JW && ii < dm.nb) {
                const int64_t row = rows ? rows[ii] : ii;
                v = *(const uint32_t*)(y + row * J + 4 * wq);              // bytes 0/1/255 == int8 0/1/-1 (vi.py:680-682)
            }
            Yw[e] = v;
        }
#pragma unroll
        for (int u = 0; u < PF; ++u) loadA(A[u], u);
    }
    __builtin_amdgcn_wave_barrier();
    NSTAMP();                                                             // 1: responses staged
    const float w1_inv = sc[1];
    // [person tile][hidden tile] for the leading terms, and as many for the remainders: 4 NP independent chains, so that no
    // MFMA waits for the one before it (with two chains of two a k-step took 600 cycles where its MFMAs take 128:
    // -DNH_STAMPS in tools/nenc_bench.hip); the response fragment of the NEXT k-step is made while this one's products run
    f32x16 acc[NP][2], acl[NP][2];
#pragma unroll
    for (int t = 0; t < NP; ++t) { acc[t][0] = zero16(); acc[t][1] = zero16(); acl[t][0] = zero16(); acl[t][1] = zero16(); }
    const int8_t* const yl = Yi + l31 * ysr + 8 * half;
    f16x8 yb[NP], ybn[NP];
#pragma unroll
    for (int t = 0; t < NP; ++t) yb[t] = fb_y_frag((const uint32_t*)(yl + 32 * t * ysr));             // rows are 4-byte aligned
    auto compute = [&](const f16x8 (&Af)[4], int ks) __attribute__((always_inline)) {
        const int kn = ks + 1 < n_ks ? ks + 1 : ks;
#pragma unroll
        for (int t = 0; t < NP; ++t) { acl[t][0] = mfma_f16(Af[1], yb[t], acl[t][0]); acl[t][1] = mfma_f16(Af[3], yb[t], acl[t][1]); }
#pragma unroll
        for (int t = 0; t < NP; ++t) ybn[t] = fb_y_frag((const uint32_t*)(yl + 32 * t * ysr + 16 * kn));
#pragma unroll
        for (int t = 0; t < NP; ++t) { acc[t][0] = mfma_f16(Af[0], yb[t], acc[t][0]); acc[t][1] = mfma_f16(Af[2], yb[t], acc[t][1]); }
#pragma unroll
        for (int t = 0; t < NP; ++t) yb[t] = ybn[t];
    };
    for (int c = 0; c < n_ks; c += RING) {
#pragma unroll
        for (int u = 0; u < RING; ++u) {
            loadA(A[(u + PF) % RING], c + u + PF);
            if (c + u < n_ks) compute(A[u], c + u);
        }
    }
#pragma unroll
    for (int t = 0; t < NP; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[t][0][r] += acl[t][0][r]; acc[t][1][r] += acl[t][1][r]; }
    NSTAMP();                                                             // 2: fc1
    // ---- softplus, the two 1-row heads as per-lane dot products over the 32 hidden units a lane holds
#pragma unroll
    for (int t = 0; t < NP; ++t) {
        const int64_t i = i0 + 32 * t + l31;
        float sl = 0.f, sr = 0.f;
#pragma unroll
        for (int ht = 0; ht < 2; ++ht)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int hh0 = 32 * ht + 8 * g + 4 * half;
                const float4 bb = *(const float4*)(b1 + hh0);
                const float4 w21 = make_float4(W21[hh0], W21[hh0 + 1], W21[hh0 + 2], W21[hh0 + 3]);
                const float4 w22 = make_float4(W22[hh0], W22[hh0 + 1], W22[hh0 + 2], W22[hh0 + 3]);
                float4 hv;
                hv.x = softplusf_(fmaf(acc[t][ht][4 * g + 0], w1_inv, bb.x));      // vi.py:432
                hv.y = softplusf_(fmaf(acc[t][ht][4 * g + 1], w1_inv, bb.y));
                hv.z = softplusf_(fmaf(acc[t][ht][4 * g + 2], w1_inv, bb.z));
                hv.w = softplusf_(fmaf(acc[t][ht][4 * g + 3], w1_inv, bb.w));
                sl += hv.x * w21.x + hv.y * w21.y + hv.z * w21.z + hv.w * w21.w;
                sr += hv.x * w22.x + hv.y * w22.y + hv.z * w22.z + hv.w * w22.w;
                if (i < dm.nb) *(float4*)(h_out + i * H + hh0) = hv;
            }
        sl = half_sum32(sl);
        sr = half_sum32(sr);
        if (half == 0 && i < dm.nb) {
            loc_out[i] = sl + b21[0];
            raw_out[i] = sr + b22[0];
        }
    }
#ifdef NH_STAMPS
    NSTAMP();                                                             // 3: softplus, heads, outputs
    if (blockIdx.x == 3000 && lane == 0)
        printf("NSTAMPS NP %d PF %d wave %d: stage %llu fc1 %llu out %llu total %llu\n", NP, PF, wave, nst_[1] - nst_[0], nst_[2] - nst_[1], nst_[3] - nst_[2], nst_[3] - nst_[0]);
#endif
#undef NSTAMP
}

