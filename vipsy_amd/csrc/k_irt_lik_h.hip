// Model likelihood + gradients for the 1PL / 2PL link, 97 <= D + 1 <= 112, on the fp16 MFMA with two-term operand splitting
// ("f16x2", vx_common.h): the tiling, pipeline and LDS images of k_irt_lik_b.hip (read that header first; vi.py:32-41
// response functions, vi.py:596-625 model + missing mask, Bernoulli log-lik) with every operand as TWO fp16 terms and THREE
// products per pair instead of three bf16 terms and six / five products: 138 MFMAs of 32 cycles per 64-person tile and wave
// instead of 244, a 28 KB x image instead of 42 KB, 120 operand registers instead of 180.
//
// Powers of two (all exact, taken off the fp32 accumulators):
//   x     2^LH_XEXP, fixed: the image is written by the guide-forward kernel before the largest |x| of the launch is known.
//         Small |x| keep an absolute error of 2^-32 (fp16 subnormals are honoured by the MFMA).  |x| 2^7 beyond the largest
//         fp16 (|x| >= 511.75) cannot be represented: the image writers saturate the value AND raise the overflow word behind
//         the image (the forward's image: byte offset tiles * LH_XT_BYTES, LH_FLAG_BYTES of its own); k_irt_lik_h returns at once when it is set, and the bf16x3
//         kernel (k_irt_lik_b, no range limit), launched behind it every time, runs instead of returning at once -- one
//         weakly discriminating item would otherwise see a latent of 2 000 as 511.75 and leave the link's clamp.
//   a, b  by the largest |a|, |b| of the workgroup's own 128-item chunk (its Z, gx partial and GA slab are its own results).
//   R     = scale Dc dlogp/dz with |dlogp/dz| <= 1 for this link: by |scale Dc|.  (The 3PL / 4PL links keep k_irt_lik_b:
//         with d < c their dlogp/dz is only bounded by 1 / eps.)
// R has two fp16 terms here (2^-22 relative) where k_irt_lik_b keeps two bf16 terms (2^-17).
#pragma once
#include "k_irt_lik_b.hip"

#define LH_XEXP 7
#define LH_XT_BYTES (2 * LB_PLANE)                 // 28672: two split planes of a 64-person tile, each laid out by lb_xoff

__device__ __forceinline__ f16x8 lh_frag(lb_u32x2 lo, lb_u32x2 hi) {
    const lb_u32x4 q = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(f16x8, q);
}
__device__ __forceinline__ f16x8 lh_read128(lb_lds* p) { return *(__attribute__((address_space(3))) const f16x8*)p; }

#define LH_FLAG_BYTES 64                           // the overflow word behind the tile images (its own 64 bytes)
// eight values of x_aug -> the two fp16 fragments of x 2^LH_XEXP, saturated; true when a value was out of range
__device__ __forceinline__ bool lh_split_x(const float (&v)[8], f16x8& fh, f16x8& fl) {
    float w[8];
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float t = v[j] * (float)(1 << LH_XEXP);
        bad |= !(fabsf(t) <= 65504.0f);                              // per element: a NaN is out of range too (fmaxf would drop it)
        w[j] = __builtin_amdgcn_fmed3f(t, -65504.0f, 65504.0f);
    }
    split2h_frag(w, 1.0f, fh, fl);
    return bad;
}

// x fp32 [nb][D] -> tile images (the two fp16 terms of x_aug 2^LH_XEXP); persons past nb: all-zero rows
__global__ __launch_bounds__(256) void k_lik_ximg_h(int D, int64_t nb, const float* __restrict__ x, uint8_t* __restrict__ img,
                                                    uint32_t* __restrict__ ovf /*the overflow word (cleared by the caller)*/) {
    const int64_t tile = blockIdx.x;
    uint8_t* out = img + tile * LH_XT_BYTES;
    for (int e = threadIdx.x; e < LB_P * 2 * LB_NKS; e += blockDim.x) {
        const int p = e / (2 * LB_NKS), ch = e - p * (2 * LB_NKS);
        const int64_t i = tile * LB_P + p;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * ch + j;
            v[j] = (i < nb) ? (k < D ? x[i * D + k] : (k == D ? 1.0f : 0.f)) : 0.f;
        }
        f16x8 fh, fl;
        if (lh_split_x(v, fh, fl)) atomicOr(ovf, 1u);
        const uint32_t o = lb_xoff(p, ch);
        *(f16x8*)(out + o) = fh;
        *(f16x8*)(out + LB_PLANE + o) = fl;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_irt_lik_h.  Arguments as k_irt_lik_b (no c / d leaves); a workgroup owns one 128-item chunk and walks 64-person tiles.
// The work of a tile is divided between TWO KINDS of waves, eight waves a workgroup (two per SIMD, each under 256
// registers):
//     cell waves 0..3      Z of the next person half (21 MFMAs)  beside  the cells of this half: link, log-lik, R and its
//                          split, R image and log-lik quads into LDS; and every global -> LDS transfer (x, y of the half
//                          three periods ahead: their only vector-memory operations, so vmcnt counts transfers exactly).
//     gradient waves 4..7  of the half the cell waves finished one period earlier: GA (24 MFMAs; A = x^T by transposed
//                          reads, B = R rows by 16-byte reads of the R image), gx (24 MFMAs; A = a registers, B = R^T by
//                          transposed reads), gx and the persons' log-lik to global memory straight from the accumulator
//                          layout (lanes = 32 consecutive persons: 128-byte rows) -- stores nobody ever waits for.
// The vector work of the cells (the longest chain of the kernel) and the 48 gradient MFMAs of a half run on the same SIMD
// from different waves: the hardware interleaves what a one-wave form (k_irt_lik_b's four phases) leaves to instruction
// scheduling, and a wave that waits (LDS, barrier) leaves the SIMD to the other.  Same harness, same box: one-wave f16x2
// form 1.70 ms, this form 1.53 ms for 1M x 500 x 100 (tools/likh_test.hip).
//
// One barrier per PERIOD = person half n (32 persons; halves of the workgroup's tiles in order):
//     period n    cell waves      transfers of half n + 3;  cells(n) [z(n), y(n) -> R(n), LP(n)]  beside  Z(n + 1) [x(n + 1)];
//                                 wait for the transfers of half n + 2
//                 gradient waves  GA(n - 1), gx(n - 1) [x(n - 1), R(n - 1)];  stores, ll(n - 1)
// LDS rings: x six halves (= three tile images in place: half n + 3 replaces half n - 3), y four halves, R and LP two.
// ABL: ablation bits for tools/likh_test.hip only (0 in the library): 1 no cell math, 4 no transfers, 8 no scheduling
// hints, 16 no output stores.
#define LH_THREADS 512
#define LH_XHALF (LB_PLANE / 2)                                     // 7168: one split plane of one person half
#define LH_R_BYTES (2 * LB_RPLANE)                                  // R image of a half: [2 terms][128 items][32 persons] fp16
#define LH_Y_BYTES (LB_JC * 32)                                     // responses of a half: [128 items][32 persons]
#define LH_LP_BYTES (32 * 32 * 4)                                   // [32 persons][32 item quads] fp32
#define LH_LDS_BYTES (3 * LH_XT_BYTES + 2 * LH_R_BYTES + 4 * LH_Y_BYTES + 2 * LH_LP_BYTES)   // 143360
#define LH_VM_HALF 5                                                // transfers of one half per cell wave (4 x + 1 y)

// LDS accesses of this wave done, then the workgroup barrier -- without the release fence of __syncthreads(), which would
// also wait for the gradient waves' global stores (the intrinsic alone is "no memory" to the compiler: the empty asm
// statements keep its own scheduling from carrying an LDS access across)
__device__ __forceinline__ void lh_barrier() {
    asm volatile("" ::: "memory");                                   // the compiler moves no LDS access across the barrier either
    __builtin_amdgcn_s_waitcnt(0xC07F);                              // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int ABL = 0>
__global__ __launch_bounds__(LH_THREADS, 1) void k_irt_lik_h(
    LikBDims dm, const uint8_t* __restrict__ yT, int64_t yT_stride, const uint8_t* __restrict__ ximg,
    const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ gx_part, float* __restrict__ ll_part,
    float* __restrict__ slabs, const uint32_t* __restrict__ ovf /*the image's overflow word*/) {
    extern __shared__ __attribute__((aligned(16))) char smem_lh[];
    const int D = dm.D, J = dm.J;
    // a latent out of the image's range: k_irt_lik_b, launched behind this kernel, does the work (uniform: no barrier is passed)
    if (*ovf != 0u) return;
    lb_lds* const Xb = (lb_lds*)smem_lh;                            // [3 tiles][2 planes][64 persons] (lb_xoff)
    lb_lds* const Rb = Xb + 3 * LH_XT_BYTES;                         // [2 slots][2 terms][128 items][64 B]
    lb_lds* const Yb = Rb + 2 * LH_R_BYTES;                         // [4 slots][128 items][32 B]
    lb_lds* const LPb = Yb + 4 * LH_Y_BYTES;                        // [2 slots][32 persons][32 quads]
    const uint32_t Xb_l = (uint32_t)(size_t)Xb, Yb_l = Xb_l + 3 * LH_XT_BYTES + 2 * LH_R_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    int g, pr;                                                       // XCD-aware decode, as k_irt_lik_b
    if ((dm.n_pr & 7) == 0) {
        const int L = blockIdx.x;
        g = (L >> 3) % dm.groups;
        pr = (L & 7) + 8 * (L / (8 * dm.groups));
    } else {
        g = blockIdx.x % dm.groups;
        pr = blockIdx.x / dm.groups;
    }
    const int j0 = g * LB_JC;
    const int64_t n_ptiles = (dm.nb + LB_P - 1) / LB_P;
    const int64_t nbp = n_ptiles * LB_P;
    const int nt = pr < n_ptiles ? (int)((n_ptiles - 1 - pr) / dm.n_pr) + 1 : 0;     // tiles pr, pr + n_pr, ... of this workgroup

    // ---- powers of two of the operands (see the head of this file)
    float amax = 0.f;
    for (int e = tid; e < (D + 1) * LB_JC; e += LH_THREADS) {
        const int k = e >> 7, jj = j0 + (e & (LB_JC - 1));
        if (jj < J) amax = fmaxf(amax, fabsf(k < D ? a[(int64_t)k * J + jj] : b[jj]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    if (lane == 0) *(__attribute__((address_space(3))) float*)(LPb + 4 * wave) = amax;
    __syncthreads();
    {
        const f32x4 m0 = *(__attribute__((address_space(3))) const f32x4*)LPb, m1 = *(__attribute__((address_space(3))) const f32x4*)(LPb + 16);
        amax = fmaxf(fmaxf(fmaxf(m0[0], m0[1]), fmaxf(m0[2], m0[3])), fmaxf(fmaxf(m1[0], m1[1]), fmaxf(m1[2], m1[3])));
    }
    const float sdc = dm.scale * dm.Dc;
    const int e_az = f16_scale_exp(fabsf(dm.Dc) * amax), e_ag = f16_scale_exp(amax), e_r = f16_scale_exp(fabsf(sdc));

    if (wave < 4) {
        // =============================================== cell waves ===============================================
        const int jl = 32 * wave + l31;                              // this lane's item within the chunk
        const int jw = j0 + jl;
        const bool jv = jw < J;
        const float s_az = ldexpf(1.0f, e_az);
        const float sdcs = ldexpf(sdc, e_r);                         // R 2^e_r = sdcs dlogp/dz
        const float z_inv = ldexpf(1.0f, -(LH_XEXP + e_az));         // Z accumulator -> z
        f16x8 aZ[2][LB_NKS];
#pragma unroll
        for (int s = 0; s < LB_NKS; ++s) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 16 * s + 8 * half + j;
                float t = 0.f;
                if (jv) {
                    if (k < D) t = a[(int64_t)k * J + jw];
                    else if (k == D) t = b[jw];
                }
                v[j] = dm.Dc * t;                                    // z = Dc * (x.a + b)
            }
            split2h_frag(v, s_az, aZ[0][s], aZ[1][s]);
        }
        // Z row reads: person row p = 32 ph + l31, chunk 2 s + half (byte offsets inside one split plane)
        uint32_t zE[2], zO[2], z6[2];
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            const int p = 32 * ph + l31, xr = (p >> 2) & 3;
            const uint32_t base = (uint32_t)(p >> 3) * (256u * LB_NKS) + 64u * (p & 7);
            zE[ph] = base + 16u * ((0 + half) ^ xr);
            zO[ph] = base + 16u * ((2 + half) ^ xr);
            z6[ph] = (uint32_t)(p >> 3) * (256u * LB_NKS) + 1536u + 32u * (p & 7) + 16u * (half ^ ((p >> 4) & 1));
        }
        const uint32_t rSw = (uint32_t)((jl & 7) ^ ((jl >> 3) & 1));     // R image write: item row jl, 8-byte piece swizzle
        lb_lds* const rWp = Rb + jl * 64;
        lb_lds* const yP0 = Yb + jl * 32;                            // this lane's item row of a y slot
        const int qc = l31 & 3;                                      // lane within its item quad: stores register 4 g4 + qc
        lb_lds* const lpW = LPb + 4 * ((4 * half + qc) * 32 + 8 * wave + (l31 >> 2));     // + 128 * (8 g4) + slot

        // transfers of one person half: x 14 pieces of 1 KB (7 per split plane), y 4 pieces (32 item rows x 32 persons each);
        // LH_VM_HALF per wave
        auto stage_half = [&](int64_t tile, int h, int xbuf, int yslot) {
            if constexpr (ABL & 4) return;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int q = wave + 4 * u;                                // wave-uniform
                q = q < 14 ? q : 13;                                 // past the end: a repeat of the last
                const int plane = q >= 7, pc = q - 7 * plane;
                const uint32_t off = (uint32_t)plane * LB_PLANE + (uint32_t)h * LH_XHALF + (uint32_t)pc * 1024u;
                dma16(ximg + tile * LH_XT_BYTES + off + lane * 16,
                      __builtin_amdgcn_readfirstlane(Xb_l + (uint32_t)xbuf * LH_XT_BYTES + off));
            }
            int jr = j0 + 32 * wave + (lane >> 1);
            jr = jr < J ? jr : J;                                    // items past J: the all-254 row
            dma16(yT + (int64_t)jr * yT_stride + tile * LB_P + 32 * h + 16 * (lane & 1),
                  __builtin_amdgcn_readfirstlane(Yb_l + (uint32_t)yslot * LH_Y_BYTES + (uint32_t)wave * 1024u));
        };
        auto z_frags = [&](lb_lds* xb, int ph, int s, f16x8& fh, f16x8& fl) {
            lb_lds* o = xb + ((s == 6) ? z6[ph] : (((s & 1) ? zO[ph] : zE[ph]) + 512 * (s >> 1)));
            fh = lh_read128(o);
            fl = lh_read128(o + LB_PLANE);
        };
        auto z_mma = [&](int s, f32x16& z, const f16x8& xh, const f16x8& xl) {
            z = mfma_f16(xl, aZ[0][s], z);
            z = mfma_f16(xh, aZ[1][s], z);
            z = mfma_f16(xh, aZ[0][s], z);
        };
        float lpq[4];
        uint32_t Rp[2][2];                                           // [term][pair of the run of four]
        // the cells of one register PAIR (2 i, 2 i + 1) of the half: link, R and its split, R image, log-lik quad sums
        auto cell_pair = [&](auto ic, const f32x16& z, const lb_u32x4& yq0, const lb_u32x4& yq1, lb_lds* rW, lb_lds* lpWs) {
            constexpr int i = decltype(ic)::value;
            constexpr int g4 = i >> 1;                               // run of four registers = persons 8 g4 + 4 half + 0..3
            const uint32_t wlo = (g4 < 2 ? yq0 : yq1)[2 * (g4 & 1)], whi = (g4 < 2 ? yq0 : yq1)[2 * (g4 & 1) + 1];
            const uint32_t yw = half ? whi : wlo;
            float rv[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int r = 2 * i + e;
                const float yf = (float)((yw >> (8 * (r & 3))) & 0xffu);
                const float zz = z[r] * z_inv;
                float lp, dz, dc, dd;
                if constexpr (ABL & 1) {
                    lp = zz; dz = zz + yf; dc = 0.f; dd = 0.f;
                } else {
                    irt_cell_f<2>(zz, yf, 0.f, 1.f, 0.f, lp, dz, dc, dd);
                }
                rv[e] = sdcs * dz;
                const float l2 = lp + dpp_mov0<0xB1, 0xF>(lp);       // + the terms of the three other items of the quad
                lpq[2 * (i & 1) + e] = l2 + dpp_mov0<0x4E, 0xF>(l2);
            }
            split2h_pair(rv[0], rv[1], Rp[0][i & 1], Rp[1][i & 1]);
            if constexpr (i & 1) {                                   // registers 4 g4 .. 4 g4 + 3 are complete
                const float t01 = (qc & 1) ? lpq[1] : lpq[0], t23 = (qc & 1) ? lpq[3] : lpq[2];
                *(__attribute__((address_space(3))) float*)(lpWs + 128 * (8 * g4)) = (qc & 2) ? t23 : t01;
                lb_lds* wb = rW + 8u * (((2 * g4 + half) ^ rSw) & 7);
#pragma unroll
                for (int sp = 0; sp < 2; ++sp) {
                    const lb_u32x2 w = {Rp[sp][0], Rp[sp][1]};
                    *(__attribute__((address_space(3))) lb_u32x2*)(wb + sp * LB_RPLANE) = w;
                }
            }
        };

        f32x16 z[2] = {zero16(), zero16()};
        // halves 0, 1, 2 before anything else; 0 and 1 must have landed for period 0
        if (nt > 0) { stage_half(pr, 0, 0, 0); stage_half(pr, 1, 0, 1); }
        if (nt > 1) {
            stage_half(pr + dm.n_pr, 0, 1, 2);
            __builtin_amdgcn_s_waitcnt(0x0F70 | LH_VM_HALF);
        } else {
            vx_wait_vmem();
        }
        lh_barrier();                                               // P
        if (nt > 0) {                                                // Z(0): once, nothing beside it
            f16x8 fh, fl;
            static_for<LB_NKS>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                z_frags(Xb, 0, s, fh, fl);
                z_mma(s, z[0], fh, fl);
            });
        }
        int b3 = 0;                                                  // it % 3: tile image of tile it
        for (int it = 0; it < nt; ++it) {
            const int b3n = (b3 == 2) ? 0 : b3 + 1, b3nn = (b3n == 2) ? 0 : b3n + 1;
            const int64_t tile = pr + (int64_t)it * dm.n_pr;
            static_for<2>([&](auto hc) {
                constexpr int h = decltype(hc)::value;
                // ---- period n = 2 it + h: transfers of half n + 3, cells(n) beside Z(n + 1)
                const bool stage = (h == 0) ? (it + 1 < nt) : (it + 2 < nt);         // uniform: half n + 3 exists
                if (stage) {
                    if constexpr (h == 0) stage_half(tile + dm.n_pr, 1, b3n, 2 * ((it + 1) & 1) + 1);
                    else stage_half(tile + 2 * (int64_t)dm.n_pr, 0, b3nn, 2 * (it & 1));
                }
                const bool has_next = (h == 0) || (it + 1 < nt);     // uniform
                lb_lds* const xn = Xb + (h == 0 ? b3 : b3n) * LH_XT_BYTES;           // tile image holding half n + 1
                lb_lds* const yP = yP0 + (2 * (it & 1) + h) * LH_Y_BYTES;
                lb_lds* const rW = rWp + h * LH_R_BYTES;
                lb_lds* const lpWs = lpW + h * LH_LP_BYTES;
                const lb_u32x4 yq0 = *(__attribute__((address_space(3))) const lb_u32x4*)(yP), yq1 = *(__attribute__((address_space(3))) const lb_u32x4*)(yP + 16);
                z[h ^ 1] = zero16();
                f16x8 fh[3], fl[3];                                  // fragments two steps ahead
                if (has_next) { z_frags(xn, h ^ 1, 0, fh[0], fl[0]); z_frags(xn, h ^ 1, 1, fh[1], fl[1]); }
                static_for<8>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    if (has_next) {
                        if constexpr (i + 2 < LB_NKS) z_frags(xn, h ^ 1, i + 2, fh[(i + 2) % 3], fl[(i + 2) % 3]);
                        if constexpr (i < LB_NKS) z_mma(i, z[h ^ 1], fh[i % 3], fl[i % 3]);
                    }
                    cell_pair(ic, z[h], yq0, yq1, rW, lpWs);
                    if constexpr (!(ABL & 8)) {
                        __builtin_amdgcn_sched_group_barrier(LB_MASK_DSR, 2, 0);
#pragma unroll
                        for (int q = 0; q < 3; ++q) {
                            __builtin_amdgcn_sched_group_barrier(LB_MASK_MFMA, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(LB_MASK_VALU, 24, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
                // the transfers of half n + 2 (issued one period ago) have landed; those of this period may be in flight
                if (stage) __builtin_amdgcn_s_waitcnt(0x0F70 | LH_VM_HALF); else vx_wait_vmem();
                lh_barrier();
            });
            b3 = b3n;
        }
        lh_barrier();                                               // period 2 nt: the gradient waves finish the last half
    } else {
        // ============================================= gradient waves =============================================
        const int gw = wave - 4;
        const int t = tid - 256;
        const int jl = 32 * gw + l31, jw = j0 + jl;                  // GA column of this lane
        const bool jv = jw < J;
        const float s_ag = ldexpf(1.0f, e_ag);
        const float gx_inv = ldexpf(1.0f, -(e_ag + e_r));
        const float ga_inv = ldexpf(1.0f, -(LH_XEXP + e_r));
        f16x8 aG[2][8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            float v[8];
            const int kg = 32 * gw + l31;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int jj = j0 + 16 * s + 8 * half + j;
                v[j] = (kg < D && jj < J) ? a[(int64_t)kg * J + jj] : 0.f;
            }
            split2h_frag(v, s_ag, aG[0][s], aG[1][s]);
        }
        f32x16 ga[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) ga[kt] = zero16();
        // transposed reads: 16-lane group (half, gl), lane 4 q4 + pp of the group
        const int gl = (lane >> 4) & 1, q4 = (lane & 15) >> 2, pp = lane & 3;
        // GA, A = x^T: person rows 16 s2 + 8 half + 4 e + q4 of the half (k index 8 half + 4 e + q = the persons of the 16-byte
        // R row read below, in order), latent columns 32 kt + 16 gl + 4 pp .. + 3
        uint32_t gaN[2], ga3N[2][2], rB[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            gaN[e] = 1792u * half + 64u * (4 * e + q4) + 16u * ((2 * gl + (pp >> 1)) ^ (2 * half + e)) + 8u * (pp & 1);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)                           // k-tile 3: columns 96..111, read by both lane groups
                ga3N[e][s2] = 1792u * half + 1536u + 32u * (4 * e + q4) + 16u * ((pp >> 1) ^ s2) + 8u * (pp & 1);
            // gx, B = R^T: item row 16 s + 8 half + 4 e + q4, persons 16 gl + 4 pp .. + 3 (8-byte piece 4 gl + pp, swizzled)
            rB[e] = 512u * half + 256u * e + 64u * q4 + 8u * (((4 * gl + pp) ^ (4 * e + q4) ^ half) & 7);
        }
        // GA, B = R rows: item row jl, persons 16 s2 + 8 half .. + 7 = 16-byte chunk 2 s2 + half; the 8-byte pieces of the
        // image are swizzled by rSw: the chunk moves by rSw >> 1 and its two pieces swap when rSw is odd
        const uint32_t rSw = (uint32_t)((jl & 7) ^ ((jl >> 3) & 1));
        const bool rswap = rSw & 1;
        float* const gxw = gx_part + ((int64_t)g * LB_DP + 32 * gw) * nbp;      // uniform: + (8 (r >> 2) + (r & 3)) nbp + i0 + 32 ph
        const uint32_t gxl = (uint32_t)(4 * half) * (uint32_t)nbp + l31;              // this lane (host: 16 nbp < 2^31)
        lb_lds* const lpR = LPb + 4 * ((t >> 3) * 32 + 4 * (t & 7));

        auto ga_frags = [&](lb_lds* xb, int ph, int u, f16x8& fh, f16x8& fl) {      // u = 4 s2 + kt
            const int s2 = u >> 2, kt = u & 3;
            const uint32_t gb = 1792u * (4 * ph + 2 * s2);
            lb_lds* o0 = xb + gb + ((kt == 3) ? ga3N[0][s2] : gaN[0] + 512 * kt);
            lb_lds* o1 = xb + gb + ((kt == 3) ? ga3N[1][s2] : gaN[1] + 512 * kt);
            fh = lh_frag(lb_tr_read(o0), lb_tr_read(o1));
            fl = lh_frag(lb_tr_read(o0 + LB_PLANE), lb_tr_read(o1 + LB_PLANE));
        };
        auto r_rows = [&](lb_lds* rs, int s2, f16x8& rh, f16x8& rl) {
            const uint32_t o = 16u * ((uint32_t)(2 * s2 + half) ^ (rSw >> 1));
            const lb_u32x4 vh = *(__attribute__((address_space(3))) const lb_u32x4*)(rs + o);
            const lb_u32x4 vl = *(__attribute__((address_space(3))) const lb_u32x4*)(rs + LB_RPLANE + o);
            const lb_u32x4 wh = {rswap ? vh[2] : vh[0], rswap ? vh[3] : vh[1], rswap ? vh[0] : vh[2], rswap ? vh[1] : vh[3]};
            const lb_u32x4 wl = {rswap ? vl[2] : vl[0], rswap ? vl[3] : vl[1], rswap ? vl[0] : vl[2], rswap ? vl[1] : vl[3]};
            rh = __builtin_bit_cast(f16x8, wh);
            rl = __builtin_bit_cast(f16x8, wl);
        };
        auto gx_frags = [&](lb_lds* rs, int s, f16x8& fh, f16x8& fl) {
            lb_lds* p0 = rs + rB[0] + 1024 * s;
            lb_lds* p1 = rs + rB[1] + 1024 * s;
            fh = lh_frag(lb_tr_read(p0), lb_tr_read(p1));
            fl = lh_frag(lb_tr_read(p0 + LB_RPLANE), lb_tr_read(p1 + LB_RPLANE));
        };
        // GA, gx, the stores and the log-lik of one finished half (ph: its half in tile image xb; R / LP slot = ph)
        auto grad_half = [&](auto phc, lb_lds* xb, int64_t i0) {
            constexpr int ph = decltype(phc)::value;
            lb_lds* const rs = Rb + ph * LH_R_BYTES;
            f16x8 rh[2], rl[2], fh[3], fl[3];                        // fragments two steps ahead
            r_rows(rs + jl * 64, 0, rh[0], rl[0]);
            r_rows(rs + jl * 64, 1, rh[1], rl[1]);
            ga_frags(xb, ph, 0, fh[0], fl[0]);
            ga_frags(xb, ph, 1, fh[1], fl[1]);
            f32x16 gx = zero16();
            static_for<16>([&](auto uc) {                            // steps 0..7: GA (s2, kt), 8..15: gx (16 items each)
                constexpr int u = decltype(uc)::value, c = u % 3, n2 = (u + 2) % 3;
                if constexpr (u + 2 < 8) ga_frags(xb, ph, u + 2, fh[n2], fl[n2]);
                else if constexpr (u + 2 < 16) gx_frags(rs, u + 2 - 8, fh[n2], fl[n2]);
                if constexpr (u < 8) {
                    constexpr int s2 = u >> 2, kt = u & 3;
                    ga[kt] = mfma_f16(fl[c], rh[s2], ga[kt]);
                    ga[kt] = mfma_f16(fh[c], rl[s2], ga[kt]);
                    ga[kt] = mfma_f16(fh[c], rh[s2], ga[kt]);
                } else {
                    constexpr int s = u - 8;
                    gx = mfma_f16(aG[1][s], fh[c], gx);
                    gx = mfma_f16(aG[0][s], fl[c], gx);
                    gx = mfma_f16(aG[0][s], fh[c], gx);
                }
            });
            // log-lik of the persons of the half: thread = (person, eighth of the 32 item quads)
            {
                const f32x4 v = *(__attribute__((address_space(3))) const f32x4*)(lpR + ph * LH_LP_BYTES);
                float sll = (v[0] + v[1]) + (v[2] + v[3]);
                sll += dpp_mov0<0xB1, 0xF>(sll);
                sll += dpp_mov0<0x4E, 0xF>(sll);
                sll += dpp_mov0<0x141, 0xF>(sll);                    // row_half_mirror: the other quad of the eight lanes
                if constexpr (!(ABL & 16)) ll_part[(int64_t)g * nbp + i0 + 32 * ph + (t >> 3)] = sll;
            }
            // gx: C layout of gx^T: lane = person, register r = latent row 32 gw + 8 (r >> 2) + 4 half + (r & 3): one 128-byte
            // row per half-wave and store (rows D .. of the last run of eight land in the row padding: no guards).  (Four
            // 16-byte stores per lane after a transposition through LDS, as k_irt_lik_b does, measured the same.)
            if constexpr (!(ABL & 16)) {
                float* dst = gxw + i0 + 32 * ph;                     // uniform base, 32-bit lane offset: no vector address arithmetic
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (32 * gw + 8 * (r >> 2) < D)                  // uniform
                        (dst + (int64_t)(8 * (r >> 2) + (r & 3)) * nbp)[gxl] = gx[r] * gx_inv;
            }
        };

        lh_barrier();                                               // P
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        int b3 = 0, b3p = 2;                                         // it % 3, (it - 1) % 3
        for (int it = 0; it < nt; ++it) {
            const int64_t tile = pr + (int64_t)it * dm.n_pr;
            // ---- period n = 2 it: gradients of half 1 of tile it - 1
            if (it > 0) grad_half(I1{}, Xb + b3p * LH_XT_BYTES, (tile - dm.n_pr) * LB_P);
            lh_barrier();
            // ---- period n = 2 it + 1: gradients of half 0 of tile it
            grad_half(I0{}, Xb + b3 * LH_XT_BYTES, tile * LB_P);
            lh_barrier();
            b3p = b3;
            b3 = (b3 == 2) ? 0 : b3 + 1;
        }
        if (nt > 0) grad_half(I1{}, Xb + b3p * LH_XT_BYTES, (pr + (int64_t)(nt - 1) * dm.n_pr) * LB_P);
        lh_barrier();                                               // period 2 nt
        // ---- item-gradient slab of this person range: GA C layout = rows k = 32 kt + crow32(r, half), column = this lane's item
        float* slab = slabs + (int64_t)pr * dm.slab_len;
        if (jv) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int k = 32 * kt + crow32(r, half);
                    if (k <= D) slab[(int64_t)k * J + jw] = ga[kt][r] * ga_inv;   // k == D lands in the b segment
                }
        }
    }
}
