// Model likelihood + gradients for 97 <= D + 1 <= 112 on the bf16 MFMA with three-term operand splitting ("bf16x3",
// vx_common.h): the same three contractions as k_irt_lik_r.hip (vi.py:32-66 response functions, vi.py:596-625 model +
// missing mask, Bernoulli log-lik), 276 MFMAs of 32 cycles per 64-person tile and wave instead of 360 of 64.
//
//   a workgroup owns ONE 128-item chunk for its whole life and walks 64-person tiles (item-stationary);
//   wave w owns items 32w..32w+31 of the chunk for Z / the cell epilogue / GA, and latent rows 32w..32w+31 for gx.
//
//     Z[p,j]    = sum_k x_aug[p,k] a_aug[k,j]     A <- x image rows (ds_read_b128),          B <- aZ registers (split once)
//     R[p,j]    = scale * Dc * dlogp/dz           epilogue, lane = item; R is split into three bf16 terms ONCE per cell
//     GA[k,j]  += sum_p x_aug[p,k] R[p,j]         A <- x image COLUMNS (ds_read_b64_tr_b16), B <- R registers: the C
//                                                 layout of Z, packed pairwise, already is a B fragment (persons permuted;
//                                                 the transposed read delivers x in the same order)
//     gx^T[k,p] = sum_j a[k,j] R[p,j]             A <- aG registers (split once),            B <- R image [item][person]
//                                                 in LDS (8-byte packed writes), read back transposed (tr_b16)
//
// Every operand that does not change from one person tile to the next lives in registers as bf16x3 fragments (a for Z:
// 84, a for gx: 96 VGPRs); x arrives ALREADY split (k_lik_ximg, or the guide-forward kernel writes the image): one image
// of [person][k] rows serves the row reads of Z and the column reads of GA (cdna_hip_programming.md T10).
//
// x image of one 64-person tile: 3 planes (h, m, l) of 64 rows x 112 bf16 (k = 0..D-1: x, k = D: 1, then zeros).  Rows are
// cut into 8-row x 32-column subtiles of 512 bytes (+ one 8 x 16 half subtile for k = 96..111) with the 16-byte chunk
// index XORed by bits of the row, so that the row reads (ds_read_b128, lanes = persons) and the transposed reads (lanes =
// 4 persons x 16 columns) are both bank-conflict free: lb_xoff.  The image in global memory IS the LDS image: linear DMA.
#pragma once
#include "vx_common.h"
#include <type_traits>

#define LB_P 64
#define LB_JC 128
#define LB_THREADS 256
#define LB_NKS 7                                   // k-steps of 16 latent columns (x_aug padded to 112)
#define LB_PLANE (8 * 256 * LB_NKS)                // bytes of one split plane of a 64-person tile (14336)
#define LB_XT_BYTES (3 * LB_PLANE)                 // 43008
#define LB_RPLANE 8192                             // R image plane: [128 items][32 persons] bf16
#define LB_YS 128
#define LB_LDS_BYTES (2 * LB_XT_BYTES + 6 * LB_RPLANE + LB_P * LB_YS + LB_P * 64 * 4)     // 159744

struct LikBDims {
    int D, J, model, groups, n_pr, gxt;
    float Dc, scale;
    int64_t nb, slab_len;
};

// byte offset inside one split plane of 16-byte chunk ch (latent columns 8 ch .. 8 ch + 7) of person row p (0..63)
__host__ __device__ inline uint32_t lb_xoff(int p, int ch) {
    const uint32_t grp = (uint32_t)(p >> 3) * (256u * LB_NKS);
    if (ch < 12) return grp + 512u * (ch >> 2) + 64u * (p & 7) + 16u * ((ch & 3) ^ ((p >> 2) & 3));
    return grp + 1536u + 32u * (p & 7) + 16u * ((ch & 1) ^ ((p >> 4) & 1));
}

#ifndef VX_STATIC_FOR
#define VX_STATIC_FOR
// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>)
template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}
#endif

// x fp32 [nb][D] -> tile images (the three bf16 terms of x_aug) + xsq[i] = |x_i|^2; persons past nb: all-zero rows
__global__ __launch_bounds__(256) void k_lik_ximg(int D, int64_t nb, const float* __restrict__ x, uint8_t* __restrict__ img,
                                                  float* __restrict__ xsq) {
    const int64_t tile = blockIdx.x;
    uint8_t* out = img + tile * LB_XT_BYTES;
    for (int e = threadIdx.x; e < LB_P * 2 * LB_NKS; e += blockDim.x) {
        const int p = e / (2 * LB_NKS), ch = e - p * (2 * LB_NKS);
        const int64_t i = tile * LB_P + p;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * ch + j;
            v[j] = (i < nb) ? (k < D ? x[i * D + k] : (k == D ? 1.0f : 0.f)) : 0.f;
        }
        bf16x8 fh, fm, fl;
        split3_frag(v, fh, fm, fl);
        const uint32_t o = lb_xoff(p, ch);
        *(bf16x8*)(out + o) = fh;
        *(bf16x8*)(out + LB_PLANE + o) = fm;
        *(bf16x8*)(out + 2 * LB_PLANE + o) = fl;
    }
    if (threadIdx.x < LB_P) {
        const int64_t i = tile * LB_P + threadIdx.x;
        if (i < nb) {
            float s = 0.f;
            for (int k = 0; k < D; ++k) { const float t = x[i * D + k]; s = fmaf(t, t, s); }
            xsq[i] = s;
        }
    }
}

typedef short lb_s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t lb_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t lb_u32x4 __attribute__((ext_vector_type(4)));

// transposed LDS read (ds_read_b64_tr_b16): per 16-lane group a block of 4 rows x 16 columns of 16-bit elements; lane
// 4q + p of the group supplies the address of row q, columns 4p..4p+3; lane i receives column i, row q in element q
__device__ __forceinline__ lb_u32x2 lb_tr_read(uint32_t lds_byte_addr) {
    const lb_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) lb_s16x4*)lds_byte_addr);
    return __builtin_bit_cast(lb_u32x2, v);
}
__device__ __forceinline__ bf16x8 lb_frag(lb_u32x2 lo, lb_u32x2 hi) {
    const lb_u32x4 q = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8, q);
}

// two fp32 values -> their three bf16 terms, packed pairwise (element 0 in the low half)
__device__ __forceinline__ void lb_split_pair(float a, float b, uint32_t& ph, uint32_t& pm, uint32_t& pl) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const __bf16 ah = (__bf16)a, bh = (__bf16)b;
    const float ar = a - (float)ah, br = b - (float)bh;
    const __bf16 am = (__bf16)ar, bm = (__bf16)br;
    const __bf16 al = (__bf16)(ar - (float)am), bl = (__bf16)(br - (float)bm);
    const bf16x2 vh = {ah, bh}, vm = {am, bm}, vl = {al, bl};
    ph = __builtin_bit_cast(uint32_t, vh);
    pm = __builtin_bit_cast(uint32_t, vm);
    pl = __builtin_bit_cast(uint32_t, vl);
}

// GEN: 3PL / 4PL cell; ROWS: the batch is a row gather (rows != null)
template <int GEN, int ROWS>
__global__ __launch_bounds__(LB_THREADS, 1) void k_irt_lik_b(
    LikBDims dm, const uint8_t* __restrict__ y, const int64_t* __restrict__ rows, const uint8_t* __restrict__ ximg,
    const float* __restrict__ xsq, const float* __restrict__ a, const float* __restrict__ b,
    const float* __restrict__ c_un, const float* __restrict__ d_un, float* __restrict__ gx_part /*[groups][nb][D] or [groups][D][nb]*/,
    float* __restrict__ ll_part /*[groups][nb]*/, float* __restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) char smem_lb[];
    const int D = dm.D, J = dm.J;
    char* xbuf = smem_lb;                                           // [2][3 planes][14336]
    char* Rimg = xbuf + 2 * LB_XT_BYTES;                            // [2 person halves][3 planes][128 items][64 B]
    uint8_t* Yb = (uint8_t*)(Rimg + 6 * LB_RPLANE);                 // [64][128]
    float* LPp = (float*)(Yb + LB_P * LB_YS);                       // [64 persons][64 item pairs]
    const uint32_t xbuf_l = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem_lb;   // LDS byte addresses
    const uint32_t Rimg_l = xbuf_l + 2 * LB_XT_BYTES, Yb_l = Rimg_l + 6 * LB_RPLANE;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    // XCD-aware decode: the `groups` workgroups that share a person tile sit on one XCD (same L2)
    int g, pr;
    if ((dm.n_pr & 7) == 0) {
        const int L = blockIdx.x;
        g = (L >> 3) % dm.groups;
        pr = (L & 7) + 8 * (L / (8 * dm.groups));
    } else {
        g = blockIdx.x % dm.groups;
        pr = blockIdx.x / dm.groups;
    }
    const int j0 = g * LB_JC;
    const int jw = j0 + 32 * wave + l31;                            // this lane's item (Z / epilogue / GA column)
    const bool jv = jw < J;
    const int64_t n_ptiles = (dm.nb + LB_P - 1) / LB_P;

    // ---- register-resident item operands, split once
    bf16x8 aZ[3][LB_NKS], aG[3][8];
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
            v[j] = dm.Dc * t;                                       // z = Dc * (x.a + b)
        }
        split3_frag(v, aZ[0][s], aZ[1][s], aZ[2][s]);
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        float v[8];
        const int kg = 32 * wave + l31;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int jj = j0 + 16 * s + 8 * half + j;
            v[j] = (kg < D && jj < J) ? a[(int64_t)kg * J + jj] : 0.f;
        }
        split3_frag(v, aG[0][s], aG[1][s], aG[2][s]);
    }
    float cj = 0.f, dj = 1.0f, omdj = 0.f, gc = 0.f, gd = 0.f;
    if (GEN) {
        cj = jv ? fminf(sigmoidf_(c_un[jw]), 1.0f - VX_EPS32) : 0.f;
        const bool has_d = (dm.model == 4 && jv);
        dj = has_d ? fminf(sigmoidf_(d_un[jw]), 1.0f - VX_EPS32) : 1.0f;
        omdj = has_d ? fmaxf(sigmoidf_(-d_un[jw]), VX_EPS32) : 0.f;
    }
    f32x16 ga[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) ga[kt] = zero16();

    // ---- per-lane LDS byte offsets (everything else is an immediate)
    // Z row reads: person row p = 32 ph + l31, chunk 2 s + half
    uint32_t zE[2], zO[2], z6[2];
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
        const int p = 32 * ph + l31, xr = (p >> 2) & 3;
        const uint32_t base = (uint32_t)(p >> 3) * (256u * LB_NKS) + 64u * (p & 7);
        zE[ph] = base + 16u * ((0 + half) ^ xr);
        zO[ph] = base + 16u * ((2 + half) ^ xr);
        z6[ph] = (uint32_t)(p >> 3) * (256u * LB_NKS) + 1536u + 32u * (p & 7) + 16u * (half ^ ((p >> 4) & 1));
    }
    // transposed reads: 16-lane group (half, gl), lane 4 q + pp of the group
    const int gl = (lane >> 4) & 1, q4 = (lane & 15) >> 2, pp = lane & 3;
    uint32_t gaB[2], ga3[2], rB[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        // GA: person row P = 32 ph + 16 s' + 8 e + 4 half + q4, latent columns 32 kt + 16 gl + 4 pp .. + 3
        gaB[e] = 64u * (4 * half + q4) + 16u * ((2 * gl + (pp >> 1)) ^ (2 * e + half)) + 8u * (pp & 1);
        // R image: item row j = 16 s + 8 half + 4 e + q4, persons 16 gl + 4 pp .. + 3 (8-byte piece 4 gl + pp, swizzled)
        rB[e] = 512u * half + 256u * e + 64u * q4 + 8u * (((4 * gl + pp) ^ (4 * e + q4) ^ half) & 7);
    }
#pragma unroll
    for (int sp2 = 0; sp2 < 2; ++sp2)                               // k-tile 3 (columns 96..111; both lane groups read them)
        ga3[sp2] = 1536u + 32u * (4 * half + q4) + 16u * ((pp >> 1) ^ sp2) + 8u * (pp & 1);
    // R image write: item row jl = 32 wave + l31, persons 8 g4 + 4 half + 0..3 = piece 2 g4 + half, swizzled
    const int jl = 32 * wave + l31;
    const uint32_t rW = (uint32_t)jl * 64u, rSw = (uint32_t)((jl & 7) ^ ((jl >> 3) & 1));

    // ---- staging of one person tile: global -> LDS DMA.  x: the tile image, 42 linear 1 KB transfers; y: as k_irt_lik_r
    const bool jfull = j0 + LB_JC <= J;                             // block-uniform: no item edge in this chunk
    for (int e = tid; e < LB_P * (LB_YS / 4); e += LB_THREADS) ((uint32_t*)Yb)[e] = 0xFEFEFEFEu;
    __syncthreads();
    auto stage_x = [&](int64_t tile, int buf) {
        const uint8_t* src = ximg + tile * LB_XT_BYTES + lane * 16;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(xbuf_l + (uint32_t)buf * LB_XT_BYTES);
#pragma unroll
        for (int u = 0; u < 11; ++u) {
            const int piece = wave + 4 * u;                         // wave-uniform
            if (piece < LB_XT_BYTES / 1024) dma16(src + piece * 1024, dst + (uint32_t)piece * 1024u);
        }
    };
    auto stage_y = [&](int64_t tile) {
        const int64_t i0 = tile * LB_P;
        const int pv = (int)((dm.nb - i0) < LB_P ? (dm.nb - i0) : LB_P);
        if (jfull) {
            for (int r8 = wave; 8 * r8 < pv; r8 += 4) {
                const int prow = 8 * r8 + (lane >> 3);
                if (prow < pv) {
                    const int64_t row = ROWS ? rows[i0 + prow] : i0 + prow;
                    dma16(y + row * J + j0 + 16 * (lane & 7), __builtin_amdgcn_readfirstlane(Yb_l + (uint32_t)(r8 * 8 * LB_YS)));
                }
            }
        } else {
            for (int r2 = wave; 2 * r2 < pv; r2 += 4) {
                const int prow = 2 * r2 + (lane >> 5), jj = j0 + 4 * (lane & 31);
                if (prow < pv && jj < J) {
                    const int64_t row = ROWS ? rows[i0 + prow] : i0 + prow;
                    dma4(y + row * J + jj, __builtin_amdgcn_readfirstlane(Yb_l + (uint32_t)(r2 * 2 * LB_YS)));
                }
            }
        }
        if (pv < LB_P)                                              // the last tile: absent persons carry no cell
            for (int e = tid; e < (LB_P - pv) * (LB_YS / 4); e += LB_THREADS) ((uint32_t*)Yb)[pv * (LB_YS / 4) + e] = 0xFEFEFEFEu;
    };

    // gx of one person half: C layout of gx^T -- lane = person 32 ph + l31, register r = latent row 32 wave + crow32(r, half).
    // - scale * x (the N(0, I) prior on x) is subtracted by ONE of the chunk workgroups per run of four rows.
    auto store_gx = [&](int ph, f32x16 gxa, int64_t i0, const char* xb) {
        const int p = 32 * ph + l31;
        const int64_t i = i0 + p;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int k0 = 32 * wave + 8 * g4 + 4 * half;
            if ((g4 % dm.groups) == g && k0 < D) {                  // chunk 4 wave + g4 of the image row, its half `half`
                const uint32_t o = lb_xoff(p, 4 * wave + g4) + 8u * half;
                const lb_u32x2 vh = *(const lb_u32x2*)(xb + o), vm = *(const lb_u32x2*)(xb + LB_PLANE + o),
                               vl = *(const lb_u32x2*)(xb + 2 * LB_PLANE + o);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t wh = vh[e >> 1], wm = vm[e >> 1], wl = vl[e >> 1];
                    const float xv = __builtin_bit_cast(float, (e & 1) ? (wh & 0xffff0000u) : (wh << 16)) +
                                     (__builtin_bit_cast(float, (e & 1) ? (wm & 0xffff0000u) : (wm << 16)) +
                                      __builtin_bit_cast(float, (e & 1) ? (wl & 0xffff0000u) : (wl << 16)));
                    gxa[4 * g4 + e] = fmaf(-dm.scale, xv, gxa[4 * g4 + e]);
                }
            }
        }
        if (i < dm.nb) {
            if (dm.gxt) {                                           // lanes = consecutive persons: 128-byte rows
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int k = 32 * wave + crow32(r, half);
                    if (k < D) gx_part[((int64_t)g * D + k) * dm.nb + i] = gxa[r];
                }
            } else {
                float* dst = gx_part + ((int64_t)g * dm.nb + i) * D;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int k = 32 * wave + crow32(r, half);
                    if (k < D) dst[k] = gxa[r];
                }
            }
        }
    };

    int64_t tile = pr;
    int buf = 0;
    f32x16 gx1 = zero16();                                          // gx of persons 32..63: stored one tile late
    int64_t i0_prev = -1;
    if (tile < n_ptiles) { stage_x(tile, 0); stage_y(tile); }
    const float sdc = dm.scale * dm.Dc;
    const int par = l31 & 1;
    for (; tile < n_ptiles; tile += dm.n_pr, buf ^= 1) {
        vx_wait_vmem();                                             // this wave's DMA of the tile has landed
        __syncthreads();                                            // S1: tile staged; R / LP of the previous tile consumed
        const char* xb = xbuf + buf * LB_XT_BYTES;
        const uint32_t xb_l = xbuf_l + (uint32_t)buf * LB_XT_BYTES;
        if (i0_prev >= 0) store_gx(1, gx1, i0_prev, xbuf + (buf ^ 1) * LB_XT_BYTES);
        gx1 = zero16();
        const int64_t i0 = tile * LB_P;
        const int64_t next = tile + dm.n_pr;
        const bool has_next = next < n_ptiles;                      // block-uniform

        // ---- operand reads
        auto z_frags = [&](int ph, int s, bf16x8& fh, bf16x8& fm, bf16x8& fl) {
            const uint32_t o = (s == 6) ? z6[ph] : (((s & 1) ? zO[ph] : zE[ph]) + 512u * (s >> 1));
            fh = *(const bf16x8*)(xb + o);
            fm = *(const bf16x8*)(xb + LB_PLANE + o);
            fl = *(const bf16x8*)(xb + 2 * LB_PLANE + o);
        };
        auto z_mma = [&](int s, f32x16& z, const bf16x8& xh, const bf16x8& xm, const bf16x8& xl) {
            z = mfma_bf16(xl, aZ[0][s], z);
            z = mfma_bf16(xh, aZ[2][s], z);
            z = mfma_bf16(xm, aZ[1][s], z);
            z = mfma_bf16(xm, aZ[0][s], z);
            z = mfma_bf16(xh, aZ[1][s], z);
            z = mfma_bf16(xh, aZ[0][s], z);
        };
        // GA step (person half ph, k-step s2 = persons 16 s2 .. + 15 of the half, latent tile kt): A = x^T by transposed reads
        auto ga_frags = [&](int ph, int s2, int kt, bf16x8& fh, bf16x8& fm, bf16x8& fl) {
            const uint32_t g0 = 256u * LB_NKS * (4 * ph + 2 * s2), g1 = g0 + 256u * LB_NKS;
            const uint32_t o0 = (kt == 3) ? g0 + ga3[s2] : g0 + gaB[0] + 512u * kt;
            const uint32_t o1 = (kt == 3) ? g1 + ga3[s2] : g1 + gaB[1] + 512u * kt;
            fh = lb_frag(lb_tr_read(xb_l + o0), lb_tr_read(xb_l + o1));
            fm = lb_frag(lb_tr_read(xb_l + LB_PLANE + o0), lb_tr_read(xb_l + LB_PLANE + o1));
            fl = lb_frag(lb_tr_read(xb_l + 2 * LB_PLANE + o0), lb_tr_read(xb_l + 2 * LB_PLANE + o1));
        };
        auto ga_mma = [&](int kt, const bf16x8& xh, const bf16x8& xm, const bf16x8& xl, const bf16x8& rh, const bf16x8& rm,
                          const bf16x8& rl) {
            ga[kt] = mfma_bf16(xl, rh, ga[kt]);
            ga[kt] = mfma_bf16(xh, rl, ga[kt]);
            ga[kt] = mfma_bf16(xm, rm, ga[kt]);
            ga[kt] = mfma_bf16(xm, rh, ga[kt]);
            ga[kt] = mfma_bf16(xh, rm, ga[kt]);
            ga[kt] = mfma_bf16(xh, rh, ga[kt]);
        };
        // gx step (person half ph, k-step s = items 16 s .. + 15 of the chunk): B = R^T by transposed reads of the R image
        auto gx_frags = [&](int ph, int s, bf16x8& fh, bf16x8& fm, bf16x8& fl) {
            const uint32_t rb = Rimg_l + (uint32_t)(ph * 3 * LB_RPLANE + 1024 * s);
            fh = lb_frag(lb_tr_read(rb + rB[0]), lb_tr_read(rb + rB[1]));
            fm = lb_frag(lb_tr_read(rb + LB_RPLANE + rB[0]), lb_tr_read(rb + LB_RPLANE + rB[1]));
            fl = lb_frag(lb_tr_read(rb + 2 * LB_RPLANE + rB[0]), lb_tr_read(rb + 2 * LB_RPLANE + rB[1]));
        };
        auto gx_mma = [&](int s, f32x16& gx, const bf16x8& rh, const bf16x8& rm, const bf16x8& rl) {
            gx = mfma_bf16(aG[2][s], rh, gx);
            gx = mfma_bf16(aG[0][s], rl, gx);
            gx = mfma_bf16(aG[1][s], rm, gx);
            gx = mfma_bf16(aG[1][s], rh, gx);
            gx = mfma_bf16(aG[0][s], rm, gx);
            gx = mfma_bf16(aG[0][s], rh, gx);
        };

        // ---- the epilogue of one register PAIR (2 i, 2 i + 1) of a person half: cells, R split, LP pair sums
        uint32_t Rp[2][3][8];                                       // [half][split][pair]: packed bf16 terms of R
        auto cell_pair = [&](auto phc, auto ic, f32x16& z) {
            constexpr int ph = decltype(phc)::value, i = decltype(ic)::value;
            float lpv[2], rv[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int r = 2 * i + e;
                const int p = 32 * ph + crow32(r, half);
                const unsigned yy = Yb[p * LB_YS + 32 * wave + l31];
                const float zz = z[r];
                float lp, dz, dc, dd;
                if (GEN) {
                    if (dm.model == 4) irt_cell<4>(zz, yy, cj, dj, omdj, lp, dz, dc, dd);
                    else irt_cell<3>(zz, yy, cj, 1.0f, 0.f, lp, dz, dc, dd);
                    gc = fmaf(dm.scale, dc, gc);
                    gd = fmaf(dm.scale, dd, gd);
                } else {
                    irt_cell<2>(zz, yy, 0.f, 1.f, 0.f, lp, dz, dc, dd);
                }
                rv[e] = sdc * dz;
                lpv[e] = lp + dpp_mov0<0xB1, 0xF>(lp);              // + the neighbouring item's term (quad_perm [1,0,3,2])
            }
            lb_split_pair(rv[0], rv[1], Rp[ph][0][i], Rp[ph][1][i], Rp[ph][2][i]);
            // item pair (l31 >> 1): the even lane stores register 2 i, the odd lane register 2 i + 1
            const int pw = 32 * ph + crow32(2 * i, half) + par;
            LPp[pw * 64 + 16 * wave + (l31 >> 1)] = par ? lpv[1] : lpv[0];
            if constexpr (i & 1) {                                  // registers 4 g4 .. 4 g4 + 3 are complete: 8 bytes per plane
                constexpr int g4 = i >> 1;
                char* wb = Rimg + ph * 3 * LB_RPLANE + rW + 8u * (((2 * g4 + half) ^ rSw) & 7);
#pragma unroll
                for (int sp = 0; sp < 3; ++sp) {
                    const lb_u32x2 w = {Rp[ph][sp][i - 1], Rp[ph][sp][i]};
                    *(lb_u32x2*)(wb + sp * LB_RPLANE) = w;
                }
            }
        };
        auto rfrag = [&](int ph, int sp, int s2) -> bf16x8 {        // B fragment of k-step s2: registers 8 s2 .. 8 s2 + 7
            const lb_u32x4 qv = {Rp[ph][sp][4 * s2], Rp[ph][sp][4 * s2 + 1], Rp[ph][sp][4 * s2 + 2], Rp[ph][sp][4 * s2 + 3]};
            return __builtin_bit_cast(bf16x8, qv);
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;

        // ---- A: Z of persons 0..31
        f32x16 z0 = zero16(), z1 = zero16();
        {
            bf16x8 ch, cm, cl, nh, nm, nl;
            z_frags(0, 0, ch, cm, cl);
            static_for<LB_NKS>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                if constexpr (s + 1 < LB_NKS) z_frags(0, s + 1, nh, nm, nl); else z_frags(1, 0, nh, nm, nl);
                z_mma(s, z0, ch, cm, cl);
                ch = nh; cm = nm; cl = nl;
                __builtin_amdgcn_sched_barrier(0);
            });
            // ---- B: Z of persons 32..63 | the epilogue of persons 0..31
            static_for<LB_NKS>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                if constexpr (s + 1 < LB_NKS) z_frags(1, s + 1, nh, nm, nl);
                z_mma(s, z1, ch, cm, cl);
                if constexpr (s + 1 < LB_NKS) { ch = nh; cm = nm; cl = nl; }
                cell_pair(I0{}, sc, z0);
                if constexpr (s == LB_NKS - 1) cell_pair(I0{}, std::integral_constant<int, 7>{}, z0);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        __syncthreads();                                            // S2: R image / LP of persons 0..31 complete
        // ---- C: GA and gx of persons 0..31 | the epilogue of persons 32..63
        f32x16 gx0 = zero16();
        {
            bf16x8 ch, cm, cl, nh, nm, nl;
            ga_frags(0, 0, 0, ch, cm, cl);
            static_for<8>([&](auto uc) {                            // (s2, kt) = (u >> 2, u & 3)
                constexpr int u = decltype(uc)::value, s2 = u >> 2, kt = u & 3;
                if constexpr (u + 1 < 8) ga_frags(0, (u + 1) >> 2, (u + 1) & 3, nh, nm, nl); else gx_frags(0, 0, nh, nm, nl);
                ga_mma(kt, ch, cm, cl, rfrag(0, 0, s2), rfrag(0, 1, s2), rfrag(0, 2, s2));
                ch = nh; cm = nm; cl = nl;
                if constexpr ((u & 1) == 0) cell_pair(I1{}, std::integral_constant<int, u / 2>{}, z1);
                __builtin_amdgcn_sched_barrier(0);
            });
            static_for<8>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                if constexpr (s + 1 < 8) gx_frags(0, s + 1, nh, nm, nl);
                gx_mma(s, gx0, ch, cm, cl);
                if constexpr (s + 1 < 8) { ch = nh; cm = nm; cl = nl; }
                if constexpr ((s & 1) == 0) cell_pair(I1{}, std::integral_constant<int, 4 + s / 2>{}, z1);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        __syncthreads();                                            // S3: R image of persons 32..63 and all LP rows complete
        if (has_next) { stage_x(next, buf ^ 1); stage_y(next); }    // DMA of the next tile flies under the 96 MFMAs of D
        // ---- D: gx and GA of persons 32..63 | ll reduce, gx store of persons 0..31
        {
            bf16x8 ch, cm, cl, nh, nm, nl;
            gx_frags(1, 0, ch, cm, cl);
            static_for<8>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                if constexpr (s + 1 < 8) gx_frags(1, s + 1, nh, nm, nl); else ga_frags(1, 0, 0, nh, nm, nl);
                gx_mma(s, gx1, ch, cm, cl);
                ch = nh; cm = nm; cl = nl;
                if constexpr (s == 1) {
                    // per-person log-lik of this chunk (+ the N(0, I) prior once, in chunk 0): thread = (person, quarter)
                    const int p = tid >> 2, qq = tid & 3;
                    float sll = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const f32x4 v = *(const f32x4*)(LPp + p * 64 + 16 * qq + 4 * ((e + p) & 3));
                        sll += (v[0] + v[1]) + (v[2] + v[3]);
                    }
                    sll += dpp_mov0<0xB1, 0xF>(sll);
                    sll += dpp_mov0<0x4E, 0xF>(sll);
                    if (qq == 0 && i0 + p < dm.nb) {
                        if (g == 0) sll = fmaf(-0.5f, xsq[i0 + p], sll);
                        ll_part[(int64_t)g * dm.nb + i0 + p] = sll;
                    }
                }
                if constexpr (s == 4) store_gx(0, gx0, i0, xb);
                __builtin_amdgcn_sched_barrier(0);
            });
            static_for<8>([&](auto uc) {
                constexpr int u = decltype(uc)::value, s2 = u >> 2, kt = u & 3;
                if constexpr (u + 1 < 8) ga_frags(1, (u + 1) >> 2, (u + 1) & 3, nh, nm, nl);
                ga_mma(kt, ch, cm, cl, rfrag(1, 0, s2), rfrag(1, 1, s2), rfrag(1, 2, s2));
                if constexpr (u + 1 < 8) { ch = nh; cm = nm; cl = nl; }
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        i0_prev = i0;
    }
    if (i0_prev >= 0) store_gx(1, gx1, i0_prev, xbuf + (buf ^ 1) * LB_XT_BYTES);
    // ---- item-gradient slab of this person range: GA C layout = rows k = 32 kt + crow32(r, half), column = this lane's item
    float* slab = slabs + (int64_t)pr * dm.slab_len;
    if (jv) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = 32 * kt + crow32(r, half);
                if (k <= D) slab[(int64_t)k * J + jw] = ga[kt][r];  // k == D lands in the b segment
            }
    }
    if (GEN) {
        gc += __shfl_xor(gc, 32, 64);
        gd += __shfl_xor(gd, 32, 64);
        if (jv && half == 0) {
            slab[(int64_t)(D + 1) * J + jw] = gc;
            slab[(int64_t)(D + 2) * J + jw] = gd;
        }
    }
}
