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
// The loop is software-pipelined ACROSS person tiles so that the cell epilogue (vector work) always runs beside 96
// MFMAs that do not depend on it:
//     top barrier | Z of persons 32..63 (t)      | GA, gx of persons 0..31 (t)  beside  cells of persons 32..63 (t)
//     mid barrier | Z of persons 0..31 (t + 1)   | gx, GA of persons 32..63 (t) beside  cells of persons 0..31 (t + 1)
// (two barriers per tile; two x buffers suffice because tile t + 1 is first read after the mid barrier of tile t).
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
#define LB_YT_BYTES (LB_JC * LB_P)                 // response bytes of one tile, item-major: [128 items][64 persons]
#define LB_LDS_BYTES (2 * LB_XT_BYTES + 6 * LB_RPLANE + 2 * LB_YT_BYTES + LB_P * 32 * 4)     // 159744

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

// x fp32 [nb][D] -> tile images (the three bf16 terms of x_aug); persons past nb: all-zero rows
// only_if (or null): the launch is the stand-by of the fp16 kernel (k_irt_lik_h.hip) and returns unless that word is set
__global__ __launch_bounds__(256) void k_lik_ximg(int D, int64_t nb, const float* __restrict__ x, uint8_t* __restrict__ img,
                                                  const uint32_t* __restrict__ only_if = nullptr) {
    if (only_if && *only_if == 0u) return;
    const int64_t n_tiles = (nb + LB_P - 1) / LB_P;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {      // (a stand-by launch keeps its grid small)
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
    }
}

typedef short lb_s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t lb_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t lb_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lb_lds;                 // LDS pointers stay in their address space: constant
                                                                       // offsets then fold into the DS instruction's offset field

// transposed LDS read (ds_read_b64_tr_b16): per 16-lane group a block of 4 rows x 16 columns of 16-bit elements; lane
// 4q + p of the group supplies the address of row q, columns 4p..4p+3; lane i receives column i, row q in element q
__device__ __forceinline__ lb_u32x2 lb_tr_read(lb_lds* p) {
    const lb_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) lb_s16x4*)p);
    return __builtin_bit_cast(lb_u32x2, v);
}
__device__ __forceinline__ bf16x8 lb_frag(lb_u32x2 lo, lb_u32x2 hi) {
    const lb_u32x4 q = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8, q);
}
__device__ __forceinline__ bf16x8 lb_read128(lb_lds* p) { return *(__attribute__((address_space(3))) const bf16x8*)p; }

// two fp32 values -> their TWO leading bf16 terms (round to nearest each: v = hi + mid to 2^-17 relative, unbiased), packed
// pairwise (element 0 in the low half); one v_cvt_pk_bf16_f32 per term.  R = dlogp/dz only feeds gx and GA, which end in
// sums over the persons (encoder and item gradients) whose fp32 accumulation noise is far above 2^-17 / sqrt(N): the third
// term, and with it one product in six of gx and GA, is not spent.  (x and a keep three terms: Z is exact to fp32.)
__device__ __forceinline__ void lb_split_pair(float a, float b, uint32_t& ph, uint32_t& pm) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    // the conversions are paired (one instruction per term), the residuals are scalar on purpose: packed f32 adds cost
    // more issue time beside MFMAs than two plain ones (MI355X_MICROARCH.md, 'price of one filler')
    ph = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){a, b}, bf16x2));
    float a1 = a - __builtin_bit_cast(float, ph << 16), b1 = b - __builtin_bit_cast(float, ph & 0xffff0000u);
    asm("" : "+v"(a1), "+v"(b1));
    pm = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){a1, b1}, bf16x2));
}

// scheduling hint for one region: interleave its MFMAs with the vector work of the region (a bf16 MFMA holds the vector
// issue port for 8 of its 32 cycles only; work placed AFTER a burst of MFMAs does not overlap them)
#define LB_MASK_VALU 0x002
#define LB_MASK_MFMA 0x008
#define LB_MASK_DSR 0x100
template <int N_MFMA, int VALU_PER, int DSR_FIRST>
__device__ __forceinline__ void lb_interleave() {
    if constexpr (DSR_FIRST > 0) __builtin_amdgcn_sched_group_barrier(LB_MASK_DSR, DSR_FIRST, 0);
#pragma unroll
    for (int i = 0; i < N_MFMA; ++i) {
        __builtin_amdgcn_sched_group_barrier(LB_MASK_MFMA, 1, 0);
        if constexpr (VALU_PER > 0) __builtin_amdgcn_sched_group_barrier(LB_MASK_VALU, VALU_PER, 0);
    }
}

// partial gx / ll layout (dimension-major, padded so that no store needs a guard):
//   gx_part[groups][LB_DP][nbp], ll_part[groups][nbp], nbp = person count rounded up to whole tiles.
// They hold the LIKELIHOOD part only; k_lik_reduce_parts adds the N(0, I) prior on x (- scale x, - 0.5 |x|^2).
#define LB_DP 128

// yT: the responses item-major, [J + 1][yT_stride] bytes: row j = item j over the batch rows (columns past nb: 254), row J
// all 254 ("outside the problem": what the lanes of items past J read); yT_stride % 64 == 0, yT_stride >= nbp.
// ABL: ablation bits for tools/likb_test.hip only (0 in the library): 1 no cell math, 2 no barriers, 4 no DMA,
// 8 no scheduling hints, 16 no output stores, 32 phase stamps
template <int GEN, int ABL = 0>
__global__ __launch_bounds__(LB_THREADS, 1) void k_irt_lik_b(
    LikBDims dm, const uint8_t* __restrict__ yT, int64_t yT_stride, const uint8_t* __restrict__ ximg,
    const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c_un,
    const float* __restrict__ d_un, float* __restrict__ gx_part, float* __restrict__ ll_part, float* __restrict__ slabs,
    long long* __restrict__ stamps = nullptr /*ABL & 32 only*/, const uint32_t* __restrict__ only_if = nullptr /*as k_lik_ximg*/) {
    extern __shared__ __attribute__((aligned(16))) char smem_lb[];
    if (only_if && *only_if == 0u) return;                          // uniform: no barrier has been passed
    const int D = dm.D, J = dm.J;
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
    auto stamp = [&](int idx) {
        if constexpr (ABL & 32) {
            const long long now = (long long)__builtin_amdgcn_s_memtime();
            tacc[idx] += now - tlast;
            tlast = now;
        }
    };
    lb_lds* const smem = (lb_lds*)smem_lb;
    lb_lds* const Rimg = smem + 2 * LB_XT_BYTES;                    // [2 person halves][3 planes][128 items][64 B]
    lb_lds* const Yb = Rimg + 6 * LB_RPLANE;                        // [2][128 items][64 persons] response bytes
    lb_lds* const LPb = Yb + 2 * LB_YT_BYTES;                       // [64 persons][32 item quads] fp32
    const uint32_t smem_l = (uint32_t)(size_t)smem;
    const uint32_t Yb_l = smem_l + 2 * LB_XT_BYTES + 6 * LB_RPLANE;
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
    const int jl = 32 * wave + l31;                                 // this lane's item within the chunk
    const int jw = j0 + jl;                                         // ... in the problem (Z / epilogue / GA column)
    const bool jv = jw < J;
    const int64_t n_ptiles = (dm.nb + LB_P - 1) / LB_P;
    const int64_t nbp = n_ptiles * LB_P;

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

    // ---- per-lane LDS byte offsets inside a tile image / the R image (everything else is an immediate)
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
    // transposed reads: 16-lane group (half, gl), lane 4 q4 + pp of the group
    const int gl = (lane >> 4) & 1, q4 = (lane & 15) >> 2, pp = lane & 3;
    uint32_t gaB[2], ga3[2], rB[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        // GA: person row P = 32 ph + 16 s2 + 8 e + 4 half + q4, latent columns 32 kt + 16 gl + 4 pp .. + 3
        gaB[e] = 64u * (4 * half + q4) + 16u * ((2 * gl + (pp >> 1)) ^ (2 * e + half)) + 8u * (pp & 1);
        // k-tile 3 (columns 96..111; both lane groups read them: rows 112..127 of GA are never stored), e = s2 here
        ga3[e] = 1536u + 32u * (4 * half + q4) + 16u * ((pp >> 1) ^ e) + 8u * (pp & 1);
        // R image: item row j = 16 s + 8 half + 4 e + q4, persons 16 gl + 4 pp .. + 3 (8-byte piece 4 gl + pp, swizzled)
        rB[e] = 512u * half + 256u * e + 64u * q4 + 8u * (((4 * gl + pp) ^ (4 * e + q4) ^ half) & 7);
    }
    // R image write: item row jl, persons 8 g4 + 4 half + 0..3 = piece 2 g4 + half, swizzled
    const uint32_t rSw = (uint32_t)((jl & 7) ^ ((jl >> 3) & 1));
    lb_lds* const rWp = Rimg + jl * 64;
    lb_lds* const rRp0 = Rimg + rB[0];
    lb_lds* const rRp1 = Rimg + rB[1];
    lb_lds* const yP0 = Yb + jl * 64;                               // this lane's item row: + 32 ph (+ buffer)
    const int qc = l31 & 3;                                         // lane within its item quad: stores register 4 g4 + qc
    lb_lds* const lpW = LPb + 4 * ((4 * half + qc) * 32 + 8 * wave + (l31 >> 2));    // + 128 * (32 ph + 8 g4)
    lb_lds* const lpR = LPb + 4 * ((tid >> 3) * 32 + 4 * (tid & 7));                  // + 4096 * ph
    // gx transposition slices: rows 0..15 / 16..31 of this wave's 32 latent rows sit in ITS OWN rows of planes 0 / 1 of the
    // R image of the person half (2 KB each), which no other wave writes
    lb_lds* const tW = Rimg + 2048 * wave + 128 * (4 * half) + 4 * l31;      // + 128 ((r & 3) + 8 ((r >> 2) & 1)) + 8192 (r >> 3)
    lb_lds* const tR = Rimg + 2048 * wave + 128 * (lane >> 3) + 16 * (lane & 7);   // + 1024 (u & 1) + 8192 (u >> 1)
    float* const gxw = gx_part + ((int64_t)g * LB_DP + 32 * wave + (lane >> 3)) * nbp + 4 * (lane & 7);

    // ---- staging: global -> LDS DMA, ONE 1 KB transfer per call (the LDS-DMA path of a CU moves ~1 KB per ~100 cycles:
    // the transfers of the next tile are issued one per step under the MFMAs).  x: piece wave + 4 u of the 42 of the tile
    // image (u = 0..10; past the end: a repeat of the last).  y: 16 item rows x 64 persons per transfer, u = 0, 1.
    auto stage_x_piece = [&](int64_t tile, int buf, int u) {
        if constexpr (ABL & (4 | 64)) return;
        int piece = wave + 4 * u;                                   // wave-uniform
        piece = piece < LB_XT_BYTES / 1024 ? piece : LB_XT_BYTES / 1024 - 1;
        dma16(ximg + tile * LB_XT_BYTES + lane * 16 + piece * 1024,
              __builtin_amdgcn_readfirstlane(smem_l + (uint32_t)buf * LB_XT_BYTES + (uint32_t)piece * 1024u));
    };
    auto stage_y_piece = [&](int64_t tile, int buf, int u) {
        if constexpr (ABL & (4 | 128)) return;
        const int piece = wave + 4 * u;                             // rows 16 piece .. + 15 of the chunk
        int jr = j0 + 16 * piece + (lane >> 2);
        jr = jr < J ? jr : J;                                       // items past J: the all-254 row
        dma16(yT + (int64_t)jr * yT_stride + tile * LB_P + 16 * (lane & 3),
              __builtin_amdgcn_readfirstlane(Yb_l + (uint32_t)buf * LB_YT_BYTES + (uint32_t)piece * 1024u));
    };

    // gx of one person half: C layout of gx^T (lane = person, register r = latent row 32 wave + crow32(r, half)) -> rows of 32
    // persons through the transposition slices -> four 16-byte stores per lane (128-byte rows; padded, no guards)
    auto store_gx = [&](int ph, const f32x16& gxa, int64_t i0) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            *(__attribute__((address_space(3))) float*)(tW + ph * 3 * LB_RPLANE + 128 * ((r & 3) + 8 * ((r >> 2) & 1)) + 8192 * (r >> 3)) = gxa[r];
        __builtin_amdgcn_wave_barrier();
        if constexpr (!(ABL & 16)) {
            float* dst = gxw + i0 + 32 * ph;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (32 * wave + 8 * u < D) {                        // uniform; rows D .. of the last run land in the row padding
                    const f32x4 v = *(__attribute__((address_space(3))) const f32x4*)(tR + ph * 3 * LB_RPLANE + 1024 * (u & 1) + 8192 * (u >> 1));
                    *(f32x4*)(dst + (int64_t)(8 * u) * nbp) = v;
                }
            }
        }
    };
    // per-person log-lik of this chunk, persons of one half: thread = (person, eighth of the 32 item quads); the eight lanes of
    // a person end up with the same total and store it to the same word
    auto ll_reduce = [&](int ph, int64_t i0) {
        const f32x4 v = *(__attribute__((address_space(3))) const f32x4*)(lpR + 4096 * ph);
        float sll = (v[0] + v[1]) + (v[2] + v[3]);
        sll += dpp_mov0<0xB1, 0xF>(sll);
        sll += dpp_mov0<0x4E, 0xF>(sll);
        sll += dpp_mov0<0x141, 0xF>(sll);                           // row_half_mirror: the other quad of the eight lanes
        if constexpr (!(ABL & 16)) ll_part[(int64_t)g * nbp + i0 + 32 * ph + (tid >> 3)] = sll;
    };

    const float sdc = dm.scale * dm.Dc;
    uint32_t Rp[2][2][8];                                           // [half][term][pair]: packed bf16 terms of R
    float lpq[4];                                                   // log-lik terms of the current run of four registers
    f32x16 z0 = zero16();                                           // Z of persons 0..31 of the CURRENT tile (made one tile early)
    f32x16 gx1 = zero16();                                          // gx of persons 32..63: stored one tile late
    int64_t i0_prev = -1;
    int64_t tile = pr;
    int buf = 0;

    // ---- operand reads and products (xb: tile image base)
    auto z_frags = [&](lb_lds* xb, int ph, int s, bf16x8& fh, bf16x8& fm, bf16x8& fl) {
        lb_lds* o = xb + ((s == 6) ? z6[ph] : (((s & 1) ? zO[ph] : zE[ph]) + 512 * (s >> 1)));
        fh = lb_read128(o);
        fm = lb_read128(o + LB_PLANE);
        fl = lb_read128(o + 2 * LB_PLANE);
    };
    auto z_mma = [&](int s, f32x16& z, const bf16x8& xh, const bf16x8& xm, const bf16x8& xl) {
        z = mfma_bf16(xl, aZ[0][s], z);
        z = mfma_bf16(xh, aZ[2][s], z);
        z = mfma_bf16(xm, aZ[1][s], z);
        z = mfma_bf16(xm, aZ[0][s], z);
        z = mfma_bf16(xh, aZ[1][s], z);
        z = mfma_bf16(xh, aZ[0][s], z);
    };
    // GA step u = (s2 = u >> 2: persons 16 s2 .. + 15 of the half, kt = u & 3: latent tile): A = x^T by transposed reads
    auto ga_frags = [&](lb_lds* xb, int ph, int u, bf16x8& fh, bf16x8& fm, bf16x8& fl) {
        const int s2 = u >> 2, kt = u & 3;
        const int g0 = 256 * LB_NKS * (4 * ph + 2 * s2), g1 = g0 + 256 * LB_NKS;
        lb_lds* o0 = xb + ((kt == 3) ? ga3[s2] + g0 : gaB[0] + g0 + 512 * kt);
        lb_lds* o1 = xb + ((kt == 3) ? ga3[s2] + g1 : gaB[1] + g1 + 512 * kt);
        fh = lb_frag(lb_tr_read(o0), lb_tr_read(o1));
        fm = lb_frag(lb_tr_read(o0 + LB_PLANE), lb_tr_read(o1 + LB_PLANE));
        fl = lb_frag(lb_tr_read(o0 + 2 * LB_PLANE), lb_tr_read(o1 + 2 * LB_PLANE));
    };
    auto rfrag = [&](int ph, int sp, int s2) -> bf16x8 {            // B fragment of k-step s2: registers 8 s2 .. 8 s2 + 7
        const lb_u32x4 qv = {Rp[ph][sp][4 * s2], Rp[ph][sp][4 * s2 + 1], Rp[ph][sp][4 * s2 + 2], Rp[ph][sp][4 * s2 + 3]};
        return __builtin_bit_cast(bf16x8, qv);
    };
    auto ga_mma = [&](int ph, int u, const bf16x8& xh, const bf16x8& xm, const bf16x8& xl) {
        const int s2 = u >> 2, kt = u & 3;
        const bf16x8 rh = rfrag(ph, 0, s2), rm = rfrag(ph, 1, s2);
        ga[kt] = mfma_bf16(xl, rh, ga[kt]);
        ga[kt] = mfma_bf16(xm, rm, ga[kt]);
        ga[kt] = mfma_bf16(xm, rh, ga[kt]);
        ga[kt] = mfma_bf16(xh, rm, ga[kt]);
        ga[kt] = mfma_bf16(xh, rh, ga[kt]);
    };
    // gx step (person half ph, k-step s = items 16 s .. + 15 of the chunk): B = R^T by transposed reads of the R image
    auto gx_frags = [&](int ph, int s, bf16x8& fh, bf16x8& fm, bf16x8& fl) {
        const int o = ph * 3 * LB_RPLANE + 1024 * s;
        fh = lb_frag(lb_tr_read(rRp0 + o), lb_tr_read(rRp1 + o));
        fm = lb_frag(lb_tr_read(rRp0 + o + LB_RPLANE), lb_tr_read(rRp1 + o + LB_RPLANE));
        (void)fl;                                                   // R has two terms
    };
    auto gx_mma = [&](int s, f32x16& gx, const bf16x8& rh, const bf16x8& rm, const bf16x8& rl) {
        (void)rl;
        gx = mfma_bf16(aG[2][s], rh, gx);
        gx = mfma_bf16(aG[1][s], rm, gx);
        gx = mfma_bf16(aG[1][s], rh, gx);
        gx = mfma_bf16(aG[0][s], rm, gx);
        gx = mfma_bf16(aG[0][s], rh, gx);
    };

    // ---- the epilogue of one register PAIR (2 i, 2 i + 1) of a person half: cells, R split, R image, LP quad sums.
    // yq: the 32 response bytes of (this lane's item, persons of the half) = two 16-byte reads; live: 0 for the phantom tile
    // past the end (its cells are computed on stale data; only the 3PL / 4PL accumulators could see them)
    auto cell_pair = [&](auto phc, auto ic, f32x16& z, const lb_u32x4& yq0, const lb_u32x4& yq1, float scale_live) {
        constexpr int ph = decltype(phc)::value, i = decltype(ic)::value;
        constexpr int g4 = i >> 1;                                  // run of four registers = persons 8 g4 + 4 half + 0..3
        const uint32_t wlo = (g4 < 2 ? yq0 : yq1)[2 * (g4 & 1)], whi = (g4 < 2 ? yq0 : yq1)[2 * (g4 & 1) + 1];
        const uint32_t yw = half ? whi : wlo;
        float rv[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int r = 2 * i + e;
            const float yf = (float)((yw >> (8 * (r & 3))) & 0xffu);
            const float zz = z[r];
            float lp, dz, dc, dd;
            if constexpr (ABL & 1) {
                lp = zz; dz = zz + yf; dc = 0.f; dd = 0.f;
            } else if (GEN) {
                if (dm.model == 4) irt_cell_f<4>(zz, yf, cj, dj, omdj, lp, dz, dc, dd);
                else irt_cell_f<3>(zz, yf, cj, 1.0f, 0.f, lp, dz, dc, dd);
                gc = fmaf(scale_live, dc, gc);
                gd = fmaf(scale_live, dd, gd);
            } else {
                irt_cell_f<2>(zz, yf, 0.f, 1.f, 0.f, lp, dz, dc, dd);
            }
            rv[e] = sdc * dz;
            const float l2 = lp + dpp_mov0<0xB1, 0xF>(lp);          // + the terms of the three other items of the quad
            lpq[2 * (i & 1) + e] = l2 + dpp_mov0<0x4E, 0xF>(l2);
        }
        lb_split_pair(rv[0], rv[1], Rp[ph][0][i], Rp[ph][1][i]);
        if constexpr (i & 1) {                                      // registers 4 g4 .. 4 g4 + 3 are complete
            // item quad (l31 >> 2): lane qc of the quad stores register 4 g4 + qc = person 32 ph + 8 g4 + 4 half + qc
            const float t01 = (qc & 1) ? lpq[1] : lpq[0], t23 = (qc & 1) ? lpq[3] : lpq[2];
            *(__attribute__((address_space(3))) float*)(lpW + 128 * (32 * ph + 8 * g4)) = (qc & 2) ? t23 : t01;
            lb_lds* wb = rWp + ph * 3 * LB_RPLANE + 8u * (((2 * g4 + half) ^ rSw) & 7);
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                const lb_u32x2 w = {Rp[ph][sp][i - 1], Rp[ph][sp][i]};
                *(__attribute__((address_space(3))) lb_u32x2*)(wb + sp * LB_RPLANE) = w;
            }
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    constexpr bool HINT = !(ABL & 8);

    // ---- prologue: stage the first tile, Z and cells of its persons 0..31 (no overlap: once per workgroup)
    if (tile < n_ptiles) {
#pragma unroll
        for (int u = 0; u < 11; ++u) stage_x_piece(tile, 0, u);
        stage_y_piece(tile, 0, 0);
        stage_y_piece(tile, 0, 1);
        vx_wait_vmem();
        __syncthreads();
        bf16x8 fh, fm, fl;
        static_for<LB_NKS>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            z_frags(smem, 0, s, fh, fm, fl);
            z_mma(s, z0, fh, fm, fl);
        });
        const lb_u32x4 yq0 = *(__attribute__((address_space(3))) const lb_u32x4*)(yP0), yq1 = *(__attribute__((address_space(3))) const lb_u32x4*)(yP0 + 16);
        static_for<8>([&](auto ic) { cell_pair(I0{}, ic, z0, yq0, yq1, dm.scale); });
    }
    if constexpr (ABL & 32) tlast = (long long)__builtin_amdgcn_s_memtime();
    for (; tile < n_ptiles; tile += dm.n_pr, buf ^= 1) {
        if constexpr (!(ABL & 2)) __syncthreads();                  // TOP: R image / LP of persons 0..31 complete; the other
        stamp(0);                                                   //      x / y buffers and the R image of persons 32..63 are free
        lb_lds* const xb = smem + buf * LB_XT_BYTES;
        lb_lds* const xn = smem + (buf ^ 1) * LB_XT_BYTES;
        const int64_t i0 = tile * LB_P;
        const int64_t next = tile + dm.n_pr;
        const bool has_next = next < n_ptiles;                      // block-uniform
        const int64_t nx = has_next ? next : tile;                  // past the last tile: a harmless reload of this one
        const float scale_next = has_next ? dm.scale : 0.f;
        f32x16 z1 = zero16(), gx0 = zero16();
        bf16x8 ch, cm, cl, nh, nm, nl;

        // ---- phase 1: Z of persons 32..63 | deferred gx store of the previous tile, log-lik of persons 0..31
        z_frags(xb, 1, 0, ch, cm, cl);
        static_for<LB_NKS>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            if constexpr (s + 1 < LB_NKS) z_frags(xb, 1, s + 1, nh, nm, nl); else ga_frags(xb, 0, 0, nh, nm, nl);
            z_mma(s, z1, ch, cm, cl);
            ch = nh; cm = nm; cl = nl;
            stage_x_piece(nx, buf ^ 1, s);
            if constexpr (s == 1) { if (i0_prev >= 0) store_gx(1, gx1, i0_prev); }
            if constexpr (s == 3) ll_reduce(0, i0);
            __builtin_amdgcn_sched_barrier(0);
        });
        gx1 = zero16();
        stamp(1);
        // ---- phase 2: GA and gx of persons 0..31 | the epilogue of persons 32..63
        {
            lb_lds* const yP = yP0 + buf * LB_YT_BYTES + 32;
            const lb_u32x4 yq0 = *(__attribute__((address_space(3))) const lb_u32x4*)(yP), yq1 = *(__attribute__((address_space(3))) const lb_u32x4*)(yP + 16);
            static_for<4>([&](auto dc_) {                           // two GA steps + one cell pair per region
                constexpr int d2 = decltype(dc_)::value, u = 2 * d2;
                ga_frags(xb, 0, u + 1, nh, nm, nl);
                ga_mma(0, u, ch, cm, cl);
                if constexpr (u + 2 < 8) ga_frags(xb, 0, u + 2, ch, cm, cl); else gx_frags(0, 0, ch, cm, cl);
                ga_mma(0, u + 1, nh, nm, nl);
                cell_pair(I1{}, dc_, z1, yq0, yq1, dm.scale);
                stage_x_piece(nx, buf ^ 1, 7 + d2);
                if constexpr (HINT) lb_interleave<10, 8, 6>();
                __builtin_amdgcn_sched_barrier(0);
            });
            static_for<4>([&](auto dc_) {                           // two gx steps + one cell pair per region
                constexpr int d2 = decltype(dc_)::value, s = 2 * d2;
                gx_frags(0, s + 1, nh, nm, nl);
                gx_mma(s, gx0, ch, cm, cl);
                if constexpr (s + 2 < 8) gx_frags(0, s + 2, ch, cm, cl);
                gx_mma(s + 1, gx0, nh, nm, nl);
                cell_pair(I1{}, std::integral_constant<int, 4 + d2>{}, z1, yq0, yq1, dm.scale);
                if constexpr (d2 < 2) stage_y_piece(nx, buf ^ 1, d2);
                if constexpr (HINT) lb_interleave<10, 8, 6>();
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        stamp(2);
        vx_wait_vmem();                                             // this wave's transfers of the next tile have landed
        stamp(3);
        if constexpr (!(ABL & 2)) __syncthreads();                  // MID: R image / LP of persons 32..63 complete; next tile staged
        stamp(4);
        // ---- phase 3: Z of persons 0..31 of the NEXT tile | log-lik of persons 32..63, gx store of persons 0..31
        z0 = zero16();
        z_frags(xn, 0, 0, ch, cm, cl);
        static_for<LB_NKS>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            if constexpr (s + 1 < LB_NKS) z_frags(xn, 0, s + 1, nh, nm, nl); else gx_frags(1, 0, nh, nm, nl);
            z_mma(s, z0, ch, cm, cl);
            ch = nh; cm = nm; cl = nl;
            if constexpr (s == 1) ll_reduce(1, i0);
            if constexpr (s == 3) store_gx(0, gx0, i0);
            __builtin_amdgcn_sched_barrier(0);
        });
        stamp(5);
        // ---- phase 4: gx and GA of persons 32..63 | the epilogue of persons 0..31 of the NEXT tile
        {
            lb_lds* const yP = yP0 + (buf ^ 1) * LB_YT_BYTES;
            const lb_u32x4 yq0 = *(__attribute__((address_space(3))) const lb_u32x4*)(yP), yq1 = *(__attribute__((address_space(3))) const lb_u32x4*)(yP + 16);
            static_for<4>([&](auto dc_) {
                constexpr int d2 = decltype(dc_)::value, s = 2 * d2;
                gx_frags(1, s + 1, nh, nm, nl);
                gx_mma(s, gx1, ch, cm, cl);
                if constexpr (s + 2 < 8) gx_frags(1, s + 2, ch, cm, cl); else ga_frags(xb, 1, 0, ch, cm, cl);
                gx_mma(s + 1, gx1, nh, nm, nl);
                cell_pair(I0{}, dc_, z0, yq0, yq1, scale_next);
                if constexpr (HINT) lb_interleave<10, 8, 6>();
                __builtin_amdgcn_sched_barrier(0);
            });
            static_for<4>([&](auto dc_) {
                constexpr int d2 = decltype(dc_)::value, u = 2 * d2;
                ga_frags(xb, 1, u + 1, nh, nm, nl);
                ga_mma(1, u, ch, cm, cl);
                if constexpr (u + 2 < 8) ga_frags(xb, 1, u + 2, ch, cm, cl);
                ga_mma(1, u + 1, nh, nm, nl);
                cell_pair(I0{}, std::integral_constant<int, 4 + d2>{}, z0, yq0, yq1, scale_next);
                if constexpr (HINT) lb_interleave<10, 8, 6>();
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        i0_prev = i0;
        stamp(6);
    }
    if constexpr (ABL & 32) {
        if (stamps && tid == 0 && blockIdx.x < 1024)
            for (int q = 0; q < 8; ++q) stamps[blockIdx.x * 8 + q] = tacc[q];
    }
    if constexpr (!(ABL & 2)) __syncthreads();                      // every wave is done reading the R image of persons 32..63
    if (i0_prev >= 0) store_gx(1, gx1, i0_prev);
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

// gxT[k][i] = sum_g gx_part[g][k][i] - scale x[i][k],  ll[i] = sum_g ll_part[g][i] - 0.5 |x_i|^2  (the N(0, I) prior on
// x, vi.py:607-613).  One block per 64 persons: x rows are read coalesced and turned through LDS; a thread owns four
// consecutive persons (16-byte accesses of the partial rows) and every 16th latent row: 4 groups x 7 rows of independent loads.
__global__ __launch_bounds__(256) void k_lik_reduce_parts(const float* __restrict__ gx_part, const float* __restrict__ ll_part,
                                                         const float* __restrict__ x, int groups, int D, int64_t nb, int64_t nbp,
                                                         float scale, float* __restrict__ gxT, float* __restrict__ ll,
                                                         const float* __restrict__ epsT = nullptr, const float* __restrict__ ldT = nullptr,
                                                         float* __restrict__ gdT = nullptr /*fused: gxT epsT ldT + scale*/,
                                                         uint32_t* __restrict__ opmax = nullptr /*with gdT: float bits of the largest
                                                         |gx|, |gd|, |eps| (integer atomicMax: order-free); cleared by the caller*/) {
    // the x tile [64][XSS]: XSS = D | 1 floats a person (odd: the column reads of a wave fall on 32 banks) -- sized by the launch,
    // so that D = 100 holds six blocks a CU instead of the four of a fixed [64][129] (the likelihood phase of the 1M step
    // 1.896 -> 1.870 ms in alternating runs on one box)
    extern __shared__ __attribute__((aligned(16))) float xs[];
    const int XSS = D | 1;
    const int64_t i0 = (int64_t)blockIdx.x * 64;
    const int pv = (int)((nb - i0) < 64 ? (nb - i0) : 64);
    const int tid = threadIdx.x;
    if (D % 4 == 0) {                                               // whole 16-byte pieces of the x rows (aligned: i0 D % 4 == 0)
        const int c4 = D >> 2;
        for (int e = tid; e < pv * c4; e += 256) {
            const int p = e / c4, c = e - p * c4;
            const f32x4 v = *(const f32x4*)(x + (i0 + p) * D + 4 * c);
            xs[p * XSS + 4 * c] = v[0]; xs[p * XSS + 4 * c + 1] = v[1]; xs[p * XSS + 4 * c + 2] = v[2]; xs[p * XSS + 4 * c + 3] = v[3];
        }
    } else {
        for (int e = tid; e < pv * D; e += 256) {
            const int p = e / D, k = e - p * D;
            xs[p * XSS + k] = x[(i0 + p) * D + k];
        }
    }
    __syncthreads();
    const int pq = tid & 15, kq = tid >> 4;                          // persons 4 pq .. 4 pq + 3, latent rows kq, kq + 16, ...
    const bool vec = (nb & 3) == 0 && 4 * pq + 4 <= pv;             // aligned full quad
    float mg = 0.f, md = 0.f, me = 0.f;                              // this thread's largest |gx|, |gd|, |eps|
    for (int k = kq; k < D; k += 16) {
        const float* src = gx_part + (int64_t)k * nbp + i0 + 4 * pq;
        f32x4 acc = *(const f32x4*)src;                              // nbp % 64 == 0: always in bounds and aligned
        for (int gq = 1; gq < groups; ++gq) acc += *(const f32x4*)(src + (int64_t)gq * LB_DP * nbp);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = fmaf(-scale, xs[(4 * pq + c) * XSS + k], acc[c]);
        const int64_t o = (int64_t)k * nb + i0 + 4 * pq;
        float* dst = gxT + o;
        if (vec) {
            *(f32x4*)dst = acc;
            if (gdT) {                                               // the DIAG-row operand of the guide backward (k_mvn_gd)
                const f32x4 e = *(const f32x4*)(epsT + o), l = *(const f32x4*)(ldT + o);
                f32x4 gdv;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    gdv[c] = fmaf(acc[c] * e[c], l[c], scale);
                    mg = fmaxf(mg, fabsf(acc[c])); md = fmaxf(md, fabsf(gdv[c])); me = fmaxf(me, fabsf(e[c]));
                }
                *(f32x4*)(gdT + o) = gdv;
            }
        } else {
            for (int c = 0; c < 4; ++c)
                if (4 * pq + c < pv) {
                    dst[c] = acc[c];
                    if (gdT) {
                        const float ev = epsT[o + c], gdv = fmaf(acc[c] * ev, ldT[o + c], scale);
                        gdT[o + c] = gdv;
                        mg = fmaxf(mg, fabsf(acc[c])); md = fmaxf(md, fabsf(gdv)); me = fmaxf(me, fabsf(ev));
                    }
                }
        }
    }
    if (opmax && gdT) {
        // the block's maxima through LDS, then one conditional atomic per word (atomic_max_raise, vx_common.h)
        __shared__ float mred[4][3];
        mg = wave_max_dpp(mg); md = wave_max_dpp(md); me = wave_max_dpp(me);
        if ((tid & 63) == 0) { mred[tid >> 6][0] = mg; mred[tid >> 6][1] = md; mred[tid >> 6][2] = me; }
        __syncthreads();
        if (tid < 3) atomic_max_raise(opmax + tid, fmaxf(fmaxf(mred[0][tid], mred[1][tid]), fmaxf(mred[2][tid], mred[3][tid])));
    }
    if (tid < pv) {
        float acc = 0.f, sq = 0.f;
        for (int gq = 0; gq < groups; ++gq) acc += ll_part[(int64_t)gq * nbp + i0 + tid];
        for (int k = 0; k < D; ++k) { const float t = xs[tid * XSS + k]; sq = fmaf(t, t, sq); }
        ll[i0 + tid] = fmaf(-0.5f, sq, acc);
    }
}
