// Model likelihood + gradients for 64 <= D <= 127 (vi.py:32-66 response functions, vi.py:596-625 model +
// missing mask, Bernoulli log-lik): the same three fp32-MFMA contractions as k_irt_lik.hip, reorganised so that
// every operand that does not change from one person tile to the next lives in registers:
//
//   a workgroup owns ONE 128-item chunk for its whole life and walks 64-person tiles;
//   wave w owns items 32w..32w+31 of the chunk for Z / the epilogue / GA, and latent rows 32w..32w+31 for gx.
//
//     Z[p,j]    = sum_k x_aug[p,k] a_aug[k,j]      A <- x_lds (one 16-byte LDS read per 4 MFMAs), B <- aZ registers
//     R[p,j]    = scale * Dc * dlogp/dz            epilogue with lane = item: c, d, gc, gd are per-lane registers
//     GA[k,j]  += sum_p x_aug[p,k] R[p,j]          A <- x_lds (16-byte read, rows k = 4*m + kt), B <- R registers
//     gx^T[k,p] = sum_j a[k,j] R[p,j]              A <- aG registers, B <- R_lds (16-byte read per 4 MFMAs)
//
// The MFMA contraction index is visited in the order the 16-byte reads deliver it (k = 8q + 4*half + i), which
// both operands agree on; a sum is a sum.  Per 64-person tile a wave issues 360 MFMAs and ~120 LDS reads.
#pragma once
#include "vx_common.h"
#include <type_traits>

#define LR_P 64
#define LR_JC 128
#define LR_THREADS 256
#define LR_RS 132            // floats per R / LP row: 128 + 4 (== 4 mod 32: conflict-free 16-byte row reads)
#define LR_YS 128            // response bytes per person row (dense: the DMA writes 8 rows per instruction)

// DMA sources of the x_aug pad columns: chunk 0 = [1, 0, 0, 0] (the bias column of a present person), chunk 1 = zeros
__device__ __attribute__((aligned(16))) const float lr_pad_chunks[8] = {1.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

struct LikRDims {
    int D, J, K8, XS;        // K8 = (D + 1) rounded up to 8;  XS = K8 + 4 (== 4 mod 8)
    int model, fast, groups, n_pr;
    int gxt;                 // 1: gx partials are written dimension-major [groups][D][nb] (for the guide backward)
    float Dc, scale;
    int64_t nb, slab_len;
};

__host__ __device__ inline size_t likr_lds_bytes(int XS) {
    return sizeof(float) * ((size_t)2 * LR_P * XS + 2 * LR_P * LR_RS) + 2 * LR_P * LR_YS;
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

// ABL: ablation bits for tools/lik_bench.hip only (0 in the library)
template <int GEN, int NQ, int FAST, int ABL = 0>
__global__ __launch_bounds__(LR_THREADS) void k_irt_lik_r(
    LikRDims dm, const uint8_t* __restrict__ y, const int64_t* __restrict__ rows, const float* __restrict__ x,
    const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c_un,
    const float* __restrict__ d_un, float* __restrict__ gx_part /*[groups][nb][D]*/,
    float* __restrict__ ll_part /*[groups][nb]*/, float* __restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int D = dm.D, J = dm.J, XS = dm.XS;
    float* x_lds = smem;                                  // [2][P][XS]  x_aug = [x, 1, 0..]
    float* R_lds = x_lds + 2 * LR_P * XS;                 // [P][RS]
    float* LP_lds = R_lds + LR_P * LR_RS;                 // [P][RS]
    uint8_t* Yb = (uint8_t*)(LP_lds + LR_P * LR_RS);      // [2][P][YS]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), half = lane >> 5, l31 = lane & 31;
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
    const int j0 = g * LR_JC;
    const int jw = j0 + 32 * wave + l31;                  // this lane's item (Z / epilogue / GA column)
    const bool jv = jw < J;
    const int64_t n_ptiles = (dm.nb + LR_P - 1) / LR_P;

    // ---- staging of one person tile: global -> registers (stage_load), registers -> LDS (stage_store)
    // Staging of one person tile.  FAST (D % 4 == 0, J % 4 == 0, aligned bases): global -> LDS DMA, no registers,
    // issued from phase D of the previous tile and waited for (vmcnt) just before the barrier that opens the tile.
    //   x: one DMA per person row (lanes < D/4 move 16 bytes each) -> x_lds row p, stride XS
    //   y: interior chunk: one DMA per 8 person rows (8 lanes x 16 bytes per row); the last, ragged chunk of the
    //      item axis: one DMA per 2 rows (32 lanes x 4 bytes), lanes past J switched off
    // The pad columns [D, XS) of x_aug ([1, 0, 0, ...]) and the response bytes past J (254 = outside the problem)
    // are never touched by the DMA: written once here, for both buffers.
    const bool jfull = j0 + LR_JC <= J;                                // block-uniform: no item edge in this chunk
    if (!FAST) {
        for (int e = tid; e < 2 * LR_P * (XS - D); e += LR_THREADS) {
            const int row = e / (XS - D), k = D + (e - row * (XS - D));
            x_lds[row * XS + k] = (k == D) ? 1.0f : 0.f;
        }
    }
    for (int e = tid; e < 2 * LR_P * (LR_YS / 4); e += LR_THREADS) ((uint32_t*)Yb)[e] = 0xFEFEFEFEu;
    // FAST: the x_aug tile [64][XS] is one contiguous run of 64 * XS / 4 chunks = XS / 4 full-wave transfers (a partial
    // EXEC mask makes a DMA several times slower); each lane's chunk is data, the [1,0,0,0] pad or zeros -- fixed per
    // lane and transfer, only the person offset of the tile changes.  Wave w issues transfers w, w + 4, ...
    constexpr int LR_MAXX = 9;                                         // XS <= 132: 33 transfers / 4 waves
    const float* xbase[LR_MAXX];                                       // data: x + row * D + 4 cc; pad: the pad chunk
    int xrow[LR_MAXX];                                                 // person row of the chunk; bit 8: chunk is data
    if (FAST) {
        const int cr = XS >> 2, c4 = D >> 2;
#pragma unroll
        for (int u = 0; u < LR_MAXX; ++u) {
            const int ch = 64 * (wave + 4 * u) + lane;
            int row = ch / cr;
            const int cc = ch - row * cr;
            const bool isdata = cc < c4;
            if (row > LR_P - 1) row = LR_P - 1;
            xbase[u] = isdata ? x + (int64_t)row * D + 4 * cc : lr_pad_chunks + (cc == c4 ? 0 : 4);
            xrow[u] = row | (isdata ? 256 : 0);
        }
    }
    __syncthreads();
    auto stage = [&](int64_t tile, int buf) {
        float* xb = x_lds + buf * LR_P * XS;
        uint8_t* yb = Yb + buf * LR_P * LR_YS;
        const int64_t i0 = tile * LR_P;
        const int pv = (int)((dm.nb - i0) < LR_P ? (dm.nb - i0) : LR_P);
        if (FAST) {
            const int cr = XS >> 2;
            const uint32_t xl = lds_addr_uniform(xb) + (uint32_t)wave * 1024u;
            const int64_t off = i0 * D;
#pragma unroll
            for (int u = 0; u < LR_MAXX; ++u) {
                if (wave + 4 * u < cr && !(ABL & 128)) {               // wave-uniform
                    const float* src = xbase[u] + ((xrow[u] & 256) ? off : 0);
                    if ((xrow[u] & 255) >= pv) src = lr_pad_chunks + 4;    // absent person (last tile): an all-zero row
                    dma16(src, xl + (uint32_t)u * 4096u);
                }
            }
            if (ABL & 256) {
            } else if (jfull) {
                for (int r8 = wave; 8 * r8 < pv; r8 += 4) {
                    const int prow = 8 * r8 + (lane >> 3);
                    if (prow < pv) {
                        const int64_t row = (FAST == 2) ? rows[i0 + prow] : i0 + prow;
                        dma16(y + row * J + j0 + 16 * (lane & 7), lds_addr_uniform(yb + r8 * 8 * LR_YS));
                    }
                }
            } else {
                for (int r2 = wave; 2 * r2 < pv; r2 += 4) {
                    const int prow = 2 * r2 + (lane >> 5), jj = j0 + 4 * (lane & 31);
                    if (prow < pv && jj < J) {
                        const int64_t row = (FAST == 2) ? rows[i0 + prow] : i0 + prow;
                        dma4(y + row * J + jj, lds_addr_uniform(yb + r2 * 2 * LR_YS));
                    }
                }
            }
        } else {
            for (int e = tid; e < pv * D; e += LR_THREADS) {
                const int p = e / D, k = e - p * D;
                xb[p * XS + k] = x[(i0 + p) * D + k];
            }
            for (int e = tid; e < pv * LR_JC; e += LR_THREADS) {
                const int p = e >> 7, jj = e & 127;
                if (j0 + jj < J) {
                    const int64_t row = rows ? rows[i0 + p] : i0 + p;
                    yb[p * LR_YS + jj] = y[row * J + j0 + jj];
                }
            }
        }
        if (pv < LR_P) {                                               // the last tile: absent persons are all-zero rows
            if (!FAST)
                for (int e = tid; e < (LR_P - pv) * XS; e += LR_THREADS) xb[pv * XS + e] = 0.f;
            for (int e = tid; e < (LR_P - pv) * (LR_YS / 4); e += LR_THREADS)
                ((uint32_t*)yb)[pv * (LR_YS / 4) + e] = 0xFEFEFEFEu;
        }
    };

    // gx of one person half: gx = sum_j R a  (- scale * x once, in chunk 0); 16-byte stores
    auto store_gx = [&](int ph, const f32x16& gxa, int64_t i0, const float* xb) {
        const int p = 32 * ph + l31;
        const int64_t i = i0 + p;
        if (dm.gxt) {                                                  // lanes = consecutive persons: 128-byte rows
            if (i < dm.nb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int k = 32 * wave + crow32(r, half);
                    if (k < D)
                        gx_part[((int64_t)g * D + k) * dm.nb + i] = gxa[r] - (g == 0 ? dm.scale * xb[p * XS + k] : 0.f);
                }
            }
        } else if (i < dm.nb) {
            float* dst = gx_part + ((int64_t)g * dm.nb + i) * D;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int k = 32 * wave + 8 * qq + 4 * half;
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = gxa[4 * qq + e];
                if (FAST) {
                    if (k < D) {
                        if (g == 0) {
                            const f32x4 xo = *(const f32x4*)(xb + p * XS + k);
                            o -= dm.scale * xo;
                        }
                        *(f32x4*)(dst + k) = o;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (k + e < D) dst[k + e] = o[e] - (g == 0 ? dm.scale * xb[p * XS + k + e] : 0.f);
                }
            }
        }
    };
    int64_t tile = pr;
    int buf = 0;
    f32x16 gx1 = zero16();                                 // gx of persons 32..63: stored one tile late (see S1)
    int64_t i0_prev = -1;
    if (tile < n_ptiles) stage(tile, 0);
    // (the item operands are fetched BEHIND the first tile's transfers: in front of them the barrier that orders the pad fills
    // against the transfers also waited for these 116 loads, and a workgroup with one tile -- the reference's B = 100 step --
    // had the two latencies one after the other)
    // ---- register-resident item operands
    float aZ[NQ][4], aG[16][4];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = 8 * q + 4 * half + i;
            float v = 0.f;
            if (jv) {
                if (k < D) v = a[(int64_t)k * J + jw];
                else if (k == D) v = b[jw];
            }
            aZ[q][i] = dm.Dc * v;                         // z = Dc * (x.a + b)
        }
#pragma unroll
    for (int q = 0; q < 16; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kg = 32 * wave + l31, j = j0 + 8 * q + 4 * half + i;
            aG[q][i] = (kg < D && j < J) ? a[(int64_t)kg * J + j] : 0.f;
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

    for (; tile < n_ptiles; tile += dm.n_pr, buf ^= 1) {
        // this wave's DMA of the tile has landed.  The wait also covers older stores: the only recent ones would be
        // the gx of persons 32..63, which is why that store is deferred to just after this barrier.
        if (FAST && !(ABL & 64)) vx_wait_vmem();
        if constexpr (!(ABL & 16)) __syncthreads();        // S1: tile staged; R / LP of the previous tile consumed
        const float* xb = x_lds + buf * LR_P * XS;
        if constexpr (!(ABL & 4)) {
            if (i0_prev >= 0) store_gx(1, gx1, i0_prev, x_lds + (buf ^ 1) * LR_P * XS);
        }
        gx1 = zero16();
        const uint8_t* yb = Yb + buf * LR_P * LR_YS;
        const int64_t i0 = tile * LR_P;
        const int64_t next = tile + dm.n_pr;
        const bool has_next = (ABL & 8) ? false : next < n_ptiles;   // block-uniform

        // ---- one epilogue cell: lane = item, register r = person; R stays in the z register
        const float sdc = dm.scale * dm.Dc;
        auto cell = [&](auto phc, auto rc, f32x16& z) {
            constexpr int ph = decltype(phc)::value, r = decltype(rc)::value;
            const int p = 32 * ph + crow32(r, half);
            const unsigned yy = yb[p * LR_YS + 32 * wave + l31];
            const float zz = z[r];
            float lp, dz, dc, dd;
            if constexpr (ABL & 1) {
                lp = zz; dz = zz + (float)yy; dc = 0.f; dd = 0.f;
            } else if (GEN) {
                if (dm.model == 4) irt_cell<4>(zz, yy, cj, dj, omdj, lp, dz, dc, dd);
                else irt_cell<3>(zz, yy, cj, 1.0f, 0.f, lp, dz, dc, dd);
                gc = fmaf(dm.scale, dc, gc);
                gd = fmaf(dm.scale, dd, gd);
            } else {
                irt_cell<2>(zz, yy, 0.f, 1.f, 0.f, lp, dz, dc, dd);
            }
            const float rv = sdc * dz;
            z[r] = rv;
            if constexpr (!(ABL & 2)) {
                R_lds[p * LR_RS + 32 * wave + l31] = rv;
                LP_lds[p * LR_RS + 32 * wave + l31] = lp;
            } else {
                gc += lp;
            }
        };
        // Operand addresses of the three contractions (all 16-byte LDS reads):
        //   xZ: x rows for Z (A operand, k = 8q + 4*half + i);  xG: x rows for GA (A operand, rows k = 4*m + kt);
        //   rG: R rows for gx (B operand, j = 8q + 4*half + i)
        auto xZ = [&](int ph, int q) { return xb + (32 * ph + l31) * XS + 8 * q + 4 * half; };
        auto xG = [&](int ph, int s2) { return xb + (32 * ph + crow32(s2, half)) * XS + 4 * l31; };
        auto rG = [&](int ph, int q) { return R_lds + (32 * ph + l31) * LR_RS + 8 * q + 4 * half; };
        // Every step is one pinned scheduling region: the operand read of the NEXT step is issued first, then four
        // MFMAs with a slice of the VALU work (epilogue cells, ll, staging) in their shadow.
        f32x4 cur = *(const f32x4*)xZ(0, 0);
        auto step_Z = [&](auto qc, f32x16& z, const float* nextp) {
            constexpr int q = decltype(qc)::value;
            const f32x4 nxt = *(const f32x4*)nextp;
#pragma unroll
            for (int i = 0; i < 4; ++i) z = mfma32(cur[i], aZ[q][i], z);
            cur = nxt;
        };
        auto step_GA = [&](float rv, const float* nextp) {
            const f32x4 nxt = *(const f32x4*)nextp;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) ga[kt] = mfma32(cur[kt], rv, ga[kt]);
            cur = nxt;
        };
        auto step_gx = [&](auto qc, f32x16& gx, const float* nextp) {
            constexpr int q = decltype(qc)::value;
            const f32x4 nxt = *(const f32x4*)nextp;
#pragma unroll
            for (int i = 0; i < 4; ++i) gx = mfma32(aG[q][i], cur[i], gx);
            cur = nxt;
        };
        constexpr int CB = 10;                             // cells of persons 0..31 done under Z1; the rest under GA0
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;

        // ---- A: Z of persons 0..31 (rows = persons, cols = this wave's items)
        f32x16 z0 = zero16(), z1 = zero16();
        static_for<NQ>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            step_Z(qc, z0, q + 1 < NQ ? xZ(0, q + 1) : xZ(1, 0));
            __builtin_amdgcn_sched_barrier(0);
        });
        // ---- B: Z of persons 32..63 | epilogue cells 0..CB-1 of persons 0..31
        static_for<NQ>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            step_Z(qc, z1, q + 1 < NQ ? xZ(1, q + 1) : xG(0, 0));
            if constexpr (q < CB) cell(I0{}, qc, z0);
            __builtin_amdgcn_sched_barrier(0);
        });
        // ---- B2: GA steps of the finished cells | the remaining cells of persons 0..31
        static_for<CB>([&](auto sc) {
            constexpr int s2 = decltype(sc)::value;
            step_GA(z0[s2], xG(0, s2 + 1));
            if constexpr (CB + s2 < 16) cell(I0{}, std::integral_constant<int, CB + s2>{}, z0);
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (!(ABL & 16)) __syncthreads();        // S2: R rows of persons 0..31 complete
        // ---- C: rest of GA0, gx of persons 0..31 | epilogue of persons 32..63
        f32x16 gx0 = zero16();
        static_for<16 - CB>([&](auto sc) {
            constexpr int s2 = CB + decltype(sc)::value;
            step_GA(z0[s2], s2 + 1 < 16 ? xG(0, s2 + 1) : rG(0, 0));
            cell(I1{}, sc, z1);
            __builtin_amdgcn_sched_barrier(0);
        });
        static_for<16>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            step_gx(qc, gx0, rG(0, (q + 1) & 15));
            if constexpr (16 - CB + q < 16) cell(I1{}, std::integral_constant<int, 16 - CB + q>{}, z1);
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (!(ABL & 16)) __syncthreads();        // S3: R rows of persons 32..63 and all LP rows complete
        cur = *(const f32x4*)rG(1, 0);
        if (has_next) stage(next, buf ^ 1);                // DMA of the next tile flies under the 128 MFMAs of D
        // ---- D: gx of persons 32..63, GA of persons 32..63 | ll reduce, gx store, staging of the next tile
        static_for<16>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            step_gx(qc, gx1, q + 1 < 16 ? rG(1, q + 1) : xG(1, 0));
            if constexpr (q == 1 && !(ABL & 32)) {
                // per-person log-lik of this chunk (+ the N(0, I) prior once, in chunk 0)
                const int p = tid >> 2, q4 = tid & 3;
                float sll = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int ee = (e + 2 * q4) & 7;       // rotate: the 4 lanes of a person hit different banks
                    const f32x4 v = *(const f32x4*)(LP_lds + p * LR_RS + 32 * q4 + 4 * ee);
                    sll += (v[0] + v[1]) + (v[2] + v[3]);
                }
                {   // prior: the chunk blocks of a tile share |x|^2 by k range (balanced; their ll parts are summed)
                    const int kb = (D * g) / dm.groups, ke = (D * (g + 1)) / dm.groups;
                    float sq = 0.f;
                    for (int k = kb + q4; k < ke; k += 4) { const float xv1 = xb[p * XS + k]; sq = fmaf(xv1, xv1, sq); }
                    sll -= 0.5f * sq;
                }
                sll += dpp_mov0<0xB1, 0xF>(sll);
                sll += dpp_mov0<0x4E, 0xF>(sll);
                if (q4 == 0 && i0 + p < dm.nb) ll_part[(int64_t)g * dm.nb + i0 + p] = sll;
            }
            if constexpr (q == 4 && !(ABL & 4)) store_gx(0, gx0, i0, xb);
            __builtin_amdgcn_sched_barrier(0);
        });
        i0_prev = i0;
        static_for<16>([&](auto sc) {
            constexpr int s2 = decltype(sc)::value;
            step_GA(z1[s2], xG(1, (s2 + 1) & 15));
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (ABL & 4) { if (gx0[0] + gx1[3] == 123.456f) ll_part[0] = 1.f; }
        if constexpr (ABL & 8) buf ^= 1;
    }
    if constexpr (!(ABL & 4)) {
        if (i0_prev >= 0) store_gx(1, gx1, i0_prev, x_lds + (buf ^ 1) * LR_P * XS);
    }
    // ---- item-gradient slab of this person range
    float* slab = slabs + (int64_t)pr * dm.slab_len;
    if (jv) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = 4 * crow32(r, half) + kt;
                if (k <= D) slab[(int64_t)k * J + jw] = ga[kt][r];      // k == D lands in the b segment
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
