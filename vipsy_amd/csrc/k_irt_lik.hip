// Model likelihood + gradients for D >= 2 IRT (vi.py:32-66 response functions, vi.py:596-625 model +
// missing mask, Bernoulli log-lik) as three chained fp32-MFMA contractions per (64 persons x 128 items):
//     Z^T[j,p]   = sum_k  a_aug[k,j] x_aug[p,k]        (x_aug = [x, 1], a_aug = [a; b]  -> z = Dc * Z)
//     gx^T[k,p] += sum_j  a_aug[k,j] R[j,p]            R = scale * Dc * dlogp/dz  (0 on missing cells)
//     GA[k,j]   += sum_p  x_aug[p,k] R[j,p]            row k = D of GA is the b-gradient
// Item parameters are staged in LDS per 128-item chunk; the response bytes are read once, coalesced.
#pragma once
#include "vx_common.h"
#include <type_traits>

#define LIK_P 64
#define LIK_THREADS 256
#define LIK_JC 128

struct LikDims {
    int D, J, DS, Dk2, model;   // DS: odd LDS stride >= D + 2;  Dk2 = (D + 1) rounded up to even
    int fast;                   // D % 4 == 0, J % 4 == 0, 16-byte aligned x / a / b / y: batched 16-byte staging
    float Dc, scale;
    int64_t nb;
    int64_t slab_len;           // D*J + 3*J
};

__host__ __device__ inline int lik_ds(int D) { return (D + 2) | 1; }
__host__ __device__ inline size_t lik_lds_floats(int D, int nch, int gen) {
    const size_t DS = lik_ds(D), Dk2 = (D + 2) & ~1;
    // ll / gc / gd accumulate as 64-bit fixed point (fx_add): two floats of space each
    return (size_t)LIK_P * DS + Dk2 * (LIK_JC + 1) + (size_t)LIK_JC * (LIK_P + 1) + LIK_P * 33 + 2 * LIK_P + 2 +
           (gen ? (size_t)(3 * LIK_JC + 4 * nch * LIK_JC + 2) : 0);
}

template <int KT, int NCH, int GEN>
__global__ __launch_bounds__(LIK_THREADS) void k_irt_lik(
    LikDims dm, const uint8_t* __restrict__ y, const int64_t* __restrict__ rows, const float* __restrict__ x,
    const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c_un,
    const float* __restrict__ d_un, float* __restrict__ gx_part /*[groups][nb][D]*/,
    float* __restrict__ ll_part /*[groups][nb]*/, float* __restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int D = dm.D, J = dm.J, DS = dm.DS, Dk2 = dm.Dk2;
    constexpr int AS = LIK_JC + 1, RS = LIK_P + 1, YS = 132;
    float* x_lds = smem;                              // [P][DS]   x_aug
    float* a_lds = x_lds + LIK_P * DS;                // [Dk2][AS] a_aug chunk
    float* R_lds = a_lds + Dk2 * AS;                  // [JC][RS]
    uint8_t* Yb = (uint8_t*)(R_lds + LIK_JC * RS);    // [P][132] bytes
    float* after_y = R_lds + LIK_JC * RS + LIK_P * 33;
    long long* ll_lds = (long long*)(after_y + (((size_t)(after_y - smem)) & 1));   // [P] fixed point (8-byte aligned)
    float* cs = (float*)(ll_lds + LIK_P);             // GEN: c[JC], d[JC], 1-d[JC], gc[NCH*JC], gd[NCH*JC] (fixed point)
    float* dsv = cs + LIK_JC;
    float* omds = dsv + LIK_JC;
    long long* gc_acc = (long long*)(omds + LIK_JC + (((size_t)(omds + LIK_JC - smem)) & 1));
    long long* gd_acc = gc_acc + NCH * LIK_JC;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int g = blockIdx.x;
    const int jbase = g * NCH * LIK_JC;
    const int64_t n_ptiles = (dm.nb + LIK_P - 1) / LIK_P;
    constexpr int GXT = (2 * KT + 3) / 4;             // gx tiles per wave

    f32x16 ga[NCH][KT];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) ga[c][kt] = zero16();
    if (GEN) {
        for (int e = tid; e < 2 * NCH * LIK_JC; e += LIK_THREADS) gc_acc[e] = 0;
    }

    for (int64_t tile = blockIdx.y; tile < n_ptiles; tile += gridDim.y) {
        const int64_t i0 = tile * LIK_P;
        if (dm.fast) {
            // [P][D] contiguous floats -> [P][DS]; all loads of a batch are in flight before the first store
            const int c4 = D / 4, n4 = LIK_P * c4;
            const int pv = (int)((dm.nb - i0) < LIK_P ? (dm.nb - i0) : LIK_P);
            const float4* src = (const float4*)(x + i0 * D);
            for (int base = 0; base < n4; base += LIK_THREADS * 4) {
                float4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int idx = base + q * LIK_THREADS + tid;
                    v[q] = (idx < pv * c4) ? src[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int idx = base + q * LIK_THREADS + tid;
                    if (idx < n4) {
                        const int p = idx / c4, c = idx - p * c4;
                        float* dst = x_lds + p * DS + 4 * c;
                        dst[0] = v[q].x; dst[1] = v[q].y; dst[2] = v[q].z; dst[3] = v[q].w;
                    }
                }
            }
            for (int e = tid; e < LIK_P * (DS - D); e += LIK_THREADS) {
                const int p = e / (DS - D), k = D + (e - p * (DS - D));
                x_lds[p * DS + k] = (k == D && p < pv) ? 1.0f : 0.f;
            }
        } else {
            for (int e = tid; e < LIK_P * DS; e += LIK_THREADS) {
                const int p = e / DS, k = e - p * DS;
                const int64_t i = i0 + p;
                float v = 0.f;
                if (i < dm.nb) v = (k < D) ? x[i * D + k] : (k == D ? 1.0f : 0.f);
                x_lds[e] = v;
            }
        }
        if (tid < LIK_P) ll_lds[tid] = 0;
        f32x16 gxa[GXT];
#pragma unroll
        for (int t = 0; t < GXT; ++t) gxa[t] = zero16();

        // one 128-item chunk; `cc` is an integral_constant so that ga[c][..] stays in registers
        auto do_chunk = [&](auto cc) {
            constexpr int c = decltype(cc)::value;
            const int jc = jbase + c * LIK_JC;
            if (jc < J) {                                                   // block-uniform
                if (dm.fast) {
                    // a_aug chunk: (D+1) rows x 32 float4; response chunk: P rows x 32 words
                    const int n4 = Dk2 * (LIK_JC / 4);
                    for (int base = 0; base < n4; base += LIK_THREADS * 4) {
                        float4 v[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int idx = base + q * LIK_THREADS + tid;
                            const int k = idx >> 5, j = jc + 4 * (idx & 31);
                            v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                            if (idx < n4 && j < J) {
                                if (k < D) v[q] = *(const float4*)(a + (int64_t)k * J + j);
                                else if (k == D) v[q] = *(const float4*)(b + j);
                            }
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int idx = base + q * LIK_THREADS + tid;
                            if (idx < n4) {
                                float* dst = a_lds + (idx >> 5) * AS + 4 * (idx & 31);
                                dst[0] = v[q].x; dst[1] = v[q].y; dst[2] = v[q].z; dst[3] = v[q].w;
                            }
                        }
                    }
                    uint32_t w[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int idx = q * LIK_THREADS + tid;                  // P * 32 words = 8 per thread
                        const int p = idx >> 5, jw = jc + 4 * (idx & 31);
                        const int64_t i = i0 + p;
                        w[q] = 0xFEFEFEFEu;                                     // 254 = outside the problem
                        if (i < dm.nb && jw < J) {
                            const int64_t row = rows ? rows[i] : i;
                            w[q] = *(const uint32_t*)(y + row * J + jw);
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int idx = q * LIK_THREADS + tid;
                        ((uint32_t*)Yb)[(idx >> 5) * (YS / 4) + (idx & 31)] = w[q];
                    }
                } else {
                    for (int e = tid; e < Dk2 * LIK_JC; e += LIK_THREADS) {
                        const int k = e / LIK_JC, jj = e - k * LIK_JC;
                        const int j = jc + jj;
                        float v = 0.f;
                        if (j < J) v = (k < D) ? a[(int64_t)k * J + j] : (k == D ? b[j] : 0.f);
                        a_lds[k * AS + jj] = v;
                    }
                    for (int e = tid; e < LIK_P * LIK_JC; e += LIK_THREADS) {
                        const int p = e / LIK_JC, jj = e - p * LIK_JC;
                        const int64_t i = i0 + p;
                        uint8_t yy = 254;                                        // 254 = outside the problem
                        if (i < dm.nb && jc + jj < J) {
                            const int64_t row = rows ? rows[i] : i;
                            yy = y[row * J + jc + jj];
                        }
                        Yb[p * YS + jj] = yy;
                    }
                }
                if (GEN && tid < LIK_JC) {
                    const int j = jc + tid;
                    cs[tid] = (j < J) ? fminf(sigmoidf_(c_un[j]), 1.0f - VX_EPS32) : 0.f;
                    const bool has_d = (dm.model == 4 && j < J);
                    dsv[tid] = has_d ? fminf(sigmoidf_(d_un[j]), 1.0f - VX_EPS32) : 1.0f;
                    omds[tid] = has_d ? fmaxf(sigmoidf_(-d_un[j]), VX_EPS32) : 0.f;
                }
                __syncthreads();
                // ---- Z^T tile: rows = items of this wave, cols = persons
                f32x16 z0 = zero16(), z1 = zero16();
                {
                    const float* ap = a_lds + half * AS + 32 * wave + l31;
                    const float* bp0 = x_lds + l31 * DS + half;
                    const float* bp1 = x_lds + (32 + l31) * DS + half;
#pragma unroll 4
                    for (int s = 0; s < Dk2 / 2; ++s) {
                        const float av = ap[2 * s * AS];
                        z0 = mfma32(av, bp0[2 * s], z0);
                        z1 = mfma32(av, bp1[2 * s], z1);
                    }
                }
#pragma unroll
                for (int uu = 0; uu < 2; ++uu) {
                    const int p = 32 * uu + l31;
                    float llp = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int jj = 32 * wave + crow32(r, half);
                        const unsigned yy = Yb[p * YS + jj];
                        const float z = dm.Dc * (uu == 0 ? z0[r] : z1[r]);
                        float lp = 0.f, dz = 0.f, dc = 0.f, dd = 0.f;
                        if (yy != 254u) {
                            if (GEN) {
                                if (dm.model == 4) irt_cell<4>(z, yy, cs[jj], dsv[jj], omds[jj], lp, dz, dc, dd);
                                else irt_cell<3>(z, yy, cs[jj], 1.0f, 0.f, lp, dz, dc, dd);
                            } else {
                                irt_cell<2>(z, yy, 0.f, 1.f, 0.f, lp, dz, dc, dd);
                            }
                        }
                        llp += lp;
                        R_lds[jj * RS + p] = dm.scale * dm.Dc * dz;
                        if (GEN) {
                            float vc = dm.scale * dc, vd = dm.scale * dd;
#pragma unroll
                            for (int o = 16; o > 0; o >>= 1) { vc += __shfl_xor(vc, o, 64); vd += __shfl_xor(vd, o, 64); }
                            if (l31 == 0) {
                                fx_add(&gc_acc[c * LIK_JC + jj], vc);
                                fx_add(&gd_acc[c * LIK_JC + jj], vd);
                            }
                        }
                    }
                    fx_add(&ll_lds[p], llp);
                }
                __syncthreads();
                // ---- gx^T tiles (rows = dims, cols = persons), contraction over the chunk's items
#pragma unroll
                for (int t = 0; t < GXT; ++t) {
                    const int id = wave + 4 * t;
                    if (id < 2 * KT) {
                        const int kt = id >> 1, uu = id & 1;
                        int krow = 32 * kt + l31;
                        krow = krow < Dk2 ? krow : Dk2 - 1;                  // rows >= D are discarded later
                        const float* ap = a_lds + krow * AS + half;
                        const float* bp = R_lds + half * RS + 32 * uu + l31;
#pragma unroll 8
                        for (int s = 0; s < LIK_JC / 2; ++s) gxa[t] = mfma32(ap[2 * s], bp[2 * s * RS], gxa[t]);
                    }
                }
                // ---- GA tiles (rows = dims incl. the b row, cols = items of this wave), over persons
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    int krow = 32 * kt + l31;
                    krow = krow < Dk2 ? krow : Dk2 - 1;
                    const float* ap = x_lds + half * DS + krow;
                    const float* bp = R_lds + (32 * wave + l31) * RS + half;
#pragma unroll 8
                    for (int s = 0; s < LIK_P / 2; ++s) ga[c][kt] = mfma32(ap[2 * s * DS], bp[2 * s], ga[c][kt]);
                }
                __syncthreads();
            }
        };
        do_chunk(std::integral_constant<int, 0>{});
        if constexpr (NCH > 1) do_chunk(std::integral_constant<int, 1>{});
        if constexpr (NCH > 2) {
            do_chunk(std::integral_constant<int, 2>{});
            do_chunk(std::integral_constant<int, 3>{});
        }
        // ---- per-person log-lik + prior; gx = sum_j R a - scale * x (only group 0 adds the prior term)
        if (tid < LIK_P) {
            const int64_t i = i0 + tid;
            if (i < dm.nb) {
                float s2 = 0.f;
                if (g == 0)
                    for (int k = 0; k < D; ++k) { const float xv = x_lds[tid * DS + k]; s2 += xv * xv; }
                ll_part[(int64_t)g * dm.nb + i] = fx_get(ll_lds[tid]) - 0.5f * s2;
            }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < GXT; ++t) {
            const int id = wave + 4 * t;
            if (id < 2 * KT) {
                const int kt = id >> 1, uu = id & 1;
                const int p = 32 * uu + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int k = 32 * kt + crow32(r, half);
                    if (k < D) {
                        const float xv = x_lds[p * DS + k];
                        x_lds[p * DS + k] = gxa[t][r] - (g == 0 ? dm.scale * xv : 0.f);
                    }
                }
            }
        }
        __syncthreads();
        if (dm.fast) {
            const int c4 = D / 4;
            float4* dst = (float4*)(gx_part + ((int64_t)g * dm.nb + i0) * D);
            const int pv = (int)((dm.nb - i0) < LIK_P ? (dm.nb - i0) : LIK_P);
            for (int idx = tid; idx < pv * c4; idx += LIK_THREADS) {
                const int p = idx / c4, c = idx - p * c4;
                const float* sp = x_lds + p * DS + 4 * c;
                dst[idx] = make_float4(sp[0], sp[1], sp[2], sp[3]);
            }
        } else {
            for (int e = tid; e < LIK_P * D; e += LIK_THREADS) {
                const int p = e / D, k = e - p * D;
                const int64_t i = i0 + p;
                if (i < dm.nb) gx_part[((int64_t)g * dm.nb + i) * D + k] = x_lds[p * DS + k];
            }
        }
        __syncthreads();
    }
    // ---- item-gradient slab of this person range: d ELBO / d a, b (and c_un, d_un)
    float* slab = slabs + (int64_t)blockIdx.y * dm.slab_len;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int j = jbase + c * LIK_JC + 32 * wave + l31;
        if (j < J) {
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int k = 32 * kt + crow32(r, half);
                    if (k <= D) slab[(int64_t)k * J + j] = ga[c][kt][r];     // k == D lands in the b segment
                }
        }
    }
    if (GEN) {
        __syncthreads();
        for (int e = tid; e < NCH * LIK_JC; e += LIK_THREADS) {
            const int j = jbase + e;
            if (j < J) {
                slab[(int64_t)(D + 1) * J + j] = fx_get(gc_acc[e]);
                slab[(int64_t)(D + 2) * J + j] = fx_get(gd_acc[e]);
            }
        }
    } else if (tid < 0) {
        (void)cs; (void)dsv; (void)omds;
    }
}
