// Amortized multivariate-normal guide (MvnEncoder, vi.py:438-455 + rsample vi.py:693) on fp32 MFMA.
//
// Row space of the encoder heads: r in [0, T)      -> fc22 row r, lower-triangle entry (k(r), l(r)) in
//                                                     torch.tril_indices order (vi.py:453)
//                                 r in [T, T + D)  -> fc21 row k = r - T (the location head)
// The (B, D, D) scale matrix of the reference (vi.py:452) is never materialised: every 32x32 MFMA tile
// of  M^T = W22 h^T  is consumed in registers,  x[p, k] += M[p, (k,l)] * eps[p, l]  (diag: exp(M)).
#pragma once
#include "vx_common.h"

#define ENC_P 64          // persons per workgroup
#define ENC_THREADS 256   // 4 waves, one per SIMD
#define ENC_ROWS 128      // head rows per LDS tile
#define ENC_JC 128        // item chunk of the fc1 contraction
#define ROW_NONE 0xFFFFFFFFu
#define ROW_LOC 0x80000000u

__device__ __forceinline__ uint32_t enc_row_code(int64_t r, int T, int D) {
    if (r < T) {
        int k = (int)((sqrtf(8.0f * (float)r + 1.0f) - 1.0f) * 0.5f);
        while ((int64_t)(k + 1) * (k + 2) / 2 <= r) ++k;
        while ((int64_t)k * (k + 1) / 2 > r) --k;
        const int l = (int)(r - (int64_t)k * (k + 1) / 2);
        return ((uint32_t)k << 16) | (uint32_t)l;
    }
    if (r < (int64_t)T + D) return ROW_LOC | (uint32_t)(r - T);
    return ROW_NONE;
}

struct EncDims {
    int D, J, H, Hp, DS, T;     // Hp = H rounded up to 32; DS = odd LDS stride >= D + 1
    int64_t nb;
};

__host__ __device__ inline int enc_ds(int D) { return (D + 1) | 1; }

// LDS carve (floats). Region U is shared between the fc1 phase and the head phase.
__host__ __device__ inline size_t enc_fwd_lds_floats(int D, int Hp) {
    const size_t DS = enc_ds(D);
    const size_t h = (size_t)ENC_P * (Hp + 1);
    const size_t ua = (size_t)ENC_P * (ENC_JC + 1) + (size_t)Hp * (ENC_JC + 1);
    const size_t ub = (size_t)ENC_ROWS * (Hp + 1) + 2 * ENC_ROWS;
    // eps [P][DS] floats | U.  x [P][DS] and ent [P] accumulate as 64-bit fixed point (fx_add: order-independent sums);
    // they exist in the head phase only and sit in U behind its head-phase part
    const size_t ubx = ub + 2 + 2 * ENC_P * DS + 2 * ENC_P;
    return h + ENC_P * DS + 2 + (ua > ubx ? ua : ubx);
}

template <int HT>   // HT = Hp / 32 hidden tiles
__global__ __launch_bounds__(ENC_THREADS) void k_mvn_enc_fwd(
    EncDims dm, const uint8_t* __restrict__ y, const int64_t* __restrict__ rows, int64_t gid0,
    const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ W21,
    const float* __restrict__ b21, const float* __restrict__ W22, const float* __restrict__ b22,
    const float* __restrict__ eps_in, uint64_t seed, uint32_t step, const uint32_t* __restrict__ step_dev, uint32_t stream,
    float* __restrict__ h_out, float* __restrict__ x_out, float* __restrict__ eps_out,
    float* __restrict__ ldT, float* __restrict__ ent_out) {
    if (step_dev) step = *step_dev;                              // captured step: the Philox step lives in device memory
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int D = dm.D, J = dm.J, H = dm.H, Hp = dm.Hp, DS = dm.DS, T = dm.T;
    const int HS = Hp + 1;
    float* h_lds = smem;                                   // [P][HS]
    float* eps_lds = h_lds + ENC_P * HS;                   // [P][DS]
    float* U = eps_lds + ENC_P * DS;
    U += ((size_t)(U - smem)) & 1;                         // 8-byte aligned
    long long* x_lds = (long long*)(U + (((size_t)ENC_ROWS * (Hp + 1) + 2 * ENC_ROWS + 1) & ~(size_t)1));   // [P][DS] fixed point
    long long* ent_lds = x_lds + ENC_P * DS;               // [P]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int64_t i0 = (int64_t)blockIdx.x * ENC_P;

    // ------------------------------------------------------------------ phase A: fc1 + softplus
    {
        float* Yf = U;                                     // [P][JC+1]  encoder input (NaN -> -1)
        float* W1c = U + ENC_P * (ENC_JC + 1);             // [Hp][JC+1]
        constexpr int TPW = (HT + 1) / 2;
        f32x16 acc[TPW];
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[t] = zero16();
        const int u = wave & 1;                            // person tile of this wave
        for (int jc = 0; jc < J; jc += ENC_JC) {
            for (int e = tid; e < ENC_P * ENC_JC; e += ENC_THREADS) {
                const int p = e / ENC_JC, jj = e - p * ENC_JC;
                const int64_t i = i0 + p;
                float v = 0.f;
                if (i < dm.nb && jc + jj < J) {
                    const int64_t row = rows ? rows[i] : i;
                    const unsigned yy = y[row * J + jc + jj];
                    v = (yy == 255u) ? -1.0f : (float)yy;   // vi.py:689-691
                }
                Yf[p * (ENC_JC + 1) + jj] = v;
            }
            for (int e = tid; e < Hp * ENC_JC; e += ENC_THREADS) {
                const int hh = e / ENC_JC, jj = e - hh * ENC_JC;
                W1c[hh * (ENC_JC + 1) + jj] = (hh < H && jc + jj < J) ? W1[(int64_t)hh * J + jc + jj] : 0.f;
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const int ht = (wave >> 1) + 2 * t;
                if (ht < HT) {
                    const float* ap = W1c + (32 * ht + l31) * (ENC_JC + 1) + half;
                    const float* bp = Yf + (32 * u + l31) * (ENC_JC + 1) + half;
#pragma unroll 8
                    for (int s = 0; s < ENC_JC / 2; ++s) acc[t] = mfma32(ap[2 * s], bp[2 * s], acc[t]);
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int ht = (wave >> 1) + 2 * t;
            if (ht < HT) {
                const int p = 32 * u + l31;
                const int64_t i = i0 + p;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int hh = 32 * ht + crow32(r, half);
                    float hv = 0.f;
                    if (hh < H) {
                        hv = softplusf_(acc[t][r] + b1[hh]);             // vi.py:449
                        if (i < dm.nb) h_out[i * H + hh] = hv;
                    }
                    h_lds[p * HS + hh] = hv;
                }
            }
        }
    }
    // ------------------------------------------------------------------ eps, x := 0
    __syncthreads();                                       // every wave is done with the fc1 staging area (x overlays it)
    {
        const int nblk = (D + 3) >> 2;
        for (int e = tid; e < ENC_P * nblk; e += ENC_THREADS) {
            const int p = e / nblk, blk = e - p * nblk;
            const int64_t i = i0 + p;
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            if (i < dm.nb) {
                if (eps_in) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (4 * blk + q < D) z[q] = eps_in[i * D + 4 * blk + q];
                } else {
                    const int64_t row = rows ? rows[i] : i;
                    z = philox_normal4(seed, step, stream, gid0 + row, (uint32_t)blk);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (4 * blk + q < D) {
                    eps_lds[p * DS + 4 * blk + q] = z[q];
                    if (i < dm.nb) eps_out[i * D + 4 * blk + q] = z[q];
                }
        }
        for (int e = tid; e < ENC_P * DS; e += ENC_THREADS) x_lds[e] = 0;
        if (tid < ENC_P) ent_lds[tid] = 0;
    }
    __syncthreads();
    // ------------------------------------------------------------------ phase B: head rows
    {
        float* Wt = U;                                      // [ROWS][HS]
        uint32_t* rowtab = (uint32_t*)(U + ENC_ROWS * HS);  // [ROWS]
        float* biasl = U + ENC_ROWS * HS + ENC_ROWS;        // [ROWS]
        const int64_t RT = (int64_t)T + D;
        const int n_tiles = (int)((RT + ENC_ROWS - 1) / ENC_ROWS);
        for (int tile = 0; tile < n_tiles; ++tile) {
            const int64_t r0 = (int64_t)tile * ENC_ROWS;
            for (int e = tid; e < ENC_ROWS * Hp; e += ENC_THREADS) {
                const int rl = e / Hp, hh = e - rl * Hp;
                const int64_t r = r0 + rl;
                float v = 0.f;
                if (hh < H) {
                    if (r < T) v = W22[r * H + hh];
                    else if (r < RT) v = W21[(r - T) * H + hh];
                }
                Wt[rl * HS + hh] = v;
            }
            if (tid < ENC_ROWS) {
                const int64_t r = r0 + tid;
                rowtab[tid] = enc_row_code(r, T, D);
                biasl[tid] = (r < T) ? b22[r] : (r < RT ? b21[r - T] : 0.f);
            }
            __syncthreads();
            f32x16 a0 = zero16(), a1 = zero16();
            {
                const float* ap = Wt + (32 * wave + l31) * HS + half;
                const float* bp0 = h_lds + l31 * HS + half;
                const float* bp1 = h_lds + (32 + l31) * HS + half;
#pragma unroll 8
                for (int s = 0; s < Hp / 2; ++s) {
                    const float a = ap[2 * s];
                    a0 = mfma32(a, bp0[2 * s], a0);
                    a1 = mfma32(a, bp1[2 * s], a1);
                }
            }
#pragma unroll
            for (int uu = 0; uu < 2; ++uu) {
                const int p = 32 * uu + l31;
                const int64_t i = i0 + p;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = 32 * wave + crow32(r, half);
                    const uint32_t code = rowtab[rl];
                    if (code == ROW_NONE) continue;
                    const float v = (uu == 0 ? a0[r] : a1[r]) + biasl[rl];
                    if (code & ROW_LOC) {
                        fx_add(&x_lds[p * DS + (int)(code & 0xFFFFu)], v);             // loc head (vi.py:450)
                    } else {
                        const int k = (int)(code >> 16), l = (int)(code & 0xFFFFu);
                        if (l < k) {
                            fx_add(&x_lds[p * DS + k], v * eps_lds[p * DS + l]);        // tril(M,-1) eps
                        } else {
                            const float ld = __expf(v);                                // exp(diag M): vi.py:686
                            fx_add(&x_lds[p * DS + k], ld * eps_lds[p * DS + k]);
                            fx_add(&ent_lds[p], v);
                            if (i < dm.nb) ldT[(int64_t)k * dm.nb + i] = ld;
                        }
                    }
                }
            }
            __syncthreads();
        }
    }
    // ------------------------------------------------------------------ phase C: write x, entropy part
    for (int e = tid; e < ENC_P * D; e += ENC_THREADS) {
        const int p = e / D, k = e - p * D;
        const int64_t i = i0 + p;
        if (i < dm.nb) x_out[i * D + k] = fx_get(x_lds[p * DS + k]);
    }
    if (tid < ENC_P) {
        const int64_t i = i0 + tid;
        if (i < dm.nb) {
            float s = 0.f;
            for (int k = 0; k < D; ++k) { const float e = eps_lds[tid * DS + k]; s += e * e; }
            ent_out[i] = 0.5f * s + fx_get(ent_lds[tid]);        // -log q + const = 0.5|eps|^2 + sum_k M_kk
        }
    }
}
