// Backward of the amortized MVN guide (autograd of vi.py:448-455 + LowerCholeskyTransform vi.py:686):
//   V[r,p]   = d ELBO / d head-row r for person p
//              tril row (k,l), l<k : gx[p,k] * eps[p,l]
//              diagonal   (k,k)    : gx[p,k] * eps[p,k] * exp(M_kk) + scale        
//              loc row     k       : gx[p,k]
//   k_mvn_enc_bwd_h : gh^T[hh,p] = sum_r W[r,hh] V[r,p]  -> ghpre = gh * sigmoid(pre)   (person-parallel)
//   k_mvn_enc_bwd_w : GW[r,hh]  += sum_p V[r,p] h[p,hh],  Gbias[r] += sum_p V[r,p]      (person-reduce)
//   k_fc1_bwd       : GW1[hh,j] += sum_p ghpre[p,hh] yin[p,j], Gb1[hh] += sum_p ghpre   (person-reduce)
// V is never stored: it is formed in registers as an MFMA operand.
#pragma once
#include "k_mvn_enc.hip"

__host__ __device__ inline size_t enc_bwdh_lds_floats(int D, int Hp) {
    const size_t DS = enc_ds(D);
    return (size_t)ENC_P * (Hp + 1) + 3 * ENC_P * DS + (size_t)ENC_ROWS * (Hp + 1) + ENC_ROWS;
}

__device__ __forceinline__ float enc_v(uint32_t code, const float* gx_p, const float* eps_p, const float* ld_p,
                                       float scale) {
    if (code == ROW_NONE) return 0.f;
    if (code & ROW_LOC) return gx_p[code & 0xFFFFu];
    const int k = (int)(code >> 16), l = (int)(code & 0xFFFFu);
    float v = gx_p[k] * eps_p[l];
    if (l == k) v = v * ld_p[k] + scale;
    return v;
}

template <int HT>
__global__ __launch_bounds__(ENC_THREADS) void k_mvn_enc_bwd_h(
    EncDims dm, float scale, const float* __restrict__ W21, const float* __restrict__ W22,
    const float* __restrict__ h_in, const float* __restrict__ eps_in, const float* __restrict__ ldT,
    const float* __restrict__ gx_in, float* __restrict__ ghpre_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int D = dm.D, H = dm.H, Hp = dm.Hp, DS = dm.DS, T = dm.T;
    const int HS = Hp + 1;
    float* h_lds = smem;                          // [P][HS]
    float* gx_lds = h_lds + ENC_P * HS;           // [P][DS]
    float* eps_lds = gx_lds + ENC_P * DS;
    float* ld_lds = eps_lds + ENC_P * DS;
    float* Wt = ld_lds + ENC_P * DS;              // [ROWS][HS]
    uint32_t* rowtab = (uint32_t*)(Wt + ENC_ROWS * HS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int64_t i0 = (int64_t)blockIdx.x * ENC_P;

    for (int e = tid; e < ENC_P * DS; e += ENC_THREADS) {
        const int p = e / DS, k = e - p * DS;
        const int64_t i = i0 + p;
        const bool ok = (i < dm.nb) && (k < D);
        gx_lds[e] = ok ? gx_in[i * D + k] : 0.f;
        eps_lds[e] = ok ? eps_in[i * D + k] : 0.f;
    }
    for (int e = tid; e < ENC_P * D; e += ENC_THREADS) {      // ldT is dimension-major: coalesced over persons
        const int k = e / ENC_P, p = e - k * ENC_P;
        const int64_t i = i0 + p;
        ld_lds[p * DS + k] = (i < dm.nb) ? ldT[(int64_t)k * dm.nb + i] : 0.f;
    }
    for (int e = tid; e < ENC_P * Hp; e += ENC_THREADS) {
        const int p = e / Hp, hh = e - p * Hp;
        const int64_t i = i0 + p;
        h_lds[p * HS + hh] = (i < dm.nb && hh < H) ? h_in[i * H + hh] : 0.f;
    }
    constexpr int TPW = (HT + 1) / 2;
    f32x16 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[t] = zero16();
    const int u = wave & 1;
    const int p = 32 * u + l31;
    const int64_t RT = (int64_t)T + D;
    const int n_tiles = (int)((RT + ENC_ROWS - 1) / ENC_ROWS);
    for (int tile = 0; tile < n_tiles; ++tile) {
        const int64_t r0 = (int64_t)tile * ENC_ROWS;
        __syncthreads();
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
        if (tid < ENC_ROWS) rowtab[tid] = enc_row_code(r0 + tid, T, D);
        __syncthreads();
        const float* gx_p = gx_lds + p * DS;
        const float* eps_p = eps_lds + p * DS;
        const float* ld_p = ld_lds + p * DS;
#pragma unroll 4
        for (int s = 0; s < ENC_ROWS / 2; ++s) {
            const int rl = 2 * s + half;
            const float v = enc_v(rowtab[rl], gx_p, eps_p, ld_p, scale);
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const int ht = (wave >> 1) + 2 * t;
                if (ht < HT) acc[t] = mfma32(Wt[rl * HS + 32 * ht + l31], v, acc[t]);
            }
        }
    }
    __syncthreads();
    // gh -> ghpre = gh * sigmoid(pre), sigmoid(pre) = 1 - exp(-softplus(pre)) = 1 - exp(-h)
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int ht = (wave >> 1) + 2 * t;
        if (ht < HT) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int hh = 32 * ht + crow32(r, half);
                if (hh < H) {
                    const float hv = h_lds[p * HS + hh];
                    h_lds[p * HS + hh] = acc[t][r] * (1.0f - __expf(-hv));
                }
            }
        }
    }
    __syncthreads();
    for (int e = tid; e < ENC_P * H; e += ENC_THREADS) {
        const int pp = e / H, hh = e - pp * H;
        const int64_t i = i0 + pp;
        if (i < dm.nb) ghpre_out[i * H + hh] = h_lds[pp * HS + hh];
    }
}

// ---------------------------------------------------------------------------------------------
#define BW_RT 4                         // row tiles (of 32) per wave  -> 512 rows per workgroup
#define BW_ROWS (4 * BW_RT * 32)

__host__ __device__ inline size_t enc_bwdw_lds_floats(int D, int Hp) {
    const size_t DS = enc_ds(D);
    return (size_t)ENC_P * (Hp + 1) + 3 * ENC_P * DS;
}

// slab layout (one per person range): [W21: D*H | b21: D | W22: T*H | b22: T]
template <int HT>
__global__ __launch_bounds__(ENC_THREADS) void k_mvn_enc_bwd_w(
    EncDims dm, float scale, const float* __restrict__ h_in, const float* __restrict__ eps_in,
    const float* __restrict__ ldT, const float* __restrict__ gx_in, float* __restrict__ slabs,
    int64_t slab_len) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int D = dm.D, H = dm.H, Hp = dm.Hp, DS = dm.DS, T = dm.T;
    const int HS = Hp + 1;
    float* h_lds = smem;
    float* gx_lds = h_lds + ENC_P * HS;
    float* eps_lds = gx_lds + ENC_P * DS;
    float* ld_lds = eps_lds + ENC_P * DS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int64_t RT = (int64_t)T + D;
    const int64_t rbase = (int64_t)blockIdx.x * BW_ROWS + (int64_t)wave * BW_RT * 32;
    uint32_t code[BW_RT];
#pragma unroll
    for (int t = 0; t < BW_RT; ++t) code[t] = enc_row_code(rbase + 32 * t + l31, T, D);
    f32x16 acc[BW_RT][HT];
    float bsum[BW_RT];
#pragma unroll
    for (int t = 0; t < BW_RT; ++t) {
        bsum[t] = 0.f;
#pragma unroll
        for (int ht = 0; ht < HT; ++ht) acc[t][ht] = zero16();
    }
    const int64_t n_ptiles = (dm.nb + ENC_P - 1) / ENC_P;
    for (int64_t tile = blockIdx.y; tile < n_ptiles; tile += gridDim.y) {
        const int64_t i0 = tile * ENC_P;
        __syncthreads();
        for (int e = tid; e < ENC_P * DS; e += ENC_THREADS) {
            const int p = e / DS, k = e - p * DS;
            const int64_t i = i0 + p;
            const bool ok = (i < dm.nb) && (k < D);
            gx_lds[e] = ok ? gx_in[i * D + k] : 0.f;
            eps_lds[e] = ok ? eps_in[i * D + k] : 0.f;
        }
        for (int e = tid; e < ENC_P * D; e += ENC_THREADS) {
            const int k = e / ENC_P, p = e - k * ENC_P;
            const int64_t i = i0 + p;
            ld_lds[p * DS + k] = (i < dm.nb) ? ldT[(int64_t)k * dm.nb + i] : 0.f;
        }
        for (int e = tid; e < ENC_P * Hp; e += ENC_THREADS) {
            const int p = e / Hp, hh = e - p * Hp;
            const int64_t i = i0 + p;
            h_lds[p * HS + hh] = (i < dm.nb && hh < H) ? h_in[i * H + hh] : 0.f;
        }
        __syncthreads();
        const int pvalid = (int)((dm.nb - i0) < ENC_P ? (dm.nb - i0) : ENC_P);
#pragma unroll 2
        for (int s = 0; s < ENC_P / 2; ++s) {
            const int p = 2 * s + half;
            const float* gx_p = gx_lds + p * DS;
            const float* eps_p = eps_lds + p * DS;
            const float* ld_p = ld_lds + p * DS;
            float bv[HT];
#pragma unroll
            for (int ht = 0; ht < HT; ++ht) bv[ht] = h_lds[p * HS + 32 * ht + l31];
#pragma unroll
            for (int t = 0; t < BW_RT; ++t) {
                float v = enc_v(code[t], gx_p, eps_p, ld_p, scale);
                if (p >= pvalid) v = 0.f;                          // rows past the batch carry no "+scale"
                bsum[t] += v;
#pragma unroll
                for (int ht = 0; ht < HT; ++ht) acc[t][ht] = mfma32(v, bv[ht], acc[t][ht]);
            }
        }
    }
    float* slab = slabs + (int64_t)blockIdx.y * slab_len;
    float* sW21 = slab;
    float* sb21 = sW21 + (int64_t)D * H;
    float* sW22 = sb21 + D;
    float* sb22 = sW22 + (int64_t)T * H;
#pragma unroll
    for (int t = 0; t < BW_RT; ++t) {
#pragma unroll
        for (int ht = 0; ht < HT; ++ht) {
            const int hh = 32 * ht + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = rbase + 32 * t + crow32(r, half);
                if (hh < H) {
                    if (row < T) sW22[row * H + hh] = acc[t][ht][r];
                    else if (row < RT) sW21[(row - T) * H + hh] = acc[t][ht][r];
                }
            }
        }
        const float bt = bsum[t] + __shfl_xor(bsum[t], 32, 64);
        const int64_t row = rbase + 32 * t + l31;
        if (half == 0) {
            if (row < T) sb22[row] = bt;
            else if (row < RT) sb21[row - T] = bt;
        }
    }
}

// ---------------------------------------------------------------------------------------------
#define FC1_IT 4                        // item tiles (of 32) per wave -> 512 items per workgroup
#define FC1_JG (4 * FC1_IT * 32)
#define FC1_YS (FC1_JG + 4)

__host__ __device__ inline size_t fc1_bwd_lds_floats(int Hp) {
    return (size_t)ENC_P * (Hp + 1) + (size_t)ENC_P * FC1_YS / 4;
}

// slab layout: [W1: H*J | b1: H]
// fast != 0 (H == 64, J % 4 == 0, aligned ghpre / y): tiles are fetched as batches of independent 16-byte /
// 4-byte loads one person tile ahead (issue before the MFMA phase, write to LDS after it).
// IT: item tiles of 32 a wave -- 128 IT items a workgroup.  IT = 4 is the form for large batches; IT = 1 (round 5) gives a small
// batch four times the workgroups: at B = 100 the kernel was two workgroups, each with 8 fp32 MFMAs a step and a 128 KB slab.
// (vbx, vby, vgy: the workgroup's place in a grid that may be a virtual one, see k_bwd_wt_fc1)
template <int HT, int IT>
__device__ __forceinline__ void fc1_bwd_body(
    const EncDims& dm, const uint8_t* __restrict__ y, const int64_t* __restrict__ rows,
    const float* __restrict__ ghpre, float* __restrict__ slabs, int64_t slab_len, int fast, float* smem, int vbx, int vby, int vgy) {
    const int J = dm.J, H = dm.H, Hp = dm.Hp;
    const int HS = fast ? 64 : Hp + 1;
    float* g_lds = smem;                                   // [P][HS]
    int8_t* Yi = (int8_t*)(g_lds + ENC_P * (Hp + 1));      // [P][YSI] encoder input as int8 (-1,0,1)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    constexpr int JG = 4 * IT * 32, YSI = JG + 4, WPP = JG / 4;          // items, LDS row stride, response words a person
    const int jg0 = vbx * JG;
    f32x16 acc[IT][HT];
    float bsum[HT];
#pragma unroll
    for (int ht = 0; ht < HT; ++ht) bsum[ht] = 0.f;
#pragma unroll
    for (int t = 0; t < IT; ++t)
#pragma unroll
        for (int ht = 0; ht < HT; ++ht) acc[t][ht] = zero16();
    const int64_t n_ptiles = (dm.nb + ENC_P - 1) / ENC_P;
    float4 pg[4];
    uint32_t pw[8 * IT];
    auto prefetch = [&](int64_t tile) {
        const int64_t i0 = tile * ENC_P;
        const int pv = (int)((dm.nb - i0) < ENC_P ? (dm.nb - i0) : ENC_P);
        const float4* g4 = (const float4*)(ghpre + i0 * 64);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = tid + ENC_THREADS * q;
            pg[q] = (idx < pv * 16) ? g4[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < 8 * IT; ++q) {
            const int idx = tid + ENC_THREADS * q;                  // P x WPP words
            const int p = idx / WPP, jw = jg0 + 4 * (idx % WPP);
            pw[q] = 0u;
            if (p < pv && jw < J) {
                const int64_t row = rows ? rows[i0 + p] : i0 + p;
                pw[q] = *(const uint32_t*)(y + row * J + jw);       // bytes 0/1/255 == int8 0/1/-1 (vi.py:689-691)
            }
        }
    };
    int64_t tile = vby;
    if (fast && tile < n_ptiles) prefetch(tile);
    for (; tile < n_ptiles; tile += vgy) {
        const int64_t i0 = tile * ENC_P;
        __syncthreads();
        if (fast) {
#pragma unroll
            for (int q = 0; q < 4; ++q) ((float4*)g_lds)[tid + ENC_THREADS * q] = pg[q];
#pragma unroll
            for (int q = 0; q < 8 * IT; ++q) {
                const int idx = tid + ENC_THREADS * q;
                ((uint32_t*)Yi)[(idx / WPP) * (YSI / 4) + (idx % WPP)] = pw[q];
            }
        } else {
            for (int e = tid; e < ENC_P * Hp; e += ENC_THREADS) {
                const int p = e / Hp, hh = e - p * Hp;
                const int64_t i = i0 + p;
                g_lds[p * HS + hh] = (i < dm.nb && hh < H) ? ghpre[i * H + hh] : 0.f;
            }
            for (int e = tid; e < ENC_P * JG; e += ENC_THREADS) {
                const int p = e / JG, jj = e - p * JG;
                const int64_t i = i0 + p;
                int8_t v = 0;
                if (i < dm.nb && jg0 + jj < J) {
                    const int64_t row = rows ? rows[i] : i;
                    const unsigned yy = y[row * J + jg0 + jj];
                    v = (yy == 255u) ? (int8_t)-1 : (int8_t)yy;
                }
                Yi[p * YSI + jj] = v;
            }
        }
        __syncthreads();
        if (fast && tile + vgy < n_ptiles) prefetch(tile + vgy);
#pragma unroll 2
        for (int s = 0; s < ENC_P / 2; ++s) {
            const int p = 2 * s + half;
            float av[HT];
#pragma unroll
            for (int ht = 0; ht < HT; ++ht) { av[ht] = g_lds[p * HS + 32 * ht + l31]; bsum[ht] += av[ht]; }
#pragma unroll
            for (int t = 0; t < IT; ++t) {
                const float bv = (float)Yi[p * YSI + 32 * (wave * IT + t) + l31];
#pragma unroll
                for (int ht = 0; ht < HT; ++ht) acc[t][ht] = mfma32(av[ht], bv, acc[t][ht]);
            }
        }
    }
    float* slab = slabs + (int64_t)vby * slab_len;
#pragma unroll
    for (int t = 0; t < IT; ++t) {
        const int j = jg0 + 32 * (wave * IT + t) + l31;
#pragma unroll
        for (int ht = 0; ht < HT; ++ht)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int hh = 32 * ht + crow32(r, half);
                if (hh < H && j < J) slab[(int64_t)hh * J + j] = acc[t][ht][r];
            }
    }
    if (vbx == 0 && wave == 0) {
#pragma unroll
        for (int ht = 0; ht < HT; ++ht) {
            const float bt = bsum[ht] + __shfl_xor(bsum[ht], 32, 64);
            const int hh = 32 * ht + l31;
            if (half == 0 && hh < H) slab[(int64_t)H * J + hh] = bt;
        }
    }
}
template <int HT, int IT = 4>
__global__ __launch_bounds__(ENC_THREADS) void k_fc1_bwd(
    EncDims dm, const uint8_t* __restrict__ y, const int64_t* __restrict__ rows,
    const float* __restrict__ ghpre, float* __restrict__ slabs, int64_t slab_len, int fast) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    fc1_bwd_body<HT, IT>(dm, y, rows, ghpre, slabs, slab_len, fast, smem, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y);
}
