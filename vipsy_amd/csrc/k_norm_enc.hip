// Amortized Normal guide for one latent dimension (NormEncoder, vi.py:417-435; used by VaeIRT with
// x_feature == 1, vi.py:677-684, and by VaeCHoDina, vi.py:968-981):
//     h = softplus(fc1 yin);  loc = fc21 h;  raw = fc22 h  (scale = exp(raw), vi.py:434)
// Forward: fc1 on fp32 MFMA (same tiling as phase A of k_mvn_enc_fwd), the two 1-row heads on the VALU.
// Backward: ghpre and the head gradients here; the fc1 weight gradient reuses k_fc1_bwd.
#pragma once
#include "k_mvn_enc.hip"

__host__ __device__ inline size_t norm_enc_fwd_lds_floats(int Hp) {
    return (size_t)ENC_P * (Hp + 1) + (size_t)ENC_P * (ENC_JC + 1) + (size_t)Hp * (ENC_JC + 1);
}

template <int HT>
__global__ __launch_bounds__(ENC_THREADS) void k_norm_enc_fwd(
    EncDims dm, const uint8_t* __restrict__ y, const int64_t* __restrict__ rows, const float* __restrict__ W1,
    const float* __restrict__ b1, const float* __restrict__ W21, const float* __restrict__ b21,
    const float* __restrict__ W22, const float* __restrict__ b22, float* __restrict__ h_out,
    float* __restrict__ loc_out, float* __restrict__ raw_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int J = dm.J, H = dm.H, Hp = dm.Hp;
    const int HS = Hp + 1;
    float* h_lds = smem;                                   // [P][HS]
    float* Yf = h_lds + ENC_P * HS;                        // [P][JC+1]
    float* W1c = Yf + ENC_P * (ENC_JC + 1);                // [Hp][JC+1]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int64_t i0 = (int64_t)blockIdx.x * ENC_P;
    constexpr int TPW = (HT + 1) / 2;
    f32x16 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[t] = zero16();
    const int u = wave & 1;
    for (int jc = 0; jc < J; jc += ENC_JC) {
        for (int e = tid; e < ENC_P * ENC_JC; e += ENC_THREADS) {
            const int p = e / ENC_JC, jj = e - p * ENC_JC;
            const int64_t i = i0 + p;
            float v = 0.f;
            if (i < dm.nb && jc + jj < J) {
                const int64_t row = rows ? rows[i] : i;
                const unsigned yy = y[row * J + jc + jj];
                v = (yy == 255u) ? -1.0f : (float)yy;       // vi.py:680-682
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
                    hv = softplusf_(acc[t][r] + b1[hh]);                  // vi.py:432
                    if (i < dm.nb) h_out[i * H + hh] = hv;
                }
                h_lds[p * HS + hh] = hv;
            }
        }
    }
    __syncthreads();
    if (tid < 2 * ENC_P) {                                                // heads: fc21 -> loc, fc22 -> raw
        const int p = tid >> 1, which = tid & 1;
        const int64_t i = i0 + p;
        if (i < dm.nb) {
            const float* w = which ? W22 : W21;
            float s = which ? b22[0] : b21[0];
            for (int hh = 0; hh < H; ++hh) s += w[hh] * h_lds[p * HS + hh];
            (which ? raw_out : loc_out)[i] = s;
        }
    }
}

// ghpre[i][hh] = d ELBO / d pre = -(gloc_i W21[hh] + graw_i W22[hh]) * sigmoid(pre),  sigmoid(pre) = 1 - exp(-h)
// head-gradient slab per block (d ELBO): [W21: H | b21: 1 | W22: H | b22: 1]
__global__ __launch_bounds__(256) void k_norm_enc_bwd_small(
    int H, int64_t nb, const float* __restrict__ W21, const float* __restrict__ W22, const float* __restrict__ h,
    const float* __restrict__ gloc, const float* __restrict__ graw, float* __restrict__ ghpre,
    float* __restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) float smem[];            // [256][2]
    const int tid = threadIdx.x;
    const int Hr = 1;                                                        // threads are laid out [persons][hh]
    (void)Hr;
    int hp = 1;
    while (hp < H) hp <<= 1;                                                 // hh slots per person row (pow2 >= H)
    if (hp > 256) hp = 256;
    const int ppb = 256 / hp;                                                // persons per block iteration
    const int hh = tid % hp, sub = tid / hp;
    float g21 = 0.f, g22 = 0.f, gb21 = 0.f, gb22 = 0.f;
    for (int hh0 = 0; hh0 < H; hh0 += hp) {                                  // H > 256: several passes over hh
        const int hcur = hh0 + hh;
        const float w21 = hcur < H ? W21[hcur] : 0.f, w22 = hcur < H ? W22[hcur] : 0.f;
        float a21 = 0.f, a22 = 0.f;
        for (int64_t i = (int64_t)blockIdx.x * ppb + sub; i < nb; i += (int64_t)gridDim.x * ppb) {
            const float gl = gloc[i], gr = graw[i];
            if (hcur < H) {
                const float hv = h[i * H + hcur];
                ghpre[i * H + hcur] = -(gl * w21 + gr * w22) * (1.0f - __expf(-hv));
                a21 -= gl * hv;
                a22 -= gr * hv;
            }
            if (hh0 == 0 && hh == 0) { gb21 -= gl; gb22 -= gr; }
        }
        g21 = a21; g22 = a22;
        // reduce over the `sub` rows of the block
        smem[2 * tid] = g21; smem[2 * tid + 1] = g22;
        __syncthreads();
        if (sub == 0 && hcur < H) {
            float s1 = 0.f, s2 = 0.f;
            for (int q = 0; q < ppb; ++q) { s1 += smem[2 * (q * hp + hh)]; s2 += smem[2 * (q * hp + hh) + 1]; }
            float* slab = slabs + (int64_t)blockIdx.x * (2 * H + 2);
            slab[hcur] = s1;
            slab[H + 1 + hcur] = s2;
        }
        __syncthreads();
    }
    smem[2 * tid] = gb21; smem[2 * tid + 1] = gb22;
    __syncthreads();
    if (tid == 0) {
        float s1 = 0.f, s2 = 0.f;
        for (int q = 0; q < ppb; ++q) { s1 += smem[2 * (q * hp)]; s2 += smem[2 * (q * hp) + 1]; }
        float* slab = slabs + (int64_t)blockIdx.x * (2 * H + 2);
        slab[H] = s1;
        slab[2 * H + 1] = s2;
    }
}

// The same for H == 64 with ghpre written DIMENSION-MAJOR (ghpreT[hh][nb], what k_fc1_bwd_b / k_fc1_bwd_t read): a block
// takes 64 consecutive persons at a time, thread (hh, quarter) computes 16 of them, the 64 x 64 tile turns in LDS and goes
// out as 256-byte rows.  Saves the person-major copy and the transpose pass over it.  Slab layout as above.
__global__ __launch_bounds__(256) void k_norm_enc_bwd_t64(
    int64_t nb, const float* __restrict__ W21, const float* __restrict__ W22, const float* __restrict__ h,
    const float* __restrict__ gloc, const float* __restrict__ graw, float* __restrict__ ghpreT,
    float* __restrict__ slabs) {
    __shared__ float T[64][65];
    __shared__ float red[2][4][64];
    __shared__ float gl_s[64], gr_s[64];
    const int tid = threadIdx.x, hh = tid & 63, pq = tid >> 6;
    const float w21 = W21[hh], w22 = W22[hh];
    float a21 = 0.f, a22 = 0.f, gb21 = 0.f, gb22 = 0.f;
    const int64_t n_tiles = (nb + 63) / 64;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t i0 = tile * 64;
        if (tid < 64) {
            const int64_t i = i0 + tid;
            gl_s[tid] = i < nb ? gloc[i] : 0.f;
            gr_s[tid] = i < nb ? graw[i] : 0.f;
        }
        __syncthreads();
#pragma unroll 4
        for (int q = 0; q < 16; ++q) {
            const int p = pq + 4 * q;
            const int64_t i = i0 + p;
            float g = 0.f;
            if (i < nb) {
                const float hv = h[i * 64 + hh];
                const float gl = gl_s[p], gr = gr_s[p];
                g = -(gl * w21 + gr * w22) * (1.0f - __expf(-hv));
                a21 -= gl * hv;
                a22 -= gr * hv;
                if (hh == 0) { gb21 -= gl; gb22 -= gr; }
            }
            T[p][hh] = g;
        }
        __syncthreads();
        {
            const int p = tid & 63, hq = tid >> 6;
            const int64_t i = i0 + p;
            if (i < nb) {
#pragma unroll 4
                for (int q = 0; q < 16; ++q) {
                    const int r = hq + 4 * q;
                    ghpreT[(int64_t)r * nb + i] = T[p][r];
                }
            }
        }
        __syncthreads();
    }
    red[0][pq][hh] = a21; red[1][pq][hh] = a22;
    __syncthreads();
    float* slab = slabs + (int64_t)blockIdx.x * (2 * 64 + 2);
    if (pq == 0) {
        slab[hh] = (red[0][0][hh] + red[0][1][hh]) + (red[0][2][hh] + red[0][3][hh]);
        slab[64 + 1 + hh] = (red[1][0][hh] + red[1][1][hh]) + (red[1][2][hh] + red[1][3][hh]);
    }
    __syncthreads();
    red[0][pq][hh] = gb21; red[1][pq][hh] = gb22;
    __syncthreads();
    if (tid == 0) {
        slab[64] = (red[0][0][0] + red[0][1][0]) + (red[0][2][0] + red[0][3][0]);
        slab[2 * 64 + 1] = (red[1][0][0] + red[1][1][0]) + (red[1][2][0] + red[1][3][0]);
    }
}
