// VaeCCDM (vi.py:866-891): pattern-enumerated DINA / DINO whose pattern prior comes from the SoftmaxEncoder
// (vi.py:473-485) -- hidden = relu(fc1(data_)), attr_p = softmax(fc2(hidden), dim=0): the softmax runs over the BATCH, so
// the prior couples the persons of a step through one column maximum and one column sum per pattern (and, multi-GPU,
// through three all-reduces of C floats).  The enumeration itself is k_hodina (k_hodina.hip) in its third prior mode; this
// file holds the encoder and the batch-wise reductions.  None of this is a BASELINE configuration: plain, coalesced kernels.
#pragma once
#include "vx_common.h"

// h[nb][H] = relu(W1 yin + b1) (yin = response bytes as int8: 255 -> -1, vi.py:884-886), z[nb][C] = W2 h + b2;
// block = 256 / HS persons x HS hidden slots, H <= HS (64 or 128)
template <int HS>
__global__ __launch_bounds__(256) void k_sm_enc_fwd(int C, int J, int H, int64_t nb, const uint8_t* __restrict__ y,
                                                    const int64_t* __restrict__ rows, const float* __restrict__ W1,
                                                    const float* __restrict__ b1, const float* __restrict__ W2,
                                                    const float* __restrict__ b2, float* __restrict__ h, float* __restrict__ z) {
    constexpr int PPB = 256 / HS;
    __shared__ float hs[PPB][HS];
    const int tid = threadIdx.x, hh = tid % HS, sub = tid / HS;
    for (int64_t i0 = (int64_t)blockIdx.x * PPB; i0 < nb; i0 += (int64_t)gridDim.x * PPB) {
        const int64_t i = i0 + sub;
        float acc = 0.f;
        if (i < nb && hh < H) {
            const int64_t row = rows ? rows[i] : i;
            const uint8_t* yr = y + row * J;
            acc = b1[hh];
            for (int j = 0; j < J; ++j) acc = fmaf(W1[(int64_t)hh * J + j], (float)(int8_t)yr[j], acc);
            acc = fmaxf(acc, 0.f);
            h[i * H + hh] = acc;
        }
        hs[sub][hh] = (hh < H) ? acc : 0.f;
        __syncthreads();
        if (i < nb) {
            for (int c = hh; c < C; c += HS) {
                float a = b2[c];
                for (int t = 0; t < H; ++t) a = fmaf(W2[(int64_t)c * H + t], hs[sub][t], a);
                z[i * C + c] = a;
            }
        }
        __syncthreads();
    }
}

// column reductions over the batch rows of v[nb][C], fixed order (one partial row per block, then k_col_final):
//   mode 0: max_i v;  mode 1: sum_i exp(v - shift_c);  mode 2: sum_i v
__global__ __launch_bounds__(256) void k_col_part(int mode, const float* __restrict__ v, int64_t nb, int C,
                                                  const float* __restrict__ shift, float* __restrict__ part) {
    const int64_t per = (nb + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < nb ? lo + per : nb;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = mode == 0 ? -3.0e38f : 0.f;
        const float sh = mode == 1 ? shift[c] : 0.f;
        for (int64_t i = lo; i < hi; ++i) {
            const float t = v[i * C + c];
            acc = mode == 0 ? fmaxf(acc, t) : mode == 1 ? acc + __expf(t - sh) : acc + t;
        }
        part[(int64_t)blockIdx.x * C + c] = acc;
    }
}
__global__ void k_col_final(int mode, const float* __restrict__ part, int n_part, int C, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float acc = mode == 0 ? -3.0e38f : 0.f;
    for (int p = 0; p < n_part; ++p) acc = mode == 0 ? fmaxf(acc, part[(int64_t)p * C + c]) : acc + part[(int64_t)p * C + c];
    out[c] = acc;
}

// gz[i][c] = gla[i][c] - a[i][c] T[c],  a = exp(z - off): the batch softmax's backward (in place over gla)
__global__ void k_vaeccdm_gz(const float* __restrict__ z, const float* __restrict__ off, const float* __restrict__ T, int64_t nb,
                             int C, float* __restrict__ gla) {
    const int64_t n = nb * C;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        gla[e] = gla[e] - __expf(z[e] - off[c]) * T[c];
    }
}

// ghpre[i][hh] = (sum_c gz[i][c] W2[c][hh]) [h > 0]   (relu');  block = 256 / HS persons x HS hidden slots
template <int HS>
__global__ __launch_bounds__(256) void k_sm_enc_bwd_h(int C, int H, int64_t nb, const float* __restrict__ W2,
                                                      const float* __restrict__ h, const float* __restrict__ gz,
                                                      float* __restrict__ ghpre) {
    constexpr int PPB = 256 / HS;
    const int tid = threadIdx.x, hh = tid % HS, sub = tid / HS;
    for (int64_t i = (int64_t)blockIdx.x * PPB + sub; i < nb; i += (int64_t)gridDim.x * PPB) {
        if (hh < H) {
            float a = 0.f;
            for (int c = 0; c < C; ++c) a = fmaf(gz[i * C + c], W2[(int64_t)c * H + hh], a);
            ghpre[i * H + hh] = h[i * H + hh] > 0.f ? a : 0.f;
        }
    }
}

// head-gradient slab per (column tile of 64 patterns, row slab): gW2[c][hh] = sum_i gz[i][c] h[i][hh], gb2[c] = sum_i gz[i][c]
// slab layout: [W2: C*H | b2: C];  thread = (pattern c of the tile, group of HS / 4 hidden units)
template <int HS>
__global__ __launch_bounds__(256) void k_sm_enc_bwd_w(int C, int H, int64_t nb, const float* __restrict__ h,
                                                      const float* __restrict__ gz, float* __restrict__ slabs) {
    constexpr int HG = HS / 4;
    __shared__ float hs[16][HS];
    const int tid = threadIdx.x, cl = tid & 63, hg = tid >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int64_t per = (nb + gridDim.y - 1) / gridDim.y;
    const int64_t lo = (int64_t)blockIdx.y * per, hi = lo + per < nb ? lo + per : nb;
    float acc[HG], accb = 0.f;
#pragma unroll
    for (int t = 0; t < HG; ++t) acc[t] = 0.f;
    for (int64_t i0 = lo; i0 < hi; i0 += 16) {
        __syncthreads();
        for (int e = tid; e < 16 * HS; e += 256) {
            const int r = e / HS, t = e % HS;
            hs[r][t] = (i0 + r < hi && t < H) ? h[(i0 + r) * H + t] : 0.f;
        }
        __syncthreads();
        for (int r = 0; r < 16 && i0 + r < hi; ++r) {
            const float gv = c < C ? gz[(i0 + r) * C + c] : 0.f;
#pragma unroll
            for (int t = 0; t < HG; ++t) acc[t] = fmaf(gv, hs[r][HG * hg + t], acc[t]);
            if (hg == 0) accb += gv;
        }
    }
    float* slab = slabs + (int64_t)blockIdx.y * ((int64_t)C * H + C);
    if (c < C) {
#pragma unroll
        for (int t = 0; t < HG; ++t)
            if (HG * hg + t < H) slab[(int64_t)c * H + HG * hg + t] = acc[t];
        if (hg == 0) slab[(int64_t)C * H + c] = accb;
    }
}
