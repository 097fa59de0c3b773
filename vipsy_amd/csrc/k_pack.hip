// Packed head layout of the amortized MVN guide.
//
// The reference stores fc22's rows in torch.tril_indices order (vi.py:453): (0,0),(1,0),(1,1),(2,0),...  For the
// MFMA kernels a 32-row tile of that order mixes several (k, l) runs and forces a per-element decode.  The packed
// order (rebuilt from the flat parameters by k_pack_heads every step, 1.4 MB, a few microseconds) is
//   section OFF : for k = 1..D-1 the entries (k, 0..k-1), each k padded with zero rows to a multiple of 8;
//                 the section itself padded to a multiple of 192 rows (the forward kernel walks it 3 tiles at a time,
//                 the hidden-gradient kernel in 64-row tiles)
//   section DIAG: (k,k) for k = 0..D-1, padded to a multiple of 32
//   section LOC : fc21 rows k = 0..D-1, padded to a multiple of 32
//   tail        : zero rows up to a multiple of 64
// so a 32-row tile never mixes sections (the kernels run one branch-free loop per section),
// so every aligned group of 8 packed rows has ONE type and ONE k (OFF) and its l0 is a multiple of 8.
// gtab[group] = type << 28 | k << 12 | l0   (DIAG / LOC groups: k = first k of the group).
#pragma once
#include "vx_common.h"

#define PK_NONE 0u
#define PK_OFF 1u
#define PK_DIAG 2u
#define PK_LOC 3u

__host__ __device__ inline int pk_off_rows(int k) {          // packed rows before the entries of row k (k >= 1)
    const int m = k - 1, q = m / 8, rem = m % 8;
    return 8 * (8 * (q * (q + 1) / 2) + rem * (q + 1));
}
__host__ __device__ inline int pk_sec(int D) { return (D + 31) / 32 * 32; }
__host__ __device__ inline int pk_off_total(int D) { return (pk_off_rows(D) + 191) / 192 * 192; }   // 3-tile groups (forward) and 64-row tiles (backward)
__host__ __device__ inline int pk_rows(int D) { return (pk_off_total(D) + 2 * pk_sec(D) + 63) / 64 * 64; }

// packed row -> (source row in the concatenated [W22 rows 0..T-1 | W21 rows T..T+D-1] space, or -1), group code
__device__ __forceinline__ void pk_decode(int pr, int D, int T, int& src, uint32_t& gcode) {
    const int offT = pk_off_total(D), sec = pk_sec(D);
    src = -1;
    gcode = PK_NONE << 28;
    if (pr < offT) {
        if (pr < pk_off_rows(D)) {
            // the largest k <= D - 1 with pk_off_rows(k) <= pr.  Blocks of eight k: 32 q (q + 1) rows before k = 8 q + 1, then
            // 8 (q + 1) rows a k -- the block from a square root (and a step either way for its rounding), the k inside it by
            // a division.  (A linear search over k was up to 99 trips a wave: most of the 17-20 us of the two launches that
            // decode every packed row, at any batch size.)
            int q = (int)((sqrtf(1.0f + 0.125f * (float)pr) - 1.0f) * 0.5f);
            while (32 * (q + 1) * (q + 2) <= pr) ++q;
            while (q > 0 && 32 * q * (q + 1) > pr) --q;
            int k = 8 * q + 1 + (pr - 32 * q * (q + 1)) / (8 * (q + 1));
            if (k > D - 1) k = D - 1;
            const int l = pr - pk_off_rows(k);
            if (l < k) src = k * (k + 1) / 2 + l;
            gcode = (PK_OFF << 28) | ((uint32_t)k << 12) | (uint32_t)(l & ~7);
        } else {
            gcode = (PK_OFF << 28) | ((uint32_t)(D - 1) << 12);       // section padding: zero rows of the last k
        }
    } else if (pr < offT + sec) {
        const int k = pr - offT;
        if (k < D) src = k * (k + 1) / 2 + k;
        gcode = (PK_DIAG << 28) | ((uint32_t)(k & ~7) << 12);
    } else if (pr < offT + 2 * sec) {
        const int k = pr - offT - sec;
        if (k < D) src = T + k;
        gcode = (PK_LOC << 28) | ((uint32_t)(k & ~7) << 12);
    }
}

// Wp[Rp][H] (H floats per row), bp[Rp], gtab[Rp/8], WpT[H][Rp] (the transpose, for the hidden-gradient kernel)
__global__ void k_pack_heads(int D, int H, const float* __restrict__ W21, const float* __restrict__ b21,
                             const float* __restrict__ W22, const float* __restrict__ b22, float* __restrict__ Wp,
                             float* __restrict__ bp, uint32_t* __restrict__ gtab, float* __restrict__ WpT) {
    const int T = D * (D + 1) / 2, Rp = pk_rows(D);
    const int pr = blockIdx.x;
    if (pr >= Rp) return;
    int src;
    uint32_t gcode;
    pk_decode(pr, D, T, src, gcode);
    const float* w = (src < 0) ? nullptr : (src < T ? W22 + (int64_t)src * H : W21 + (int64_t)(src - T) * H);
    for (int hh = threadIdx.x; hh < H; hh += blockDim.x) {
        const float v = w ? w[hh] : 0.f;
        Wp[(int64_t)pr * H + hh] = v;
        if (WpT) WpT[(int64_t)hh * Rp + pr] = v;
    }
    if (threadIdx.x == 0) {
        if (bp) bp[pr] = (src < 0) ? 0.f : (src < T ? b22[src] : b21[src - T]);   // (null: a caller that needs Wp and gtab only)
        if ((pr & 7) == 0) gtab[pr >> 3] = gcode;
    }
}

// out = alpha * sum_s slabs[s] gathered from packed rows back to the reference layout
//   slab: [Wp-grad: Rp*H | bp-grad: Rp];  out: [W21: D*H | b21: D | W22: T*H | b22: T]
__device__ __forceinline__ void unpack_head_rows(int blk, int D, int H, const float* __restrict__ slabs, int n_slabs,
                                                 int64_t slab_len, float alpha, float* __restrict__ out) {
    // one wave per packed row (four rows a block): lane = hidden unit, the slabs summed in ascending order, eight loads
    // in flight per lane
    const int T = D * (D + 1) / 2, Rp = pk_rows(D);
    const int pr = blk * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (pr >= Rp) return;
    int src;
    uint32_t gcode;
    pk_decode(pr, D, T, src, gcode);
    if (src < 0) return;
    float* oW = (src < T) ? out + (int64_t)D * H + D + (int64_t)src * H : out + (int64_t)(src - T) * H;
    float* ob = (src < T) ? out + (int64_t)D * H + D + (int64_t)T * H + src : out + (int64_t)D * H + (src - T);
    for (int hh = lane; hh <= H; hh += 64) {
        float acc = 0.f;
        const float* sp = slabs + ((hh < H) ? (int64_t)pr * H + hh : (int64_t)Rp * H + pr);
#pragma unroll 8
        for (int s = 0; s < n_slabs; ++s) acc += sp[(int64_t)s * slab_len];
        if (hh < H) oW[hh] = alpha * acc; else *ob = alpha * acc;
    }
}
__global__ __launch_bounds__(256) void k_unpack_head_grads(int D, int H, const float* __restrict__ slabs, int n_slabs,
                                                           int64_t slab_len, float alpha, float* __restrict__ out) {
    unpack_head_rows((int)blockIdx.x, D, H, slabs, n_slabs, slab_len, alpha, out);
}
