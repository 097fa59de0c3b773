// Utility kernels: Philox fill, fixed-order reductions, segmented Adam.
#pragma once
#include "vx_common.h"

__global__ void k_philox_normals(float* __restrict__ eps, const int64_t* __restrict__ gids, int64_t gid0,
                                 int64_t n, int D, uint64_t seed, uint32_t step, uint32_t stream) {
    const int nblk = (D + 3) >> 2;
    const int64_t total = n * nblk;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = e / nblk;
        const int blk = (int)(e - i * nblk);
        const int64_t gid = gids ? gids[i] : gid0 + i;
        f32x4 z = philox_normal4(seed, step, stream, gid, (uint32_t)blk);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (4 * blk + q < D) eps[i * D + 4 * blk + q] = z[q];
    }
}

__global__ void k_philox_raw(uint32_t* __restrict__ out, int64_t gid0, int64_t n, uint64_t seed, uint32_t step,
                             uint32_t stream) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t gid = gid0 + i;
        u32x4 w = philox4x32_10((uint32_t)gid, (uint32_t)((uint64_t)gid >> 32), step, stream << 16,
                                (uint32_t)seed, (uint32_t)(seed >> 32));
        out[4 * i + 0] = w.x; out[4 * i + 1] = w.y; out[4 * i + 2] = w.z; out[4 * i + 3] = w.w;
    }
}

// out[i] = alpha * sum_s slabs[s * stride + i].  Block = 64 columns x 4 slab groups (slab s goes to group s % 4,
// summed in ascending s; the four group sums are added in fixed order) -> deterministic, and parallel enough
// when there are many slabs but few columns (the D = 1 kernels: ~1000 slabs x 2000 columns).
__device__ __forceinline__ void reduce_slabs_cols(int blk, int nblk, float (*part)[64], const float* __restrict__ slabs,
                                                  int64_t n_slabs, int64_t stride, int64_t len, float alpha,
                                                  float* __restrict__ out) {
    const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
    for (int64_t c0 = (int64_t)blk * 64; c0 < len; c0 += (int64_t)nblk * 64) {
        const int64_t i = c0 + col;
        float acc = 0.f;
        if (i < len) {
#pragma unroll 8
            for (int64_t s = grp; s < n_slabs; s += 4) acc += slabs[s * stride + i];
        }
        part[grp][col] = acc;
        __syncthreads();
        if (grp == 0 && i < len) out[i] = alpha * ((part[0][col] + part[1][col]) + (part[2][col] + part[3][col]));
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k_reduce_slabs(const float* __restrict__ slabs, int64_t n_slabs, int64_t stride,
                                                      int64_t len, float alpha, float* __restrict__ out) {
    __shared__ float part[4][64];
    reduce_slabs_cols((int)blockIdx.x, (int)gridDim.x, part, slabs, n_slabs, stride, len, alpha, out);
}

// One launch behind an item-stationary likelihood kernel whose item chunks went to `groups` workgroups (k_irt_lik_r; on the
// reference's own B = 100 step three reductions, a clearing pass and the DIAG-row operand were five launches of ~5 us each):
//   blocks [0, nblk_gx)              gx = the chunk partials added in ascending chunk order; with gdT: gd = gx eps ld + scale
//   blocks [nblk_gx, + nblk_ll)      ll = the log-lik partials added the same way
//   the rest                         gitem = -(sum of the person ranges' item slabs), 64 columns a block as k_reduce_slabs;
//                                    columns from len_w on (the c / d leaves of a model that has none) are not written by the
//                                    likelihood kernel and come out as zeros: the slabs need no clearing
// Every sum has a fixed order: bit-reproducible.
__global__ __launch_bounds__(256) void k_lik_finish(const float* __restrict__ gx_part, int groups, int64_t n_gx, float* __restrict__ gx_sum,
                                                   const float* __restrict__ epsT, const float* __restrict__ ldT, float* __restrict__ gdT,
                                                   float scale, const float* __restrict__ ll_part, int64_t nb, float* __restrict__ ll,
                                                   const float* __restrict__ slabs, int64_t n_slabs, int64_t slab_len, int64_t len_w,
                                                   float* __restrict__ gitem, int nblk_gx, int nblk_ll) {
    const int bid = blockIdx.x;
    if (bid < nblk_gx) {
        for (int64_t i = (int64_t)bid * 256 + threadIdx.x; i < n_gx; i += (int64_t)nblk_gx * 256) {
            float acc = gx_part[i];
            for (int g = 1; g < groups; ++g) acc += gx_part[(int64_t)g * n_gx + i];
            gx_sum[i] = acc;
            if (gdT) gdT[i] = fmaf(acc * epsT[i], ldT[i], scale);
        }
        return;
    }
    if (bid < nblk_gx + nblk_ll) {
        for (int64_t i = (int64_t)(bid - nblk_gx) * 256 + threadIdx.x; i < nb; i += (int64_t)nblk_ll * 256) {
            float acc = ll_part[i];
            for (int g = 1; g < groups; ++g) acc += ll_part[(int64_t)g * nb + i];
            ll[i] = acc;
        }
        return;
    }
    __shared__ float part[4][64];
    const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int nblk_s = (int)gridDim.x - nblk_gx - nblk_ll;
    for (int64_t c0 = (int64_t)(bid - nblk_gx - nblk_ll) * 64; c0 < slab_len; c0 += (int64_t)nblk_s * 64) {
        const int64_t i = c0 + col;
        float acc = 0.f;
        if (i < len_w) {
#pragma unroll 8
            for (int64_t s2 = grp; s2 < n_slabs; s2 += 4) acc += slabs[s2 * slab_len + i];
        }
        part[grp][col] = acc;
        __syncthreads();
        if (grp == 0 && i < slab_len) gitem[i] = -((part[0][col] + part[1][col]) + (part[2][col] + part[3][col]));
        __syncthreads();
    }
}

// Many slabs, few columns, and the tail of a D = 1 step folded in: block = COLS columns x (1024 / COLS) slab groups (group g
// sums slabs g, g + GRPS, ... in ascending order; the group sums are added in a fixed order: ascending for COLS = 32, a
// fixed pairwise tree for COLS = 8) -> deterministic.  The last column may go to its own address (the loss slot), and the
// step counter of a captured step advances here.  COLS = 8 is for MANY slabs (the person-per-lane D = 1 kernel writes one
// per 64 persons while they fit the chip: 1 563 at 100 k persons): four times the workgroups and a quarter of the
// dependent loads a thread -- the kernel is a chain of load latencies, not bytes.
template <int COLS>
__global__ __launch_bounds__(1024) void k_reduce_wide(const float* __restrict__ slabs, int64_t n_slabs, int64_t stride,
                                                      int64_t len, float alpha, float* __restrict__ out,
                                                      float* __restrict__ last_out, uint32_t* __restrict__ tick) {
    constexpr int GRPS = 1024 / COLS;
    __shared__ float part[GRPS][COLS + 1];
    const int col = threadIdx.x % COLS, grp = threadIdx.x / COLS;
    if (tick && blockIdx.x == 0 && threadIdx.x == 0) *tick += 1u;
    for (int64_t c0 = (int64_t)blockIdx.x * COLS; c0 < len; c0 += (int64_t)gridDim.x * COLS) {
        const int64_t i = c0 + col;
        float acc = 0.f;
        if (i < len) {
#pragma unroll 8
            for (int64_t s = grp; s < n_slabs; s += GRPS) acc += slabs[s * stride + i];
        }
        part[grp][col] = acc;
        __syncthreads();
        if constexpr (COLS == 32) {
            if (grp == 0 && i < len) {
                float t = part[0][col];
#pragma unroll
                for (int g = 1; g < GRPS; ++g) t += part[g][col];
                if (last_out && i == len - 1) last_out[0] = alpha * t;
                else out[i] = alpha * t;
            }
        } else {
#pragma unroll
            for (int s = GRPS / 2; s > 0; s >>= 1) {
                if (grp < s) part[grp][col] += part[grp + s][col];
                __syncthreads();
            }
            if (grp == 0 && i < len) {
                const float t = part[0][col];
                if (last_out && i == len - 1) last_out[0] = alpha * t;
                else out[i] = alpha * t;
            }
        }
        __syncthreads();
    }
}

// few slabs, long rows (the 4 item-chunk partials of gx at 1M persons: 2 GB): 16-byte streaming loads, slabs added in
// ascending order -> deterministic; HBM-bound
__global__ __launch_bounds__(256) void k_reduce_few(const float4* __restrict__ slabs, int n_slabs, int64_t stride4,
                                                    int64_t len4, float alpha, float4* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 acc = slabs[i];
        for (int s = 1; s < n_slabs; ++s) {
            const float4 v = slabs[(int64_t)s * stride4 + i];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        out[i] = make_float4(alpha * acc.x, alpha * acc.y, alpha * acc.z, alpha * acc.w);
    }
}

// two-stage fixed-order sum: stage 1 -> partial[blockIdx], stage 2 (1 block) -> out[0]
__device__ __forceinline__ void sum_block(int blk, int nblk, float* red, const float* __restrict__ v, int64_t n,
                                          float* __restrict__ partial, const float* __restrict__ v2, float alpha,
                                          float* __restrict__ out, uint32_t* __restrict__ tick) {
    float acc = 0.f;
    const int64_t per = (n + nblk - 1) / nblk;
    const int64_t lo = (int64_t)blk * per;
    const int64_t hi = lo + per < n ? lo + per : n;
    if (v2) { for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) acc += v[i] + v2[i]; }
    else { for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) acc += v[i]; }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
        partial[blk] = t;
        if (out) {                                             // (nblk == 1)
            out[0] = alpha * t;
            if (tick) *tick += 1u;                             // (every kernel that reads it as the Philox step ran before this one)
        }
    }
}
__global__ void k_sum_stage1(const float* __restrict__ v, int64_t n, float* __restrict__ partial,
                             const float* __restrict__ v2 = nullptr /*optional second vector of the same length*/,
                             float alpha = 0.f, float* __restrict__ out = nullptr /*ONE block: the launch is the whole sum, out[0]
                             = alpha * total -- what stage 2 would make of one partial, bit for bit, without its launch*/,
                             uint32_t* __restrict__ tick = nullptr) {
    __shared__ float red[256 / VX_WAVE];
    sum_block((int)blockIdx.x, (int)gridDim.x, red, v, n, partial, v2, alpha, out, tick);
}
__global__ void k_sum_stage2(const float* __restrict__ partial, int n, float alpha, float* __restrict__ out,
                             uint32_t* __restrict__ tick = nullptr /*the device step counter: advanced by one*/) {
    __shared__ float red[256 / VX_WAVE];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc += partial[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
        out[0] = alpha * t;
        if (tick) *tick += 1u;                                 // (every kernel that reads it as the Philox step ran before this one)
    }
}

#define VX_MAX_SEGS 16
struct AdamSegs { int64_t begin[VX_MAX_SEGS]; int64_t end[VX_MAX_SEGS]; float lr[VX_MAX_SEGS]; int n; };

// torch.optim.Adam for element i of one buffer with gradient gi (k_adam / k_adam2 / k_reduce_adam: one arithmetic)
struct AdamBuf { float* p; const float* g; float* m; float* v; const float* free_mask; int64_t n; };
// The update of ONE element, written once and compiled without fused-multiply-add contraction: the scalar, the 16-byte and the
// operands-in-flight forms below inline it in different surroundings, and a contraction chosen differently in one of them would
// cost the last bit that tests/test_gpu_parity.py::test_optimiser_in_the_steps_last_launch_equals_separate_launches compares.
__device__ __forceinline__ void adam_math(float gi, float& p, float& m, float& v, float lr, float beta1, float beta2, float eps,
                                          float bc1, float bc2_sqrt) {
#pragma clang fp contract(off)
    const float mi = beta1 * m + (1.f - beta1) * gi;
    const float vi = beta2 * v + (1.f - beta2) * gi * gi;
    m = mi; v = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p = p - (lr / bc1) * (mi / denom);
}
__device__ __forceinline__ void adam_one(const AdamBuf& buf, const AdamSegs& segs, int64_t i, float gi, float beta1, float beta2,
                                         float eps, float bc1, float bc2_sqrt) {
    float lr = 0.f;
    bool found = false;
    for (int s = 0; s < segs.n; ++s)
        if (i >= segs.begin[s] && i < segs.end[s]) { lr = segs.lr[s]; found = true; }
    if (!found) return;
    if (buf.free_mask) gi *= buf.free_mask[i];
    float pi = buf.p[i], mi = buf.m[i], vi = buf.v[i];
    adam_math(gi, pi, mi, vi, lr, beta1, beta2, eps, bc1, bc2_sqrt);
    buf.m[i] = mi; buf.v[i] = vi; buf.p[i] = pi;
}

// Four consecutive elements i .. i + 3 of one buffer (i % 4 == 0, cnt = how many of them exist) in one thread: 16-byte loads and
// stores where the quad lies inside ONE learning-rate segment and the buffers are 16-byte aligned, the scalar update otherwise.
// (One element a thread moved 2 M per-person parameters in 19 us, 3 TB/s of 4-byte accesses: BASELINE config 5's optimiser
// launch.)  The arithmetic of an element is adam_one's, term for term.
__device__ __forceinline__ void adam_quad(const AdamBuf& buf, const AdamSegs& segs, int64_t i, int cnt, float beta1, float beta2,
                                          float eps, float bc1, float bc2_sqrt) {
    float lr = 0.f;
    bool whole = false;
    for (int s = 0; s < segs.n; ++s)
        if (i >= segs.begin[s] && i + 4 <= segs.end[s]) { lr = segs.lr[s]; whole = true; }
    const uintptr_t al = (uintptr_t)buf.p | (uintptr_t)buf.g | (uintptr_t)buf.m | (uintptr_t)buf.v | (uintptr_t)buf.free_mask;
    if (cnt == 4 && whole && (al & 15) == 0) {
        f32x4 g4 = *(const f32x4*)(buf.g + i);
        const f32x4 m4 = *(const f32x4*)(buf.m + i), v4 = *(const f32x4*)(buf.v + i), p4 = *(const f32x4*)(buf.p + i);
        if (buf.free_mask) {
            const f32x4 f4 = *(const f32x4*)(buf.free_mask + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) g4[e] *= f4[e];
        }
        f32x4 mo, vo, po;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float pe = p4[e], me = m4[e], ve = v4[e];
            adam_math(g4[e], pe, me, ve, lr, beta1, beta2, eps, bc1, bc2_sqrt);
            mo[e] = me; vo[e] = ve; po[e] = pe;
        }
        *(f32x4*)(buf.m + i) = mo; *(f32x4*)(buf.v + i) = vo; *(f32x4*)(buf.p + i) = po;
        return;
    }
    for (int e = 0; e < cnt; ++e) adam_one(buf, segs, i + e, buf.g[i + e], beta1, beta2, eps, bc1, bc2_sqrt);
}

// torch.optim.Adam: p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
__global__ void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                       float* __restrict__ v, const float* __restrict__ free_mask, int64_t n, AdamSegs segs,
                       float beta1, float beta2, float eps, float bc1, float bc2_sqrt,
                       const uint32_t* __restrict__ t_dev, uint32_t t_host = 0, const float* __restrict__ loss_src = nullptr,
                       float* __restrict__ loss_ring = nullptr) {
    // the step's loss (all-reduced by now) into slot t of the ring: what step() returns stays valid for VX_LOSS_RING - 1
    // further steps, eager or replayed, without a launch of its own
    if (loss_ring && blockIdx.x == 0 && threadIdx.x == 0) loss_ring[(t_dev ? *t_dev : t_host) & (VX_LOSS_RING - 1)] = *loss_src;
    if (t_dev) {
        // replayed from a HIP graph: the step count lives in device memory; the bias corrections are made here, in
        // double like the host makes them, once per block
        __shared__ float bc[2];
        if (threadIdx.x == 0) {
            const double t = (double)*t_dev;
            bc[0] = (float)(1.0 - pow((double)beta1, t));
            bc[1] = (float)sqrt(1.0 - pow((double)beta2, t));
        }
        __syncthreads();
        bc1 = bc[0]; bc2_sqrt = bc[1];
    }
    const AdamBuf buf{p, g, m, v, free_mask, n};
    const int64_t nq = (n + 3) >> 2;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = 4 * q;
        adam_quad(buf, segs, i, (int)((n - i) < 4 ? (n - i) : 4), beta1, beta2, eps, bc1, bc2_sqrt);
    }
}

// the same update in two halves, so that a launch that still has to SUM the gradient has Adam's operands in flight meanwhile
// (one memory round trip instead of two behind each other)
struct AdamOperands { float p, m, v, fm, lr; bool found; };
__device__ __forceinline__ AdamOperands adam_fetch(bool on, const AdamBuf& buf, const AdamSegs& segs, int64_t i) {
    AdamOperands o{0.f, 0.f, 0.f, 1.f, 0.f, false};
    if (!on) return o;
    for (int s = 0; s < segs.n; ++s)
        if (i >= segs.begin[s] && i < segs.end[s]) { o.lr = segs.lr[s]; o.found = true; }
    if (!o.found) return o;
    o.p = buf.p[i]; o.m = buf.m[i]; o.v = buf.v[i];
    if (buf.free_mask) o.fm = buf.free_mask[i];
    return o;
}
__device__ __forceinline__ void adam_finish(const AdamBuf& buf, int64_t i, float gi, const AdamOperands& o, float beta1, float beta2,
                                            float eps, float bc1, float bc2_sqrt) {
    if (!o.found) return;
    if (buf.free_mask) gi *= o.fm;
    float pi = o.p, mi = o.m, vi = o.v;
    adam_math(gi, pi, mi, vi, o.lr, beta1, beta2, eps, bc1, bc2_sqrt);
    buf.m[i] = mi; buf.v[i] = vi; buf.p[i] = pi;
}

// The tail of a D = 1 step on ONE rank in ONE launch (k_reduce_wide + k_adam2, 5 us each at BASELINE config 2, a quarter of
// its step): blocks [0, n_red) sum the slabs column by column exactly as k_reduce_wide<COLS> does -- and hand every finished
// column straight to Adam (buffer A = the item leaves, gA = out) or, the last column, to the loss slot and the loss ring;
// the blocks behind them run Adam over buffer B (the per-person rows, whose gradients the step kernel wrote).  Adam's count
// comes from the word the step kernel left behind its slabs (t_copy; a captured step) or from the host (t_host): nobody in
// this launch reads the device step counter, so block 0 may advance it.  Same sums, same arithmetic, same bits.
template <int COLS>
__global__ __launch_bounds__(1024) void k_reduce_adam(const float* __restrict__ slabs, int64_t n_slabs, int64_t stride, int64_t len,
                                                      float alpha, float* __restrict__ out, float* __restrict__ last_out,
                                                      uint32_t* __restrict__ tick, const uint32_t* __restrict__ t_copy, uint32_t t_host,
                                                      AdamBuf A, AdamSegs sA, AdamBuf B, AdamSegs sB, float beta1, float beta2, float eps,
                                                      float bc1, float bc2_sqrt, float* __restrict__ loss_ring, int n_red) {
    constexpr int GRPS = 1024 / COLS;
    __shared__ float part[GRPS][COLS + 1];
    __shared__ float bc[2];
    const uint32_t tt = t_copy ? *t_copy : t_host;
    if (tick && blockIdx.x == 0 && threadIdx.x == 0) *tick += 1u;
    if (t_copy) {
        if (threadIdx.x == 0) {
            const double t = (double)tt;
            bc[0] = (float)(1.0 - pow((double)beta1, t));
            bc[1] = (float)sqrt(1.0 - pow((double)beta2, t));
        }
        __syncthreads();
        bc1 = bc[0]; bc2_sqrt = bc[1];
    }
    if ((int)blockIdx.x >= n_red) {
        const int64_t nblk = (int64_t)gridDim.x - n_red, nq = (B.n + 3) >> 2;
        for (int64_t q = ((int64_t)blockIdx.x - n_red) * 1024 + threadIdx.x; q < nq; q += nblk * 1024)      // four elements a thread
            adam_quad(B, sB, 4 * q, (int)((B.n - 4 * q) < 4 ? (B.n - 4 * q) : 4), beta1, beta2, eps, bc1, bc2_sqrt);
        return;
    }
    const int col = threadIdx.x % COLS, grp = threadIdx.x / COLS;
    AdamOperands ops;
    auto finish = [&](int64_t i, float t) __attribute__((always_inline)) {
        const float g = alpha * t;
        if (last_out && i == len - 1) {
            last_out[0] = g;
            if (loss_ring) loss_ring[tt & (VX_LOSS_RING - 1)] = g;
        } else {
            out[i] = g;
            adam_finish(A, i, g, ops, beta1, beta2, eps, bc1, bc2_sqrt);
        }
    };
    for (int64_t c0 = (int64_t)blockIdx.x * COLS; c0 < len; c0 += (int64_t)n_red * COLS) {
        const int64_t i = c0 + col;
        // (Adam's operands of this column in flight under the slab sum)
        ops = adam_fetch(grp == 0 && i < len && !(last_out && i == len - 1), A, sA, i);
        float acc = 0.f;
        if (i < len) {
#pragma unroll 8
            for (int64_t s = grp; s < n_slabs; s += GRPS) acc += slabs[s * stride + i];
        }
        part[grp][col] = acc;
        __syncthreads();
        if constexpr (COLS == 32) {
            if (grp == 0 && i < len) {
                float t = part[0][col];
#pragma unroll
                for (int g = 1; g < GRPS; ++g) t += part[g][col];
                finish(i, t);
            }
        } else {
#pragma unroll
            for (int s = GRPS / 2; s > 0; s >>= 1) {
                if (grp < s) part[grp][col] += part[grp + s][col];
                __syncthreads();
            }
            if (grp == 0 && i < len) finish(i, part[0][col]);
        }
        __syncthreads();
    }
}

// two parameter buffers in one launch (the replicated leaves and the per-person rows of a BBVI guide): indices
// [0, nA) -> buffer A (with its free mask), [nA, nA + nB) -> buffer B
__global__ void k_adam2(AdamBuf A, AdamSegs sA, AdamBuf B, AdamSegs sB, float beta1, float beta2, float eps, float bc1,
                        float bc2_sqrt, const uint32_t* __restrict__ t_dev, uint32_t t_host = 0,
                        const float* __restrict__ loss_src = nullptr, float* __restrict__ loss_ring = nullptr) {
    if (loss_ring && blockIdx.x == 0 && threadIdx.x == 0) loss_ring[(t_dev ? *t_dev : t_host) & (VX_LOSS_RING - 1)] = *loss_src;
    if (t_dev) {
        __shared__ float bc[2];
        if (threadIdx.x == 0) {
            const double t = (double)*t_dev;
            bc[0] = (float)(1.0 - pow((double)beta1, t));
            bc[1] = (float)sqrt(1.0 - pow((double)beta2, t));
        }
        __syncthreads();
        bc1 = bc[0]; bc2_sqrt = bc[1];
    }
    // quads of A, then quads of B: each buffer's quads start at ITS element 0 (16-byte aligned whatever A.n is)
    const int64_t qA = (A.n + 3) >> 2, qB = (B.n + 3) >> 2;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < qA + qB; q += (int64_t)gridDim.x * blockDim.x) {
        // (two calls, not one call on a buffer chosen at run time: a kernel-argument structure selected by reference is copied to
        // scratch memory -- the optimiser launch of BASELINE config 5 took 67 us that way, 19 before it)
        if (q < qA) {
            adam_quad(A, sA, 4 * q, (int)((A.n - 4 * q) < 4 ? (A.n - 4 * q) : 4), beta1, beta2, eps, bc1, bc2_sqrt);
        } else {
            const int64_t i = 4 * (q - qA);
            adam_quad(B, sB, i, (int)((B.n - i) < 4 ? (B.n - i) : 4), beta1, beta2, eps, bc1, bc2_sqrt);
        }
    }
}
