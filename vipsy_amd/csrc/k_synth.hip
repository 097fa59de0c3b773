// On-device synthesis of response matrices with the distributions of the reference's Random* generators
// (vi.py:120-412), straight into the uint8 storage contract (0 / 1 / 255 = missing): benchmark / test INPUT, never inside
// a timed region.  Every draw is a Philox4x32-10 word keyed by the GLOBAL person id, so a data set does not depend on how
// the persons are sharded (restated in oracle/vi_oracle.py::synth_*):
//   latent x[i][k]        : philox_normal4(seed, step 0, stream SY_X, gid, k >> 2)[k & 3]            (unless x is supplied)
//   response uniform      : u01(word j & 3 of block j >> 2, stream SY_Y);  y = u < P
//   missing-cell uniform  : the same with stream SY_M;  u < rate -> 255
//   attribute uniform     : word k & 3 of block k >> 2, stream SY_A
#pragma once
#include "vx_common.h"

#define SY_X 0xE0u
#define SY_Y 0xD1u
#define SY_M 0xD2u
#define SY_A 0xE1u
#define SY_P 64                                    // persons per block tile

__device__ __forceinline__ float sy_uniform(uint64_t seed, uint32_t stream, int64_t gid, int idx) {
    const u32x4 w = philox4x32_10((uint32_t)gid, (uint32_t)((uint64_t)gid >> 32), 0u, (stream << 16) | (uint32_t)(idx >> 2),
                                  (uint32_t)seed, (uint32_t)(seed >> 32));
    const int c = idx & 3;
    return u01(c == 0 ? w.x : c == 1 ? w.y : c == 2 ? w.z : w.w);
}

// IRT 1..4PL, any latent dimension D <= 128 (irt_1pl..4pl, vi.py:22-66): P = c + (d - c) sigmoid(Dc (x.a + b)).
// A thread owns one item and accumulates the 64 logits of the person tile while `a` streams through once per tile.
__global__ __launch_bounds__(256) void k_synth_irt(int model, int D, int J, float Dc, int64_t nb, int64_t gid0,
                                                   const float* __restrict__ x_in /*[nb][D] or null*/, const float* __restrict__ a,
                                                   const float* __restrict__ b, const float* __restrict__ c, const float* __restrict__ d,
                                                   float missing, uint64_t seed, uint8_t* __restrict__ y, float* __restrict__ x_out) {
    extern __shared__ __attribute__((aligned(16))) float xs[];          // [SY_P][D]
    const int tid = threadIdx.x;
    for (int64_t i0 = (int64_t)blockIdx.x * SY_P; i0 < nb; i0 += (int64_t)gridDim.x * SY_P) {
        const int pv = (int)((nb - i0) < SY_P ? (nb - i0) : SY_P);
        __syncthreads();
        for (int e = tid; e < SY_P * D; e += 256) {
            const int p = e / D, k = e - p * D;
            float v = 0.f;
            if (p < pv) {
                v = x_in ? x_in[(i0 + p) * D + k] : philox_normal4(seed, 0u, SY_X, gid0 + i0 + p, (uint32_t)(k >> 2))[k & 3];
                if (x_out) x_out[(i0 + p) * D + k] = v;
            }
            xs[e] = v;
        }
        __syncthreads();
        for (int j = tid; j < J; j += 256) {
            float z[SY_P];
#pragma unroll
            for (int p = 0; p < SY_P; ++p) z[p] = 0.f;
            if (model >= 2) {
                for (int k = 0; k < D; ++k) {
                    const float ak = a[(int64_t)k * J + j];
#pragma unroll
                    for (int p = 0; p < SY_P; ++p) z[p] = fmaf(xs[p * D + k], ak, z[p]);
                }
            } else {
#pragma unroll
                for (int p = 0; p < SY_P; ++p) z[p] = xs[p * D];          // 1PL: one latent dimension, no slope (vi.py:29)
            }
            const float bj = b[j], cj = model >= 3 ? c[j] : 0.f, dj = model == 4 ? d[j] : 1.f;
#pragma unroll 4
            for (int p = 0; p < SY_P; ++p) {
                if (p < pv) {
                    const float P = cj + (dj - cj) * sigmoidf_(Dc * (z[p] + bj));
                    const int64_t gid = gid0 + i0 + p;
                    uint8_t v = sy_uniform(seed, SY_Y, gid, j) < P ? 1 : 0;
                    if (missing > 0.f && sy_uniform(seed, SY_M, gid, j) < missing) v = 255;
                    y[(i0 + p) * J + j] = v;
                }
            }
        }
    }
}

// DINA / DINO / HO-DINA (vi.py:69-116, 134-199); one lane = one person.  hodina: attr_k ~ Bern(sigmoid(theta lam1_k + lam0_k)),
// theta ~ N(0, 1); else attr_k ~ Bern(attr_p).  dino = the reference's dino() with its in-place sequencing (vi.py:96-100).
__global__ __launch_bounds__(256) void k_synth_cdm(int K, int J, int dino, int hodina, float attr_p, int64_t nb, int64_t gid0,
                                                   const float* __restrict__ q, const float* __restrict__ g, const float* __restrict__ s,
                                                   const float* __restrict__ lam0, const float* __restrict__ lam1, float missing,
                                                   uint64_t seed, uint8_t* __restrict__ y, uint8_t* __restrict__ attr_out,
                                                   float* __restrict__ theta_out) {
    extern __shared__ __attribute__((aligned(16))) uint32_t req[];      // [J]: mask | count << 16
    for (int j = threadIdx.x; j < J; j += blockDim.x) {
        uint32_t m = 0, cnt = 0;
        for (int k = 0; k < K; ++k)
            if (q[(int64_t)k * J + j] != 0.f) { m |= 1u << k; ++cnt; }
        req[j] = m | (cnt << 16);
    }
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t gid = gid0 + i;
        float th = 0.f;
        if (hodina) {
            th = philox_normal4(seed, 0u, SY_X, gid, 0u)[0];
            if (theta_out) theta_out[i] = th;
        }
        uint32_t abits = 0;
        for (int k = 0; k < K; ++k) {
            const float p = hodina ? sigmoidf_(th * lam1[k] + lam0[k]) : attr_p;
            const bool bit = sy_uniform(seed, SY_A, gid, k) < p;
            if (bit) abits |= 1u << k;
            if (attr_out) attr_out[i * K + k] = bit ? 1 : 0;
        }
        for (int j = 0; j < J; ++j) {
            const uint32_t rq = req[j], m = rq & 0xffffu, cnt = rq >> 16;
            const uint32_t miss = __builtin_popcount(m & ~abits);
            const bool eta = dino ? (cnt > 1 && miss < cnt) : (miss == 0);
            const float P = eta ? 1.0f - s[j] : g[j];
            uint8_t v = sy_uniform(seed, SY_Y, gid, j) < P ? 1 : 0;
            if (missing > 0.f && sy_uniform(seed, SY_M, gid, j) < missing) v = 255;
            y[i * J + j] = v;
        }
    }
}
