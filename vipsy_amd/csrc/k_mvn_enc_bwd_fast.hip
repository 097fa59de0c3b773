// Fast paths of the amortized MVN guide backward for hidden_dim == 64 and D % 4 == 0 (same mathematics
// and outputs as k_mvn_enc_bwd_h / k_mvn_enc_bwd_w in k_mvn_enc_bwd.hip).  What changes is the staging:
// every global -> LDS copy is a batch of independent 16-byte loads held in registers across the previous
// MFMA phase (issue early, write late), tiles are exact linear images of the global rows, and bwd_h is
// sized for two workgroups per CU.
#pragma once
#include "k_mvn_enc_bwd.hip"
#include "k_mvn_enc_fast.hip"
#include "k_pack.hip"

#define BH_ROWS 64                      // head rows per LDS tile in bwd_h_fast

// gx / eps tiles are [P][D + 4]: slots D..D+3 hold {0,..} for gx and {1,0,0,0} for eps, so that with the row
// code of enc_row_code_fast every non-diagonal row is  V = gx[p][kx] * eps[p][lx]  with no branch.
__host__ __device__ inline size_t enc_bwdh_fast_lds_floats(int D) {
    return 2 * (size_t)ENC_P * enc_ds(D) + (size_t)BH_ROWS * 64 + BH_ROWS + 8;
}

// stage a [P][D] tile of contiguous global rows into a [P][DS] LDS image with ODD stride DS >= D + 1
// (lanes index persons in bwd_h, so an even stride would be an 8-way bank conflict); slot D := padv
__device__ __forceinline__ void stage_rows_odd(float* lds, const float* g, int D, int DS, int pvalid, float padv,
                                               int tid) {
    const int c4 = D / 4, n4 = ENC_P * c4;
    for (int base = 0; base < n4; base += ENC_THREADS * 4) {
        float4 a[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = base + q * ENC_THREADS + tid;
            a[q] = (idx < pvalid * c4) ? ((const float4*)g)[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = base + q * ENC_THREADS + tid;
            if (idx < n4) {
                const int p = idx / c4, c = idx - p * c4;
                float* dst = lds + p * DS + 4 * c;
                dst[0] = a[q].x; dst[1] = a[q].y; dst[2] = a[q].z; dst[3] = a[q].w;
            }
        }
    }
    if (tid < ENC_P) lds[tid * DS + D] = padv;
}

// stage a [P][D] tile of contiguous global rows into a [P][D+4] LDS image (pad = padv,0,0,0)
__device__ __forceinline__ void stage_rows_padded(float* lds, const float* g, int D, int pvalid, float padv,
                                                  int tid) {
    const int c4 = D / 4, n4 = ENC_P * c4;
    for (int base = 0; base < n4; base += ENC_THREADS * 4) {
        float4 a[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = base + q * ENC_THREADS + tid;
            a[q] = (idx < pvalid * c4) ? ((const float4*)g)[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = base + q * ENC_THREADS + tid;
            if (idx < n4) { const int p = idx / c4, c = idx - p * c4; ((float4*)lds)[p * (c4 + 1) + c] = a[q]; }
        }
    }
    if (tid < ENC_P) ((float4*)lds)[tid * (c4 + 1) + c4] = make_float4(padv, 0.f, 0.f, 0.f);
}

__global__ __launch_bounds__(ENC_THREADS, 2) void k_mvn_enc_bwd_h_fast(
    EncDims dm, float scale, const float* __restrict__ W21, const float* __restrict__ W22,
    const float* __restrict__ h_in, const float* __restrict__ eps_in, const float* __restrict__ ldT,
    const float* __restrict__ gx_in, float* __restrict__ ghpre_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int H = 64;
    const int D = dm.D, T = dm.T;
    const int DP = dm.DS;                              // odd stride >= D + 1
    float* gx_lds = smem;                              // [P][DS]
    float* eps_lds = gx_lds + ENC_P * DP;              // [P][DS]
    // [BH_ROWS][64]; offset rounded up to a multiple of 4 floats by INDEX arithmetic (a uintptr_t cast would
    // make the pointer generic and turn every LDS access below into a slow flat_load / flat_store)
    float* Wt = smem + ((2 * ENC_P * DP + 3) & ~3);
    uint32_t* rowtab = (uint32_t*)(Wt + BH_ROWS * H);  // [BH_ROWS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int64_t i0 = (int64_t)blockIdx.x * ENC_P;
    const int64_t RT = (int64_t)T + D;
    const int n_tiles = (int)((RT + BH_ROWS - 1) / BH_ROWS);
    const int pvalid = (int)((dm.nb - i0) < ENC_P ? (dm.nb - i0) : ENC_P);

    stage_rows_odd(gx_lds, gx_in + i0 * D, D, DP, pvalid, 0.f, tid);
    stage_rows_odd(eps_lds, eps_in + i0 * D, D, DP, pvalid, 1.f, tid);
    const int u = wave & 1, ht = wave >> 1;
    const int p = 32 * u + l31;
    const int64_t i = i0 + p;
    f32x16 acc = zero16();
    // W tile prefetch: 64 rows x 64 floats = 1024 float4, 4 per thread
    float4 wp[4];
    auto prefetch = [&](int tile) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int f = tid + ENC_THREADS * q;
            const int64_t r = (int64_t)tile * BH_ROWS + (f >> 4);
            const float* src = (r < T) ? W22 + r * H : (r < RT ? W21 + (r - T) * H : nullptr);
            wp[q] = src ? *(const float4*)(src + 4 * (f & 15)) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    prefetch(0);
    const float* gx_p = gx_lds + p * DP;
    const float* eps_p = eps_lds + p * DP;
    for (int tile = 0; tile < n_tiles; ++tile) {
        __syncthreads();                                // previous MFMA phase done with Wt / rowtab
#pragma unroll
        for (int q = 0; q < 4; ++q) ((float4*)Wt)[tid + ENC_THREADS * q] = wp[q];
        if (tid < BH_ROWS) rowtab[tid] = enc_row_code_fast((int64_t)tile * BH_ROWS + tid, T, D);
        __syncthreads();
        if (tile + 1 < n_tiles) prefetch(tile + 1);     // in flight during the MFMA phase
        const float* ap = Wt + (half * 32) * H + 32 * ht + l31;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint4 cd = *(const uint4*)(rowtab + half * 32 + 4 * q);
            const uint32_t cdv[4] = {cd.x, cd.y, cd.z, cd.w};
            float vv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) vv[e] = gx_p[(cdv[e] >> 16) & 0x7FFFu] * eps_p[cdv[e] & 0xFFFFu];
            if ((cd.x | cd.y | cd.z | cd.w) & FC_DIAG) {                          // half-wave uniform, rare
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (cdv[e] & FC_DIAG) {
                        const int k = (int)(cdv[e] & 0xFFFFu);
                        vv[e] = (i < dm.nb) ? vv[e] * ldT[(int64_t)k * dm.nb + i] + scale : 0.f;
                    }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = mfma32(ap[(4 * q + e) * H], vv[e], acc);
        }
    }
    // gh -> ghpre = gh * sigmoid(pre), sigmoid(pre) = 1 - exp(-h)
    if (i < dm.nb) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int hh0 = 32 * ht + 8 * g + 4 * half;
            const float4 hv = *(const float4*)(h_in + i * H + hh0);
            float4 o;
            o.x = acc[4 * g + 0] * (1.0f - __expf(-hv.x));
            o.y = acc[4 * g + 1] * (1.0f - __expf(-hv.y));
            o.z = acc[4 * g + 2] * (1.0f - __expf(-hv.z));
            o.w = acc[4 * g + 3] * (1.0f - __expf(-hv.w));
            *(float4*)(ghpre_out + i * H + hh0) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------
#define BWF_RT 5                        // row tiles (of 32) per wave -> 768 head rows per workgroup
#define BWF_ROWS (4 * BWF_RT * 32)

__host__ __device__ inline size_t enc_bwdw_fast_lds_floats(int D) { return 2 * (size_t)ENC_P * (D + 4) + (size_t)ENC_P * (D + 1) + ENC_P * 64; }

// slab layout (one per person range): PACKED = false: [W21: D*H | b21: D | W22: T*H | b22: T] (reference rows);
//                                       PACKED = true : [Wp-grad: Rp*H | bp-grad: Rp] (packed rows, k_pack.hip)
template <bool PACKED>
__global__ __launch_bounds__(ENC_THREADS, 1) void k_mvn_enc_bwd_w_fast(
    EncDims dm, float scale, const float* __restrict__ h_in, const float* __restrict__ eps_in,
    const float* __restrict__ ldT, const float* __restrict__ gx_in, const uint32_t* __restrict__ gtab,
    float* __restrict__ slabs, int64_t slab_len) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int H = 64;
    const int D = dm.D, T = dm.T;
    const int DP = D + 4, C4 = D / 4;
    float* gx_lds = smem;                   // [P][D+4]  slot D = 0
    float* eps_lds = gx_lds + ENC_P * DP;   // [P][D+4]  slot D = 1
    float* ld_lds = eps_lds + ENC_P * DP;   // [D + 1][P]  (dimension-major, as ldT); row D holds ones
    float* h_lds = ld_lds + ENC_P * (D + 1);   // [P][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int64_t RT = (int64_t)T + D;
    const int64_t rbase = (int64_t)blockIdx.x * BWF_ROWS + (int64_t)wave * BWF_RT * 32;
    // per-lane row description (enc_row_code_fast):  V = gx[kx] * eps[lx] * ld[ldrow] + addv * valid
    // (ldrow = D, the row of ones, and addv = 0 for every non-diagonal row) -- no branch in the hot loop
    int koff[BWF_RT], loff[BWF_RT], ldoff[BWF_RT];
    float addv[BWF_RT];
#pragma unroll
    for (int t = 0; t < BWF_RT; ++t) {
        uint32_t cc;
        if (PACKED) {
            const int64_t pr = rbase + 32 * t + l31;
            cc = ((uint32_t)D << 16) | (uint32_t)D;                              // padding: gx slot D = 0
            if (pr < pk_rows(D)) {
                const uint32_t code = gtab[pr >> 3];
                const uint32_t type = code >> 28, k = (code >> 12) & 0xFFFFu, l0 = code & 0xFFFu;
                const uint32_t jx = (uint32_t)(pr & 7);
                if (type == PK_OFF) { if (l0 + jx < k) cc = (k << 16) | (l0 + jx); }
                else if (type == PK_DIAG) { if (k + jx < (uint32_t)D) cc = FC_DIAG | ((k + jx) << 16) | (k + jx); }
                else if (type == PK_LOC) { if (k + jx < (uint32_t)D) cc = ((k + jx) << 16) | (uint32_t)D; }
            }
        } else {
            cc = enc_row_code_fast(rbase + 32 * t + l31, T, D);
        }
        koff[t] = (int)((cc >> 16) & 0x7FFFu);
        loff[t] = (int)(cc & 0xFFFFu);
        const bool isd = (cc & FC_DIAG) != 0;
        ldoff[t] = (isd ? loff[t] : D) * ENC_P;
        addv[t] = isd ? scale : 0.f;
    }
    f32x16 acc[BWF_RT][2];
    float bsum[BWF_RT];
#pragma unroll
    for (int t = 0; t < BWF_RT; ++t) { bsum[t] = 0.f; acc[t][0] = zero16(); acc[t][1] = zero16(); }

    const int n4d = ENC_P * D / 4;          // float4 count of one [P][D] tile
    constexpr int NQ = 7;                   // ceil(64*127/4 / 256) = 8 would be the bound for D = 127; D%4==0 -> <= 124
    float4 pg[NQ + 1], pe[NQ + 1], pl[NQ + 1], ph[4];
    const int64_t n_ptiles = (dm.nb + ENC_P - 1) / ENC_P;
    auto prefetch = [&](int64_t tile) {
        const int64_t i0 = tile * ENC_P;
        const int pv = (int)((dm.nb - i0) < ENC_P ? (dm.nb - i0) : ENC_P);
        const float4* g4 = (const float4*)(gx_in + i0 * D);
        const float4* e4 = (const float4*)(eps_in + i0 * D);
#pragma unroll
        for (int q = 0; q <= NQ; ++q) {
            const int idx = tid + ENC_THREADS * q;
            const bool ok = idx < pv * D / 4;
            pg[q] = ok ? g4[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
            pe[q] = ok ? e4[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
            // ld tile [D][P]: float4 idx -> k = idx / 16, persons 4*(idx%16)..+3
            const int k = idx >> 4, p4 = 4 * (idx & 15);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < n4d) {
                const float* src = ldT + (int64_t)k * dm.nb + i0 + p4;
                if (p4 + 3 < pv && ((reinterpret_cast<uintptr_t>(src) & 15u) == 0)) v = *(const float4*)src;
                else {
                    if (p4 + 0 < pv) v.x = src[0];
                    if (p4 + 1 < pv) v.y = src[1];
                    if (p4 + 2 < pv) v.z = src[2];
                    if (p4 + 3 < pv) v.w = src[3];
                }
            }
            pl[q] = v;
        }
        const float4* h4 = (const float4*)(h_in + i0 * H);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = tid + ENC_THREADS * q;
            ph[q] = (idx < pv * H / 4) ? h4[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    int64_t tile = blockIdx.y;
    if (tile < n_ptiles) prefetch(tile);
    for (; tile < n_ptiles; tile += gridDim.y) {
        const int64_t i0 = tile * ENC_P;
        const int pvalid = (int)((dm.nb - i0) < ENC_P ? (dm.nb - i0) : ENC_P);
        __syncthreads();
#pragma unroll
        for (int q = 0; q <= NQ; ++q) {
            const int idx = tid + ENC_THREADS * q;
            if (idx < n4d) {
                const int pp = idx / C4, c = idx - pp * C4;
                ((float4*)gx_lds)[pp * (C4 + 1) + c] = pg[q];
                ((float4*)eps_lds)[pp * (C4 + 1) + c] = pe[q];
                ((float4*)ld_lds)[idx] = pl[q];
            }
        }
        if (tid < ENC_P) {
            // gx slot D = 0 (padding rows), slot D+1 = 1 for persons inside the batch (gates the "+ scale")
            ((float4*)gx_lds)[tid * (C4 + 1) + C4] = make_float4(0.f, tid < pvalid ? 1.f : 0.f, 0.f, 0.f);
            ((float4*)eps_lds)[tid * (C4 + 1) + C4] = make_float4(1.f, 0.f, 0.f, 0.f);
            ld_lds[D * ENC_P + tid] = 1.0f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) ((float4*)h_lds)[tid + ENC_THREADS * q] = ph[q];
        __syncthreads();
        if (tile + gridDim.y < n_ptiles) prefetch(tile + gridDim.y);
        // operands of k-step s+1 are read from LDS while the MFMAs of step s run (one step of software
        // pipelining: with one wave per SIMD nothing else hides the LDS latency)
        float g0[BWF_RT], e0[BWF_RT], d0[BWF_RT], g1[BWF_RT], e1[BWF_RT], d1[BWF_RT];
        float hb0[2], hb1[2], vf0, vf1;
        auto fetch = [&](float (&g)[BWF_RT], float (&e)[BWF_RT], float (&d)[BWF_RT], float (&hb)[2], float& vf, int s) {
            const int p = 2 * s + half;
            const float* gx_p = gx_lds + p * DP;
            const float* eps_p = eps_lds + p * DP;
            hb[0] = h_lds[p * H + l31];
            hb[1] = h_lds[p * H + 32 + l31];
            vf = gx_p[D + 1];
#pragma unroll
            for (int t = 0; t < BWF_RT; ++t) { g[t] = gx_p[koff[t]]; e[t] = eps_p[loff[t]]; d[t] = ld_lds[ldoff[t] + p]; }
        };
        auto mma = [&](const float (&g)[BWF_RT], const float (&e)[BWF_RT], const float (&d)[BWF_RT],
                       const float (&hb)[2], float vf) {
#pragma unroll
            for (int t = 0; t < BWF_RT; ++t) {
                const float v = fmaf(g[t] * e[t], d[t], addv[t] * vf);
                bsum[t] += v;
                acc[t][0] = mfma32(v, hb[0], acc[t][0]);
                acc[t][1] = mfma32(v, hb[1], acc[t][1]);
            }
        };
        fetch(g0, e0, d0, hb0, vf0, 0);
        for (int s = 0; s < ENC_P / 2; s += 2) {
            fetch(g1, e1, d1, hb1, vf1, s + 1);
            mma(g0, e0, d0, hb0, vf0);
            if (s + 2 < ENC_P / 2) fetch(g0, e0, d0, hb0, vf0, s + 2);
            mma(g1, e1, d1, hb1, vf1);
        }
    }
    float* slab = slabs + (int64_t)blockIdx.y * slab_len;
    float* sW21 = slab;
    float* sb21 = sW21 + (int64_t)D * H;
    float* sW22 = sb21 + D;
    float* sb22 = sW22 + (int64_t)T * H;
    const int64_t Rp = pk_rows(D);
#pragma unroll
    for (int t = 0; t < BWF_RT; ++t) {
#pragma unroll
        for (int ht = 0; ht < 2; ++ht) {
            const int hh = 32 * ht + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = rbase + 32 * t + crow32(r, half);
                if (PACKED) {
                    if (row < Rp) slab[row * H + hh] = acc[t][ht][r];
                } else {
                    if (row < T) sW22[row * H + hh] = acc[t][ht][r];
                    else if (row < RT) sW21[(row - T) * H + hh] = acc[t][ht][r];
                }
            }
        }
        const float bt = bsum[t] + __shfl_xor(bsum[t], 32, 64);
        const int64_t row = rbase + 32 * t + l31;
        if (half == 0) {
            if (PACKED) {
                if (row < Rp) slab[Rp * H + row] = bt;
            } else {
                if (row < T) sb22[row] = bt;
                else if (row < RT) sb21[row - T] = bt;
            }
        }
    }
}
