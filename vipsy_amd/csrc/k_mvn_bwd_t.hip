// Weight gradients of the amortized MVN guide heads from DIMENSION-MAJOR operands (packed head layout, H == 64):
//
//     gWp[r][hh] = sum_p V[r][p] h[p][hh],    gbp[r] = sum_p V[r][p]
//     V[r][p]    = gx[p][k_r] * eps[p][l_r]                         OFF rows (k_r, l_r)     -- 97 % of the rows
//                = gx[p][k]                                         LOC rows  (the loc head, vi.py:450)
//                = gx[p][k] eps[p][k] exp(M_kk)[p] + scale          DIAG rows (vi.py:686 + the entropy term)
//
// Output-stationary: a wave owns BT_RT row tiles (accumulators resident), the workgroup streams 32-person tiles.
// Every operand is read from LDS along the CONTRACTION (person) axis with ds_read_b128, which needs the operands
// person-contiguous: gxT [D][nb], epsT [D][nb], hT [64][nb], ldT [D][nb] -- written that way by their producers.
// A tile row is 32 persons = 128 bytes = 8 chunks of 16 bytes, stored UNPADDED with the chunk index XOR-swizzled by
// (row & 7); the swizzle costs nothing because the global->LDS DMA takes an arbitrary global address per lane.
// Per 32 MFMAs: 10 ds_read_b128, 16 multiplies and the bias adds -- VALU issue time adds to fp32-MFMA time on
// gfx950 (tools/overlap_ubench.hip), so the instruction count per MFMA is the quantity to minimise.
#pragma once
#include "k_pack.hip"
#include <type_traits>

#define BT_P 32
#define BT_RT 4
#define BT_THREADS 256
#define BT_ROWS (4 * BT_RT * 32)          // packed rows per workgroup

// LDS row map of one buffer.  A tile row is 32 persons = 128 bytes = 8 chunks of 16 bytes.  ds_read_b128 banks are
// (addr / 4) % 64 -- a 256-byte bank row of 16 chunk slots -- and it is serviced in four 16-lane groups, so the 16
// rows a group reads (chunk c of each) must land on 16 different slots.  Row R, chunk c lives at
//     bank row (R >> 4) * 8 + (R & 7),   half (R >> 3) & 1,   slot within the half  c ^ (R & 7)
// i.e. rows R and R + 8 share a bank row.  The placement costs nothing: the global->LDS DMA takes a per-lane
// global address, so each lane simply fetches the chunk that belongs at its (fixed) LDS position.
// Regions start at multiples of 16 rows (DR = D rounded up to 16):
//     G (gxT) [0, DR) | E (epsT) [DR, 2 DR) | H (hT) [2 DR, 2 DR + 64) | GD (gdT) [.., + DR) | C (row 0: ones, row 1: zeros)
// gdT[k][i] = gxT * epsT * exp(M_kk) + scale is the DIAG-row operand, made by k_mvn_gd below, so that every row of
// every section is V = (G or GD row) * (E row or ones): one code path, no per-section branch in the hot loop.
__host__ __device__ inline int bt_dr(int D) { return (D + 15) & ~15; }
__host__ __device__ inline int bt_row_E(int D) { return bt_dr(D); }
__host__ __device__ inline int bt_row_H(int D) { return 2 * bt_dr(D); }
__host__ __device__ inline int bt_row_GD(int D) { return 2 * bt_dr(D) + 64; }
__host__ __device__ inline int bt_row_ones(int D) { return 3 * bt_dr(D) + 64; }
__host__ __device__ inline int bt_row_zero(int D) { return 3 * bt_dr(D) + 65; }
__host__ __device__ inline int bt_rows(int D) { return 3 * bt_dr(D) + 80; }
__host__ __device__ inline uint32_t bt_addr(int R, int c) {
    return (uint32_t)(((R >> 4) * 8 + (R & 7)) * 256 + (((R >> 3) & 1) << 7) + ((c ^ (R & 7)) << 4));
}
// buffer stride: a compile-time constant (D <= 128), so that the second buffer is an immediate offset of ds_read
#define BT_BUF 59392
__host__ __device__ inline size_t bt_lds_bytes(int D) { return (size_t)BT_BUF + (size_t)bt_rows(D) * 128; }

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

__device__ __forceinline__ void bwd_w_t_body(
    const EncDims& dm, const float* __restrict__ hT, const float* __restrict__ epsT, const float* __restrict__ gdT,
    const float* __restrict__ gxT, const uint32_t* __restrict__ gtab, float* __restrict__ slabs, int64_t slab_len,
    char* smem) {
    const int D = dm.D;
    const int64_t nb = dm.nb;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    constexpr uint32_t BUF = BT_BUF;
    const int Rp = pk_rows(D);
    const int64_t rbase = (int64_t)blockIdx.x * BT_ROWS + (int64_t)wave * BT_RT * 32;
    const int rE = bt_row_E(D), rH = bt_row_H(D), rGD = bt_row_GD(D), rOnes = bt_row_ones(D), rZero = bt_row_zero(D);
    // workgroups whose rows reach into the DIAG section also stage the GD region (block-uniform)
    const bool need_gd = (int64_t)(blockIdx.x + 1) * BT_ROWS > pk_off_total(D);

    // ---- per-lane row description -> LDS byte addresses of the 4 chunk pairs (q = 0..3) of each operand row
    auto chunk_addr = [&](int row, int q) -> uint32_t { return bt_addr(row, 2 * q + half); };   // chunk 2q + half
    uint32_t aG[BT_RT][4], aE[BT_RT][4], aH[2][4];
#pragma unroll
    for (int t = 0; t < BT_RT; ++t) {
        const int64_t pr = rbase + 32 * t + l31;
        int g = rZero, e = rOnes;
        if (pr < Rp) {
            const uint32_t code = gtab[pr >> 3];
            const uint32_t type = code >> 28, k = (code >> 12) & 0xFFFFu, l0 = code & 0xFFFu, jx = (uint32_t)(pr & 7);
            if (type == PK_OFF) { if (l0 + jx < k) { g = (int)k; e = rE + (int)(l0 + jx); } }
            else if (type == PK_LOC) { if (k + jx < (uint32_t)D) g = (int)(k + jx); }
            else if (type == PK_DIAG) { if (k + jx < (uint32_t)D) g = rGD + (int)(k + jx); }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { aG[t][q] = chunk_addr(g, q); aE[t][q] = chunk_addr(e, q); }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { aH[0][q] = chunk_addr(rH + l31, q); aH[1][q] = chunk_addr(rH + 32 + l31, q); }

    // ---- constant rows of both buffers; everything else is (re)written by the DMA of each tile
    for (int b = 0; b < 2; ++b) {
        if (tid < 64) *(float*)(smem + b * BUF + bt_addr(rOnes + (tid >> 5), (tid & 31) >> 2) + 4 * (tid & 3)) = (tid < 32) ? 1.0f : 0.f;
    }
    f32x16 acc[BT_RT][2];
    float bsum[BT_RT];
#pragma unroll
    for (int t = 0; t < BT_RT; ++t) { bsum[t] = 0.f; acc[t][0] = zero16(); acc[t][1] = zero16(); }

    const int64_t n_ptiles = (nb + BT_P - 1) / BT_P;
    // One DMA instruction fills 1 KB = four bank rows = the rows {16 P + 8 beta + i : beta = 0, 1; i = 4 j .. 4 j + 3}
    // of one region; LDS position of lane: bank row lane >> 4, half (lane >> 3) & 1, slot lane & 7.  Wave w issues
    // the transfers d = w, w + 4, ... < n_dma; the per-lane global address of a transfer (row and chunk) never
    // changes, only the person offset of the tile is added -- no scalar control flow per transfer.
    constexpr int BT_MAXD = 14;                                        // (3 * 128 + 64) / 8 / 4 transfers per wave
    const int n_dma = (need_gd ? rOnes : rGD) / 8;
    const float* gptr[BT_MAXD];
#pragma unroll
    for (int u = 0; u < BT_MAXD; ++u) {
        const int d = wave + 4 * u;
        const int i = 4 * (d & 1) + (lane >> 4), beta = (lane >> 3) & 1;
        const int R = 16 * (d >> 1) + 8 * beta + i;
        const int c = (lane & 7) ^ i;
        const float* rb = gxT;
        int rl = R, rows_in = D;
        if (R >= rGD) { rb = gdT; rl = R - rGD; }
        else if (R >= rH) { rb = hT; rl = R - rH; rows_in = 64; }
        else if (R >= rE) { rb = epsT; rl = R - rE; }
        if (rl >= rows_in) rl = rows_in - 1;                           // padding rows of a region: a harmless duplicate
        gptr[u] = rb + (int64_t)rl * nb + 4 * c;
    }
    auto stage = [&](int64_t tile, int b) {
        const int64_t i0 = tile * BT_P;
        const int pv = (int)((nb - i0) < BT_P ? (nb - i0) : BT_P);
        const uint32_t lbase = lds_addr_uniform(smem + b * BUF) + (uint32_t)wave * 1024u;
        if (pv == BT_P) {                                              // all 64 lanes: a partial EXEC makes the DMA slow
#pragma unroll
            for (int u = 0; u < BT_MAXD; ++u)
                if (wave + 4 * u < n_dma) dma16(gptr[u] + i0, lbase + (uint32_t)u * 4096u);
        } else {                                                       // the last tile: absent persons are zeros
            for (int e = tid; e < bt_rows(D) * 32; e += BT_THREADS) {
                const int row = e >> 5;
                if (row != rOnes) *(float*)(smem + b * BUF + bt_addr(row, (e & 31) >> 2) + 4 * (e & 3)) = 0.f;
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < BT_MAXD; ++u) {
                const int d = wave + 4 * u;
                if (d < n_dma && 4 * ((lane & 7) ^ (4 * (d & 1) + (lane >> 4))) < pv)
                    dma16(gptr[u] + i0, lbase + (uint32_t)u * 4096u);
            }
        }
    };

    // 4 * BT_RT steps per tile (q = chunk pair, t = row tile), 8 MFMAs each; the operands of step s + 1 are read while the
    // MFMAs of step s run (every step is a pinned scheduling region, as in k_irt_lik_r.hip)
    auto compute = [&](auto bc) {
        constexpr int b = decltype(bc)::value;
        const char* base = smem + b * BUF;
        f32x4 gc = *(const f32x4*)(base + aG[0][0]), ec = *(const f32x4*)(base + aE[0][0]);
        f32x4 h0 = *(const f32x4*)(base + aH[0][0]), h1 = *(const f32x4*)(base + aH[1][0]);
        f32x4 hn0 = h0, hn1 = h1;
        static_for<4 * BT_RT>([&](auto sc) {
            constexpr int s2 = decltype(sc)::value, t = s2 % BT_RT;
            constexpr int qn = (s2 + 1) / BT_RT, tn = (s2 + 1) % BT_RT;
            f32x4 gn = gc, en = ec;
            if constexpr (s2 + 1 < 4 * BT_RT) {
                gn = *(const f32x4*)(base + aG[tn][qn]);
                en = *(const f32x4*)(base + aE[tn][qn]);
                if constexpr (t == BT_RT - 1) {
                    hn0 = *(const f32x4*)(base + aH[0][qn]);
                    hn1 = *(const f32x4*)(base + aH[1][qn]);
                }
            }
            const f32x4 v = gc * ec;
            bsum[t] += (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[t][0] = mfma32(v[i], h0[i], acc[t][0]);
                acc[t][1] = mfma32(v[i], h1[i], acc[t][1]);
            }
            gc = gn; ec = en;
            if constexpr (t == BT_RT - 1) { h0 = hn0; h1 = hn1; }
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    int64_t tile = blockIdx.y;
    if (tile < n_ptiles) stage(tile, 0);
    // two tiles per trip so that the buffer index is a compile-time constant (LDS offsets become immediates)
    while (tile < n_ptiles) {
        vx_wait_vmem();
        __syncthreads();                                               // tile in buffer 0 landed; buffer 1 free
        int64_t nx = tile + gridDim.y;
        if (nx < n_ptiles) stage(nx, 1);
        compute(std::integral_constant<int, 0>{});
        tile = nx;
        if (tile >= n_ptiles) break;
        vx_wait_vmem();
        __syncthreads();                                               // tile in buffer 1 landed; buffer 0 free
        nx = tile + gridDim.y;
        if (nx < n_ptiles) stage(nx, 0);
        compute(std::integral_constant<int, 1>{});
        tile = nx;
    }

    // ---- slab of this person range: [Wp-grad: Rp*H | bp-grad: Rp]
    float* slab = slabs + (int64_t)blockIdx.y * slab_len;
#pragma unroll
    for (int t = 0; t < BT_RT; ++t) {
#pragma unroll
        for (int ht = 0; ht < 2; ++ht) {
            const int hh = 32 * ht + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = rbase + 32 * t + crow32(r, half);
                if (row < Rp) slab[row * 64 + hh] = acc[t][ht][r];
            }
        }
        const float bt = half_sum32(bsum[t]);
        const int64_t row = rbase + 32 * t + l31;
        if (half == 0 && row < Rp) slab[(int64_t)Rp * 64 + row] = bt;
    }
}

__global__ __launch_bounds__(BT_THREADS, 1) void k_mvn_enc_bwd_w_t(
    EncDims dm, const float* __restrict__ hT, const float* __restrict__ epsT, const float* __restrict__ gdT,
    const float* __restrict__ gxT, const uint32_t* __restrict__ gtab, float* __restrict__ slabs, int64_t slab_len) {
    extern __shared__ __attribute__((aligned(16))) char smem_bt[];
    bwd_w_t_body(dm, hT, epsT, gdT, gxT, gtab, slabs, slab_len, smem_bt);
}

// gdT[k][i] = gxT[k][i] * epsT[k][i] * exp(M_kk)[i] + scale: the DIAG-row operand (vi.py:686 and the entropy term)
__global__ void k_mvn_gd(const float4* __restrict__ gxT, const float4* __restrict__ epsT, const float4* __restrict__ ldT,
                         float scale, int64_t n4, float4* __restrict__ gdT) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 g = gxT[i], e = epsT[i], l = ldT[i];
        gdT[i] = make_float4(fmaf(g.x * e.x, l.x, scale), fmaf(g.y * e.y, l.y, scale), fmaf(g.z * e.z, l.z, scale),
                             fmaf(g.w * e.w, l.w, scale));
    }
}

// ------------------------------------------------------------------------------------------------------------
// out[c][r] = in[r][c]  (in: [R][C] row-major); 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void k_transpose(const float* __restrict__ in, float* __restrict__ out, int64_t R,
                                                   int64_t C) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;           // 32 x 8
    const int64_t ntc = (C + 31) / 32, ntr = (R + 31) / 32;
    for (int64_t t = blockIdx.x; t < ntc * ntr; t += gridDim.x) {
        const int64_t r0 = (t / ntc) * 32, c0 = (t % ntc) * 32;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t r = r0 + ty + 8 * k, c = c0 + tx;
            tile[ty + 8 * k][tx] = (r < R && c < C) ? in[r * C + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t c = c0 + ty + 8 * k, r = r0 + tx;
            if (r < R && c < C) out[c * R + r] = tile[tx][ty + 8 * k];
        }
        __syncthreads();
    }
}
