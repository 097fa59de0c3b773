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

// workgroup -> (row block rb, person range pr).  The grid is (row blocks, person ranges); the hardware deals the linear
// workgroup ids round-robin over the 8 XCDs, and the row blocks of one person range all read the same tiles.  They
// are placed on consecutive slots of ONE XCD (a group straddles two XCDs at most), so that the re-reads are served by
// that XCD's L2 instead of crossing the fabric once per XCD.
// (the grid may be a VIRTUAL one: k_bwd_wt_fc1 gives this body the first gx * gy workgroups of a launch it shares; `id` is
// the workgroup's linear index in it)
struct VGrid { int id, gx, gy; };
__device__ __forceinline__ VGrid vgrid_launch() { return VGrid{(int)(blockIdx.x + gridDim.x * blockIdx.y), (int)gridDim.x, (int)gridDim.y}; }
__device__ __forceinline__ void bt_decode(const VGrid& vg, int& rb, int& pr) {
    const int n_rb = vg.gx, nblk = vg.gx * vg.gy;
    const int id = vg.id, x = id & 7;
    int p = id >> 3;                                                   // slot within the XCD
    for (int xx = 0; xx < x; ++xx) p += (nblk - xx + 7) >> 3;          // + the slots of the XCDs before it
    pr = p / n_rb;
    rb = p - pr * n_rb;
}

__device__ __forceinline__ void bwd_w_t_body(
    const EncDims& dm, const float* __restrict__ hT, const float* __restrict__ epsT, const float* __restrict__ gdT,
    const float* __restrict__ gxT, const uint32_t* __restrict__ gtab, float* __restrict__ slabs, int64_t slab_len,
    char* smem, const VGrid vg) {
    const int D = dm.D;
    const int64_t nb = dm.nb;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    constexpr uint32_t BUF = BT_BUF;
    const int Rp = pk_rows(D);
    int rb_, pr_;
    bt_decode(vg, rb_, pr_);
    const int64_t rbase = (int64_t)rb_ * BT_ROWS + (int64_t)wave * BT_RT * 32;
    const int rE = bt_row_E(D), rH = bt_row_H(D), rGD = bt_row_GD(D), rOnes = bt_row_ones(D), rZero = bt_row_zero(D);
    // workgroups whose rows reach into the DIAG section also stage the GD region (block-uniform)
    const bool need_gd = (int64_t)(rb_ + 1) * BT_ROWS > pk_off_total(D);

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
            const f32x4 v = gc * ec;
            bsum[t] += (v[0] + v[1]) + (v[2] + v[3]);
            // LDS instructions issue beside a running MFMA, VALU ones do not: next step's reads follow the first MFMA
            acc[t][0] = mfma32(v[0], h0[0], acc[t][0]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (s2 + 1 < 4 * BT_RT) {
                gn = *(const f32x4*)(base + aG[tn][qn]);
                en = *(const f32x4*)(base + aE[tn][qn]);
                if constexpr (t == BT_RT - 1) {
                    hn0 = *(const f32x4*)(base + aH[0][qn]);
                    hn1 = *(const f32x4*)(base + aH[1][qn]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[t][1] = mfma32(v[0], h1[0], acc[t][1]);
#pragma unroll
            for (int i = 1; i < 4; ++i) {
                acc[t][0] = mfma32(v[i], h0[i], acc[t][0]);
                acc[t][1] = mfma32(v[i], h1[i], acc[t][1]);
            }
            gc = gn; ec = en;
            if constexpr (t == BT_RT - 1) { h0 = hn0; h1 = hn1; }
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    int64_t tile = pr_;
    if (tile < n_ptiles) stage(tile, 0);
    // two tiles per trip so that the buffer index is a compile-time constant (LDS offsets become immediates)
    while (tile < n_ptiles) {
        vx_wait_vmem();
        __syncthreads();                                               // tile in buffer 0 landed; buffer 1 free
        int64_t nx = tile + vg.gy;
        if (nx < n_ptiles) stage(nx, 1);
        compute(std::integral_constant<int, 0>{});
        tile = nx;
        if (tile >= n_ptiles) break;
        vx_wait_vmem();
        __syncthreads();                                               // tile in buffer 1 landed; buffer 0 free
        nx = tile + vg.gy;
        if (nx < n_ptiles) stage(nx, 0);
        compute(std::integral_constant<int, 1>{});
        tile = nx;
    }

    // ---- slab of this person range: [Wp-grad: Rp*H | bp-grad: Rp]
    float* slab = slabs + (int64_t)pr_ * slab_len;
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
    bwd_w_t_body(dm, hT, epsT, gdT, gxT, gtab, slabs, slab_len, smem_bt, vgrid_launch());
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

// ------------------------------------------------------------------------------------------------------------
// Hidden-layer gradient of the amortized MVN guide, person-stationary:
//     gh^T[hh][p] = sum_r WpT[hh][r] V[r][p],       ghpre[p][hh] = gh * (1 - exp(-h))        (softplus', vi.py:449)
// A workgroup keeps 128 persons (4 waves x 32) resident in LDS -- gxT [D][128] and eps [128][ES] -- and streams the
// packed head weights in 64-row tiles (transposed copy WpT [64][Rp] from k_pack_heads; DMA, double buffered, each
// 256-byte LDS row = hidden unit hh holds the tile's 16 chunks at slot c ^ (hh & 15)).  Per 8 MFMAs (4 packed rows
// x both hidden tiles): two 16-byte weight reads, one 16-byte eps read, one 4-byte gx read and 4 multiplies.
#define BH_P 128
#define BH_THREADS 256
#define BH_TR 64
#define BH_WBUF (64 * BH_TR * 4)                                       // bytes of one weight tile

__host__ __device__ inline int bh_es(int D) { int c = (D + 3) / 4 + 2; if ((c & 1) == 0) ++c; return 4 * c; }   // chunks odd
#define BH_NBUF 3                                                      // weight tiles in flight: prefetch distance 2
__host__ __device__ inline size_t bh_lds_bytes(int D) {
    return BH_NBUF * (size_t)BH_WBUF + (size_t)D * BH_P * 4 + (size_t)BH_P * bh_es(D) * 4 + (size_t)pk_rows(D) / 8 * 4;
}

__global__ __launch_bounds__(BH_THREADS, 1) void k_mvn_enc_bwd_h_t(
    EncDims dm, float scale, const float* __restrict__ WpT, const uint32_t* __restrict__ gtab,
    const float* __restrict__ h_in, const float* __restrict__ eps_in, const float* __restrict__ ldT,
    const float* __restrict__ gxT, const float* __restrict__ gdT /*DIAG-row operand [D][nb]*/,
    float* __restrict__ ghpre_out /*[nb][64] or null*/,
    const float* __restrict__ hT /*[64][nb], with ghpreT_out*/, float* __restrict__ ghpreT_out /*[64][nb] or null*/) {
    extern __shared__ __attribute__((aligned(16))) char smem_bh[];
    constexpr int H = 64;
    const int D = dm.D, ES = bh_es(D);
    const int64_t nb = dm.nb;
    char* Wt = smem_bh;                                                // [3][64 hh][64 r]
    float* gx_lds = (float*)(smem_bh + BH_NBUF * BH_WBUF);             // [D][128]
    float* eps_lds = gx_lds + D * BH_P;                                // [128][ES]
    uint32_t* gt_lds = (uint32_t*)(eps_lds + BH_P * ES);               // [Rp / 8] group codes
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int64_t i0 = (int64_t)blockIdx.x * BH_P;
    const int Rp = pk_rows(D), n_tiles = Rp / BH_TR, n_off = pk_off_total(D) / BH_TR;
    const int offT = pk_off_total(D), sec = pk_sec(D);

    // ---- resident person data (once per workgroup).  Persons past the batch read the last valid person: finite
    // values whose results are never stored.
    {
        const int c128 = BH_P / 4;                                     // gxT rows: 32 chunks; one DMA = 2 rows
        for (int j = wave; 2 * j < D; j += 4) {
            int k = 2 * j + (lane >> 5);
            if (k >= D) k = D - 1;
            int64_t pp = i0 + 4 * (lane & 31);
            if (pp + 4 > nb) pp = nb - 4;                              // nb % 4 == 0, nb >= 4
            dma16(gxT + (int64_t)k * nb + pp, lds_addr_uniform(gx_lds + 2 * j * BH_P));
        }
        (void)c128;
        const int ec = ES / 4, n_chunks = BH_P * ec;                   // eps tile as one contiguous chunk array
        for (int j = wave; 64 * j < n_chunks; j += 4) {
            const int ch = 64 * j + lane;
            int pr = ch / ec, cc = ch - pr * ec;
            if (pr >= BH_P) { pr = BH_P - 1; cc = 0; }
            if (4 * cc >= D) cc = 0;                                   // pad chunks: any finite values
            int64_t ii = i0 + pr;
            if (ii >= nb) ii = nb - 1;
            dma16(eps_in + ii * D + 4 * cc, lds_addr_uniform((char*)eps_lds + (size_t)j * 1024));
        }
    }
    // ---- weight tiles: one DMA = 4 hidden rows x 16 chunks; wave w moves rows 16 w .. 16 w + 15
    auto stage_w = [&](int tile, int b) {
        const uint32_t lbase = lds_addr_uniform(Wt + b * BH_WBUF) + (uint32_t)wave * 4096u;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int hh = 16 * wave + 4 * u + (lane >> 4);
            const int c = (lane & 15) ^ (hh & 15);
            dma16(WpT + (int64_t)hh * Rp + (int64_t)tile * BH_TR + 4 * c, lbase + (uint32_t)u * 1024u);
        }
    };
    stage_w(0, 0);
    if (n_tiles > 1) stage_w(1, 1);
    for (int e = tid; e < Rp / 8; e += BH_THREADS) gt_lds[e] = gtab[e];

    const int p = 32 * wave + l31;                                     // this lane's person (B / C column)
    const int64_t i = i0 + p;
    const float* gx_p = gx_lds + p;                                    // + k * BH_P
    const float* eps_p = eps_lds + p * ES;
    uint32_t aW[2][8];                                                 // weight chunk addresses of this lane (per q')
#pragma unroll
    for (int ht = 0; ht < 2; ++ht)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int hh = 32 * ht + l31;
            aW[ht][q] = (uint32_t)(hh * 256 + (((2 * q + half) ^ (hh & 15)) << 4));
        }
    f32x16 acc0 = zero16(), acc1 = zero16();

    // OFF tile: 8 eight-row groups (q), 8 MFMAs each.  The operands of group q + 1 are read while the MFMAs of group q
    // run (pinned scheduling regions); the group codes (k, l0) come from the LDS copy of gtab (a uniform address).
    auto off_tile = [&](auto bc, int tile) {
        constexpr int b = decltype(bc)::value;
        const char* wb = Wt + b * BH_WBUF;
        const uint4 gcA = *(const uint4*)(gt_lds + tile * (BH_TR / 8)), gcB = *(const uint4*)(gt_lds + tile * (BH_TR / 8) + 4);
        const uint32_t gt[8] = {gcA.x, gcA.y, gcA.z, gcA.w, gcB.x, gcB.y, gcB.z, gcB.w};
        uint32_t code[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) code[q] = __builtin_amdgcn_readfirstlane(gt[q]);
        auto ld_g = [&](uint32_t c) { return gx_p[(int)((c >> 12) & 0xFFFFu) * BH_P]; };
        auto ld_e = [&](uint32_t c) { return *(const f32x4*)(eps_p + (int)(c & 0xFFFu) + 4 * half); };
        float gk = ld_g(code[0]);
        f32x4 e4 = ld_e(code[0]);
        f32x4 w0 = *(const f32x4*)(wb + aW[0][0]), w1 = *(const f32x4*)(wb + aW[1][0]);
        static_for<8>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            float gn = gk;
            f32x4 en = e4, wn0 = w0, wn1 = w1;
            const f32x4 v = gk * e4;
            // LDS instructions issue beside a running MFMA (VALU ones do not): the reads of group q + 1 go right
            // after the first MFMA of group q -- early enough to land before they are needed, not ahead of the chain
            acc0 = mfma32(w0[0], v[0], acc0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (q + 1 < 8) {
                gn = ld_g(code[q + 1]);
                en = ld_e(code[q + 1]);
                wn0 = *(const f32x4*)(wb + aW[0][q + 1]);
                wn1 = *(const f32x4*)(wb + aW[1][q + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc1 = mfma32(w1[0], v[0], acc1);
#pragma unroll
            for (int i2 = 1; i2 < 4; ++i2) {
                acc0 = mfma32(w0[i2], v[i2], acc0);
                acc1 = mfma32(w1[i2], v[i2], acc1);
            }
            gk = gn; e4 = en; w0 = wn0; w1 = wn1;
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    // DIAG / LOC sections (a handful of tiles): every eight-row group has one type; buffer index at run time.
    // V = gd[p][k] (DIAG, made by k_mvn_gd) or gx[p][k] (LOC); the gdT tile is DMA'd over the eps region once the OFF
    // section is done, so these tiles read LDS only.
    const float* gd_p = eps_lds + p;                                   // + k * BH_P  (after stage_gd)
    auto stage_gd = [&]() {
        for (int j = wave; 2 * j < D; j += 4) {
            int k = 2 * j + (lane >> 5);
            if (k >= D) k = D - 1;
            int64_t pp = i0 + 4 * (lane & 31);
            if (pp + 4 > nb) pp = nb - 4;
            dma16(gdT + (int64_t)k * nb + pp, lds_addr_uniform(eps_lds + 2 * j * BH_P));
        }
    };
    auto tail_tile = [&](int b, int tile) {
        const char* wb = Wt + b * BH_WBUF;
#pragma unroll 2
        for (int q = 0; q < 8; ++q) {
            const int r0 = tile * BH_TR + 8 * q;                       // first packed row of this 8-group (uniform)
            const bool is_diag = r0 < offT + sec, is_loc = !is_diag && r0 < offT + 2 * sec;
            const int k0 = r0 - (is_diag ? offT : offT + sec) + 4 * half;
            const float* src = is_diag ? gd_p : gx_p;
            f32x4 v;
#pragma unroll
            for (int i2 = 0; i2 < 4; ++i2) {
                const int k = k0 + i2;
                const float t = src[(k < D ? k : D - 1) * BH_P];
                v[i2] = ((is_diag || is_loc) && k < D) ? t : 0.f;
            }
            const uint32_t hx = (uint32_t)(l31 & 15);                  // (hh & 15) of both hidden tiles
            const f32x4 w0 = *(const f32x4*)(wb + l31 * 256 + (((2 * q + half) ^ hx) << 4));
            const f32x4 w1 = *(const f32x4*)(wb + (32 + l31) * 256 + (((2 * q + half) ^ hx) << 4));
#pragma unroll
            for (int i2 = 0; i2 < 4; ++i2) {
                acc0 = mfma32(w0[i2], v[i2], acc0);
                acc1 = mfma32(w1[i2], v[i2], acc1);
            }
        }
    };

    // Weight tiles run three deep: before tile t is used only the transfers of tile t + 1 (this wave's 4 newest)
    // may still be in flight -- vmcnt(4); the transfer of tile t + 2 is issued right after the barrier, into the
    // buffer tile t - 1 has just left.  n_off is a multiple of 3 (the OFF section is padded to 192 rows).
    auto wait_tile = [&](int tile) {
        if (tile + 1 < n_tiles) __builtin_amdgcn_s_waitcnt(0x0F74);    // vmcnt(4)
        else vx_wait_vmem();
        __syncthreads();
        if (tile + 2 < n_tiles) stage_w(tile + 2, (tile + 2) % BH_NBUF);
    };
    int tile = 0;
    for (; tile + 2 < n_off; tile += 3) {
        wait_tile(tile);
        off_tile(std::integral_constant<int, 0>{}, tile);
        wait_tile(tile + 1);
        off_tile(std::integral_constant<int, 1>{}, tile + 1);
        wait_tile(tile + 2);
        off_tile(std::integral_constant<int, 2>{}, tile + 2);
    }
    __syncthreads();                                                   // every wave is done with eps
    stage_gd();
    for (; tile < n_tiles; ++tile) {
        if (tile == n_off) {                                           // first tail tile: also wait for the gd tile
            vx_wait_vmem();
            __syncthreads();
            if (tile + 2 < n_tiles) stage_w(tile + 2, (tile + 2) % BH_NBUF);
        } else {
            wait_tile(tile);
        }
        tail_tile(tile % BH_NBUF, tile);
    }
    // ---- ghpre = gh * softplus'(pre) = gh * (1 - exp(-h));  C layout: rows hh = crow32(r, half), cols p
    if (ghpreT_out) {                                                  // dimension-major: 128-byte rows per half-wave
        if (i < nb) {
#pragma unroll
            for (int ht = 0; ht < 2; ++ht)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t o = (int64_t)(32 * ht + crow32(r, half)) * nb + i;
                    ghpreT_out[o] = (ht ? acc1 : acc0)[r] * (1.0f - __expf(-hT[o]));
                }
        }
    } else if (i < nb) {
#pragma unroll
        for (int ht = 0; ht < 2; ++ht)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int hh0 = 32 * ht + 8 * g + 4 * half;
                const float4 hv = *(const float4*)(h_in + i * H + hh0);
                float4 o;
                o.x = (ht ? acc1 : acc0)[4 * g + 0] * (1.0f - __expf(-hv.x));
                o.y = (ht ? acc1 : acc0)[4 * g + 1] * (1.0f - __expf(-hv.y));
                o.z = (ht ? acc1 : acc0)[4 * g + 2] * (1.0f - __expf(-hv.z));
                o.w = (ht ? acc1 : acc0)[4 * g + 3] * (1.0f - __expf(-hv.w));
                *(float4*)(ghpre_out + i * H + hh0) = o;
            }
    }
}

// ------------------------------------------------------------------------------------------------------------
// fc1 weight gradient from dimension-major operands (full batch, no row gather):
//     GW1[hh][j] = sum_p ghpreT[hh][p] yin[p][j],   Gb1[hh] = sum_p ghpreT[hh][p]         (yin = int8 -1 / 0 / 1)
// yT [J][n] is the item-major copy of the response bytes (made once by the host; responses never change).
// A wave owns 4 item tiles x both hidden tiles (128 AGPRs); 32-person tiles by DMA, double buffered:
//   ghpreT rows in the bank-row-aware layout of bt_addr(); yT rows are 32 bytes = 2 chunks of 16 persons, chunk c of
//   items 16 B .. 16 B + 15 shares bank row 2 B + c (slot = item & 15): a 16-byte read hands a lane the bytes of 16
//   persons for its item.  Person order inside the MFMA steps: 16 c + 8 m + 4 half + i.
#define F1_P 32
#define F1_THREADS 256
#define F1_JW 128                                                      // items per wave (4 tiles)
#define F1_GBUF (64 * 128)                                             // bytes of the ghpreT tile
__host__ __device__ inline int f1_jrows(int J) { return (J + 31) & ~31; }
__host__ __device__ inline size_t f1_buf_bytes(int J) { return F1_GBUF + (size_t)f1_jrows(J) * 32; }
__host__ __device__ inline size_t f1_lds_bytes(int J) { return 2 * f1_buf_bytes(J > 512 ? 512 : J); }

__global__ __launch_bounds__(F1_THREADS, 1) void k_fc1_bwd_t(
    EncDims dm, const uint8_t* __restrict__ yT, int64_t ystride, const float* __restrict__ ghpreT,
    float* __restrict__ slabs, int64_t slab_len) {
    extern __shared__ __attribute__((aligned(16))) char smem_f1[];
    const int J = dm.J;
    const int64_t nb = dm.nb;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int jg0 = blockIdx.x * 512;                                  // items of this workgroup: jg0 .. jg0 + 511
    const int jn = (J - jg0) < 512 ? (J - jg0) : 512;
    const uint32_t BUF = (uint32_t)f1_buf_bytes(jn);
    const int n_ydma = f1_jrows(jn) / 32;                              // one transfer = 32 item rows x 2 chunks
    // per-lane LDS addresses
    uint32_t aA[2][4];                                                 // ghpreT chunk 4c + 2m + half of rows l31, 32 + l31
#pragma unroll
    for (int ht = 0; ht < 2; ++ht)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            aA[ht][u] = bt_addr(32 * ht + l31, 2 * u + half);
        }
    uint32_t aY[4][2];                                                 // yT chunk c of item 128 wave + 32 t + l31
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int jl = F1_JW * wave + 32 * t + l31;
            aY[t][c] = (uint32_t)(F1_GBUF + ((jl >> 4) * 2 + c) * 256 + (jl & 15) * 16);
        }
    // per-lane DMA sources (tile-invariant part)
    const float* gsrc[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {                                      // ghpreT: 8 transfers, wave w issues d = w, w + 4
        const int d = wave + 4 * u;
        const int i = 4 * (d & 1) + (lane >> 4), beta = (lane >> 3) & 1;
        const int R = 16 * (d >> 1) + 8 * beta + i, c = (lane & 7) ^ i;
        gsrc[u] = ghpreT + (int64_t)R * nb + 4 * c;
    }
    const uint8_t* ysrc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {                                      // yT: n_ydma transfers
        const int d = wave + 4 * u;
        int j = jg0 + 32 * d + 16 * (lane >> 5) + (lane & 15);
        if (j >= J) j = J - 1;
        ysrc[u] = yT + (int64_t)j * ystride + 16 * ((lane >> 4) & 1);
    }
    auto stage = [&](int64_t tile, int b) {
        const int64_t i0 = tile * F1_P;
        const int pv = (int)((nb - i0) < F1_P ? (nb - i0) : F1_P);
        const uint32_t lb = lds_addr_uniform(smem_f1 + b * BUF);
        if (pv < F1_P) {                                               // last tile: absent persons have ghpre = 0
            for (int e = tid; e < F1_GBUF / 4; e += F1_THREADS) ((float*)(smem_f1 + b * BUF))[e] = 0.f;
            __syncthreads();
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int d = wave + 4 * u;
            const int c = (lane & 7) ^ (4 * (d & 1) + (lane >> 4));
            if (pv == F1_P) dma16(gsrc[u] + i0, lb + (uint32_t)d * 1024u);
            else if (4 * c < pv) dma16(gsrc[u] + i0, lb + (uint32_t)d * 1024u);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int d = wave + 4 * u;
            if (d < n_ydma) dma16(ysrc[u] + i0, lb + F1_GBUF + (uint32_t)d * 1024u);
        }
    };
    f32x16 acc[4][2];
    float bsum[2] = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t) { acc[t][0] = zero16(); acc[t][1] = zero16(); }

    auto compute = [&](int b) {
        const char* base = smem_f1 + b * BUF;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            uint4 yc[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) yc[t] = *(const uint4*)(base + aY[t][c]);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const f32x4 g0 = *(const f32x4*)(base + aA[0][2 * c + m]);
                const f32x4 g1 = *(const f32x4*)(base + aA[1][2 * c + m]);
                bsum[0] += (g0[0] + g0[1]) + (g0[2] + g0[3]);
                bsum[1] += (g1[0] + g1[1]) + (g1[2] + g1[3]);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const uint32_t lo = m ? yc[t].z : yc[t].x, hi = m ? yc[t].w : yc[t].y;
                    const uint32_t wu = half ? hi : lo;                // bytes of persons 16c + 8m + 4half + 0..3
                    const int w = (int)(wu & ((wu & 0x01010101u) * 0xFFu));   // 254 (pad byte: even) -> 0; 0 / 1 / 255 unchanged
                    const float y0 = (float)((w << 24) >> 24), y1 = (float)((w << 16) >> 24);
                    const float y2 = (float)((w << 8) >> 24), y3 = (float)(w >> 24);
                    acc[t][0] = mfma32(g0[0], y0, acc[t][0]); acc[t][1] = mfma32(g1[0], y0, acc[t][1]);
                    acc[t][0] = mfma32(g0[1], y1, acc[t][0]); acc[t][1] = mfma32(g1[1], y1, acc[t][1]);
                    acc[t][0] = mfma32(g0[2], y2, acc[t][0]); acc[t][1] = mfma32(g1[2], y2, acc[t][1]);
                    acc[t][0] = mfma32(g0[3], y3, acc[t][0]); acc[t][1] = mfma32(g1[3], y3, acc[t][1]);
                }
            }
        }
    };
    const int64_t n_ptiles = (nb + F1_P - 1) / F1_P;
    int64_t tile = blockIdx.y;
    int b = 0;
    if (tile < n_ptiles) stage(tile, 0);
    for (; tile < n_ptiles; tile += gridDim.y, b ^= 1) {
        vx_wait_vmem();
        __syncthreads();
        if (tile + gridDim.y < n_ptiles) stage(tile + gridDim.y, b ^ 1);
        compute(b);
    }
    // slab: [W1-grad: 64 * J | b1-grad: 64]
    float* slab = slabs + (int64_t)blockIdx.y * slab_len;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int j = jg0 + F1_JW * wave + 32 * t + l31;
        if (j < J) {
#pragma unroll
            for (int ht = 0; ht < 2; ++ht)
#pragma unroll
                for (int r = 0; r < 16; ++r) slab[(int64_t)(32 * ht + crow32(r, half)) * J + j] = acc[t][ht][r];
        }
    }
    if (blockIdx.x == 0 && wave == 0) {                                // every wave holds the same sums
#pragma unroll
        for (int ht = 0; ht < 2; ++ht) {
            const float s = half_sum32(bsum[ht]);
            if (half == 0) slab[(int64_t)64 * J + 32 * ht + l31] = s;
        }
    }
}
