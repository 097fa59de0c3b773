// Amortized MVN guide on the PACKED head layout (k_pack.hip): hidden_dim == 64, J % 4 == 0, D % 4 == 0.
// Same mathematics and outputs as the generic kernels (k_mvn_enc.hip / k_mvn_enc_bwd.hip).
//
// Because every aligned group of 8 packed rows has one type and one k with l0 % 8 == 0, nothing is decoded
// per element:
//   forward : per 4 accumulator rows one 16-byte eps read + 4 FMAs; the partial x[p][k] lives in a REGISTER
//             and is flushed to LDS (plain read-modify-write by the lower half-wave, fixed order, no atomics)
//             only when k changes -- about 2 D flushes per person instead of T scatter-adds;
//   backward: V[r,p] = gx[p][k] * eps[p][l0 + ..] -- one 16-byte eps read per 4 MFMA B operands.
// The bias of a head row enters as one extra k-step (A = bias, B = 1), so the epilogue has no bias lookup.
#pragma once
#include "k_pack.hip"
#include "k_mvn_enc_fast.hip"
#include "k_mvn_enc_bwd.hip"

// LDS stride (floats) for arrays read with ds_read_b128 by lanes = persons: multiple of 4 with an ODD
// multiple-of-4 count (conflict-free 16-lane groups), covering index roundup8(D) + 3
__host__ __device__ inline int pk_dse(int D) {
    int s = ((D + 7) / 8 * 8 + 4) / 4;
    if ((s & 1) == 0) ++s;
    return 4 * s;
}

#define EP_THREADS 256
#define EP_WAVES 4
#define EP_WP 32

__host__ __device__ inline size_t enc_p_wave_floats(int D, int J) {
    const size_t a = (size_t)EP_WP * ef_ys(J) / 4;
    const size_t b = (size_t)EP_WP * pk_dse(D) + (size_t)EP_WP * (size_t)((D + 3) & ~3);   // eps | x, both 16-byte rows
    return ((a > b ? a : b) + 3) & ~(size_t)3;
}
__host__ __device__ inline size_t enc_p_lds_floats(int D, int J) { return EP_WAVES * enc_p_wave_floats(D, J); }

__global__ __launch_bounds__(EP_THREADS, 1) void k_mvn_enc_fwd_p(
    EncDims dm, const uint8_t* __restrict__ y, const int64_t* __restrict__ rows, int64_t gid0,
    const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ Wp,
    const float* __restrict__ bp, const uint32_t* __restrict__ gtab, const float* __restrict__ eps_in,
    uint64_t seed, uint32_t step, const uint32_t* __restrict__ step_dev, uint32_t stream, float* __restrict__ h_out, float* __restrict__ x_out,
    float* __restrict__ eps_out, float* __restrict__ ldT, float* __restrict__ ent_out,
    float* __restrict__ hT_out /*[64][nb] or null*/, float* __restrict__ epsT_out /*[D][nb] or null*/) {
    if (step_dev) step = *step_dev;                              // captured step: the Philox step lives in device memory
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int H = 64;
    const int D = dm.D, J = dm.J;
    const int DS = pk_dse(D), DX = (D + 3) & ~3;
    const int YS = ef_ys(J);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    float* R1 = smem + wave * enc_p_wave_floats(D, J);
    int8_t* Yi = (int8_t*)R1;                                 // phase A
    float* eps_lds = R1;                                      // phase B  [32][DS]
    float* x_lds = R1 + EP_WP * DS;                           //          [32][DX]
    const int64_t i0 = ((int64_t)blockIdx.x * EP_WAVES + wave) * EP_WP;
    const int p = l31;
    const int64_t i = i0 + p;
    if (i0 >= dm.nb) return;                                  // waves share nothing: no workgroup barrier below

    // ---------------------------------------------------------------- stage this wave's response rows (bytes)
    // Dense mode (no row gather, J / 4 odd, not the last rows of y): the wave's 32 rows are one contiguous run of
    // 32 J bytes -> ceil(32 J / 1024) full-wave DMA transfers, all in flight at once; LDS row stride = J bytes
    // (J / 4 words, odd: conflict-free word reads by lanes = persons).  Otherwise: padded rows through registers.
    const int n_ydma = (32 * J + 1023) / 1024;
    const bool ydense = !rows && ((J >> 2) & 1) && i0 + EP_WP <= dm.nb && (i0 * J + (int64_t)n_ydma * 1024 <= dm.nb * (int64_t)J) &&
                        (size_t)n_ydma * 1024 <= enc_p_wave_floats(D, J) * sizeof(float);
    const int ysr = ydense ? J : YS;                          // LDS row stride of the response bytes
    if (ydense) {
        const uint8_t* src = y + i0 * J + 16 * lane;
        const uint32_t lb = lds_addr_uniform(R1);
        for (int d = 0; d < n_ydma; ++d) dma16(src + d * 1024, lb + (uint32_t)d * 1024u);
        vx_wait_vmem();
    } else {
        const int YW = YS / 4, JW = J / 4;
        uint32_t* Yw = (uint32_t*)R1;
        for (int base = 0; base < EP_WP * YW; base += 64 * 8) {
            uint32_t v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int idx = base + q * 64 + lane;
                v[q] = 0u;
                if (idx < EP_WP * YW) {
                    const int pp = idx / YW, wq = idx - pp * YW;
                    const int64_t ii = i0 + pp;
                    if (wq < JW && ii < dm.nb) {
                        const int64_t row = rows ? rows[ii] : ii;
                        v[q] = *(const uint32_t*)(y + row * J + 4 * wq);    // bytes 0/1/255 == int8 0/1/-1 (vi.py:689-691)
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int idx = base + q * 64 + lane;
                if (idx < EP_WP * YW) Yw[idx] = v[q];
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    // ---------------------------------------------------------------- phase A: fc1 (+ softplus), both hidden tiles
    f32x16 hreg[2];
    {
        f32x16 acc0 = zero16(), acc1 = zero16();
        const int nfull = J / 32;                             // chunks of 32 items entirely inside [0, J)
        // W1 chunk of this lane: rows l31 and 32 + l31, items c * 32 + half * 16 .. + 15.  No load sits under a branch
        // and the ring is four chunks deep: the L2 latency of a chunk (~3 chunks of MFMA time) stays off the chain.
        auto loadA = [&](float4 (&A)[2][4], int c) {
            c = c < nfull ? c : nfull - 1;                    // past the end: reload the last chunk (never used)
            const int j0 = c * 32 + half * 16;
#pragma unroll
            for (int ht = 0; ht < 2; ++ht) {
                const float* src = W1 + (int64_t)(32 * ht + l31) * J + j0;
#pragma unroll
                for (int q = 0; q < 4; ++q) A[ht][q] = *(const float4*)(src + 4 * q);
            }
        };
        auto compute = [&](const float4 (&A)[2][4], int c) {
            const int8_t* yp = Yi + p * ysr + c * 32 + half * 16;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int w = *(const int*)(yp + 4 * q);
                const float y0 = (float)((w << 24) >> 24), y1 = (float)((w << 16) >> 24);
                const float y2 = (float)((w << 8) >> 24), y3 = (float)(w >> 24);
                acc0 = mfma32(A[0][q].x, y0, acc0); acc1 = mfma32(A[1][q].x, y0, acc1);
                acc0 = mfma32(A[0][q].y, y1, acc0); acc1 = mfma32(A[1][q].y, y1, acc1);
                acc0 = mfma32(A[0][q].z, y2, acc0); acc1 = mfma32(A[1][q].z, y2, acc1);
                acc0 = mfma32(A[0][q].w, y3, acc0); acc1 = mfma32(A[1][q].w, y3, acc1);
            }
        };
        if (nfull > 0) {
            float4 A[4][2][4];
            loadA(A[0], 0); loadA(A[1], 1); loadA(A[2], 2);
            const int nloop = nfull;
            for (int c = 0; c < nloop; c += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    loadA(A[(u + 3) & 3], c + u + 3);
                    if (c + u < nloop) compute(A[u], c + u);
                }
            }
        }
        if (nfull * 32 < J) {                                 // ragged last chunk: items past J contribute nothing
            float4 At[2][4];
            const int j0 = nfull * 32 + half * 16;
#pragma unroll
            for (int ht = 0; ht < 2; ++ht) {
                const float* src = W1 + (int64_t)(32 * ht + l31) * J + j0;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    At[ht][q] = (j0 + 4 * q + 4 <= J) ? *(const float4*)(src + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            compute(At, nfull);
        }
#pragma unroll
        for (int ht = 0; ht < 2; ++ht) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int hh0 = 32 * ht + 8 * g + 4 * half;
                const float4 bb = *(const float4*)(b1 + hh0);
                float4 hv;
                hv.x = softplusf_((ht ? acc1 : acc0)[4 * g + 0] + bb.x);            // vi.py:449
                hv.y = softplusf_((ht ? acc1 : acc0)[4 * g + 1] + bb.y);
                hv.z = softplusf_((ht ? acc1 : acc0)[4 * g + 2] + bb.z);
                hv.w = softplusf_((ht ? acc1 : acc0)[4 * g + 3] + bb.w);
                hreg[ht][4 * g + 0] = hv.x; hreg[ht][4 * g + 1] = hv.y;
                hreg[ht][4 * g + 2] = hv.z; hreg[ht][4 * g + 3] = hv.w;
                if (i < dm.nb) *(float4*)(h_out + i * H + hh0) = hv;
            }
        }
        if (hT_out && i < dm.nb) {                            // dimension-major copy for the weight-gradient kernel
#pragma unroll
            for (int ht = 0; ht < 2; ++ht)
#pragma unroll
                for (int r = 0; r < 16; ++r) hT_out[(int64_t)(32 * ht + crow32(r, half)) * dm.nb + i] = hreg[ht][r];
        }
    }
    __builtin_amdgcn_wave_barrier();                          // response bytes no longer needed
    // ---------------------------------------------------------------- eps (zero padded to DS), x := 0
    {
        for (int e = lane; e < EP_WP * (DS + DX); e += 64) R1[e] = 0.f;
        __builtin_amdgcn_wave_barrier();
        const int nblk = D >> 2;                               // D % 4 == 0 on this path
        for (int e = lane; e < EP_WP * nblk; e += 64) {
            const int pp = e / nblk, blk = e - pp * nblk;
            int64_t ii = i0 + pp;
            if (ii >= dm.nb) ii = dm.nb - 1;                   // absent persons: any finite values, never stored
            f32x4 z;
            if (eps_in) {
                z = *(const f32x4*)(eps_in + ii * D + 4 * blk);
            } else {
                const int64_t row = rows ? rows[ii] : ii;
                z = philox_normal4(seed, step, stream, gid0 + row, (uint32_t)blk);
            }
            *(f32x4*)(eps_lds + pp * DS + 4 * blk) = z;
            if (i0 + pp < dm.nb) *(f32x4*)(eps_out + ii * D + 4 * blk) = z;
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (epsT_out && i < dm.nb) {                              // dimension-major copy: 128-byte rows per half-wave
#pragma unroll 4
        for (int k = half; k < D; k += 2) epsT_out[(int64_t)k * dm.nb + i] = eps_lds[p * DS + k];
    }
    // ---------------------------------------------------------------- phase B: packed head rows, 32 per tile
    float ent_acc = 0.f;
    {
        const int n_off = pk_off_total(D) / 32;                // multiple of 3
        const int n_sec = pk_sec(D) / 32;
        const float* ep = eps_lds + p * DS;
        float* xp = x_lds + p * DX;
        const float one = (half == 0) ? 1.0f : 0.f;
        // A[ht][g] = Wp[row][32ht + 8g + 4half .. +3]: the hidden units this lane's hreg[ht][4g..4g+3] hold.
        // NOTE: nothing in the OFF loop is conditional -- with loads issued under a condition hipcc falls back to
        // s_waitcnt vmcnt(0) at every use and the whole L2 latency is exposed each tile.
        auto prefetch = [&](float4 (&A)[2][4], float& biasA, uint4& gc, int tt) {
            const float* src = Wp + ((int64_t)tt * 32 + l31) * H;
#pragma unroll
            for (int ht = 0; ht < 2; ++ht)
#pragma unroll
                for (int g = 0; g < 4; ++g) A[ht][g] = *(const float4*)(src + 32 * ht + 8 * g + 4 * half);
            const float bv = bp[tt * 32 + l31];
            biasA = (half == 0) ? bv : 0.f;
            gc = *(const uint4*)(gtab + 4 * tt);               // uniform address: 4 group codes of this tile
        };
        auto mma = [&](const float4 (&A)[2][4], float biasA) {
            f32x16 a = zero16();
            a = mfma32(biasA, one, a);                         // + bias[row] as a 65th k-step
#pragma unroll
            for (int ht = 0; ht < 2; ++ht)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    a = mfma32(A[ht][g].x, hreg[ht][4 * g + 0], a);
                    a = mfma32(A[ht][g].y, hreg[ht][4 * g + 1], a);
                    a = mfma32(A[ht][g].z, hreg[ht][4 * g + 2], a);
                    a = mfma32(A[ht][g].w, hreg[ht][4 * g + 3], a);
                }
            return a;
        };
        // ---- OFF section: x[p][k] += sum_l M[p,(k,l)] eps[p,l]; the partial sum of the current k stays in a register
        uint32_t cur_k = 1;                                    // the first packed group belongs to k = 1
        float cur_part = 0.f;
        auto flush = [&]() {                                   // every k of the OFF section is flushed exactly once:
            const float tot = half_sum32(cur_part);            // a plain store, no read-modify-write, no LDS shuffle
            if (half == 0) xp[cur_k] = tot;
        };
        auto tile_off = [&](const float4 (&A)[2][4], float biasA, uint4 gc) {
            // eps reads are issued BEFORE the MFMA chain: an LDS read in the dependent path costs 20-45 % of the
            // matrix pipe (tools/mfma_ubench.hip)
            const uint32_t gcv[4] = {gc.x, gc.y, gc.z, gc.w};
            uint32_t kq[4];
            float4 e4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const uint32_t code = __builtin_amdgcn_readfirstlane(gcv[g]);
                kq[g] = (code >> 12) & 0xFFFFu;
                e4[g] = *(const float4*)(ep + (code & 0xFFFu) + 4 * half);
            }
            __builtin_amdgcn_sched_barrier(0);                 // keep the reads above the chain (the scheduler sinks them)
            const f32x16 a = mma(A, biasA);
#pragma unroll
            for (int g = 0; g < 4; ++g) {                      // rows (k, l0 + 4half + j) live in a[4g + j]
                const float part = a[4 * g + 0] * e4[g].x + a[4 * g + 1] * e4[g].y + a[4 * g + 2] * e4[g].z +
                                   a[4 * g + 3] * e4[g].w;
                const bool changed = kq[g] != cur_k;                                  // wave-uniform, rare
                if (__builtin_expect(changed, 0)) flush();                            // one-sided branch, falls through
                cur_part = part + (changed ? 0.f : cur_part);
                cur_k = kq[g];
            }
        };
        {
            float4 A0[2][4], A1[2][4], A2[2][4];
            float bA0, bA1, bA2;
            uint4 g0, g1, g2;
            prefetch(A0, bA0, g0, 0);
            prefetch(A1, bA1, g1, 1);
            for (int tt = 0; tt < n_off; tt += 3) {            // straight-line body; tiles n_off, n_off+1 exist (DIAG)
                prefetch(A2, bA2, g2, tt + 2);
                tile_off(A0, bA0, g0);
                prefetch(A0, bA0, g0, tt + 3);
                tile_off(A1, bA1, g1);
                prefetch(A1, bA1, g1, tt + 4);
                tile_off(A2, bA2, g2);
            }
            flush();
            // ---- DIAG section (exp(M_kk) eps_k, entropy, ldT) and LOC section (the loc head): 2 * n_sec tiles, on the
            // same three-deep weight ring (A0 / A1 already hold the first two of them).  The 16 x entries a lane
            // updates are read together, updated and written together (one LDS round trip per tile).
            const int t_end = n_off + 2 * n_sec;
            auto tile_sec = [&](const float4 (&A)[2][4], float biasA, int t2) {
                const f32x16 a = mma(A, biasA);
                const bool is_diag = t2 < n_off + n_sec;
                const int k0 = 32 * (t2 - (is_diag ? n_off : n_off + n_sec));
                const bool allv = k0 + 32 <= D;                        // wave-uniform: no per-entry bounds below
                float xo[16], ev[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int kk = k0 + crow32(r, half);
                    kk = (allv || kk < D) ? kk : D - 1;
                    xo[r] = xp[kk];
                    ev[r] = ep[kk];
                }
                if (is_diag) {                                         // exp(diag M) eps_k, entropy, ldT  (vi.py:686)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int kk = k0 + crow32(r, half);
                        const float ld = __expf(a[r]);
                        const bool ok = allv || kk < D;
                        if (ok) {
                            xp[kk] = fmaf(ld, ev[r], xo[r]);
                            ent_acc += a[r];
                            if (i < dm.nb) ldT[(int64_t)kk * dm.nb + i] = ld;
                        }
                    }
                } else {                                               // loc head (vi.py:450)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int kk = k0 + crow32(r, half);
                        if (allv || kk < D) xp[kk] = xo[r] + a[r];
                    }
                }
            };
            auto clampt = [&](int t2) { return t2 < t_end ? t2 : t_end - 1; };   // past the end: reload the last tile
            for (int tt = n_off; tt < t_end; tt += 3) {
                prefetch(A2, bA2, g2, clampt(tt + 2));
                tile_sec(A0, bA0, tt);
                prefetch(A0, bA0, g0, clampt(tt + 3));
                if (tt + 1 < t_end) tile_sec(A1, bA1, tt + 1);
                prefetch(A1, bA1, g1, clampt(tt + 4));
                if (tt + 2 < t_end) tile_sec(A2, bA2, tt + 2);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    // ---------------------------------------------------------------- write x, entropy part
    {
        const int pv = (int)((dm.nb - i0) < EP_WP ? (dm.nb - i0) : EP_WP);
        const int c4 = D >> 2;
        for (int e = lane; e < pv * c4; e += 64) {
            const int pp = e / c4, c = e - pp * c4;
            *(f32x4*)(x_out + (i0 + pp) * D + 4 * c) = *(const f32x4*)(x_lds + pp * DX + 4 * c);
        }
        ent_acc += __shfl_xor(ent_acc, 32, 64);
        if (half == 0 && i < dm.nb) {
            float s = 0.f;
            for (int k = 0; k < D; ++k) { const float e = eps_lds[p * DS + k]; s += e * e; }
            ent_out[i] = 0.5f * s + ent_acc;                  // -log q + const = 0.5|eps|^2 + sum_k M_kk
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// backward, person-parallel: gh^T[hh,p] = sum_r Wp[r,hh] V[r,p];  ghpre = gh * sigmoid(pre)
#define BHP_ROWS 64

__host__ __device__ inline size_t enc_bwdh_p_lds_floats(int D) {
    return (((size_t)ENC_P * enc_ds(D) + 3) & ~(size_t)3) + (size_t)ENC_P * pk_dse(D) + (size_t)BHP_ROWS * 64;
}

__global__ __launch_bounds__(ENC_THREADS, 2) void k_mvn_enc_bwd_h_p(
    EncDims dm, float scale, const float* __restrict__ Wp, const uint32_t* __restrict__ gtab,
    const float* __restrict__ h_in, const float* __restrict__ eps_in, const float* __restrict__ ldT,
    const float* __restrict__ gx_in, float* __restrict__ ghpre_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int H = 64;
    const int D = dm.D;
    const int DG = dm.DS;                               // odd stride: gx is read one float per lane (lane = person)
    const int DE = pk_dse(D);                           // 16-byte friendly stride: eps is read four floats per lane
    float* gx_lds = smem;                                                    // [P][DG]
    float* eps_lds = smem + (((size_t)ENC_P * DG + 3) & ~(size_t)3);         // [P][DE]
    float* Wt = eps_lds + ENC_P * DE;                                        // [BHP_ROWS][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int64_t i0 = (int64_t)blockIdx.x * ENC_P;
    const int n_tiles = pk_rows(D) / BHP_ROWS + ((pk_rows(D) % BHP_ROWS) ? 1 : 0);
    const int Rp = pk_rows(D);
    const int pvalid = (int)((dm.nb - i0) < ENC_P ? (dm.nb - i0) : ENC_P);

    for (int e = tid; e < ENC_P * DE; e += ENC_THREADS) eps_lds[e] = 0.f;
    __syncthreads();
    {
        const int c4 = D / 4, n4 = ENC_P * c4;
        const float4* g4 = (const float4*)(gx_in + i0 * D);
        const float4* e4 = (const float4*)(eps_in + i0 * D);
        for (int base = 0; base < n4; base += ENC_THREADS * 4) {
            float4 a[4], b[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = base + q * ENC_THREADS + tid;
                const bool ok = idx < pvalid * c4;
                a[q] = ok ? g4[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
                b[q] = ok ? e4[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = base + q * ENC_THREADS + tid;
                if (idx < n4) {
                    const int pp = idx / c4, c = idx - pp * c4;
                    float* dg = gx_lds + pp * DG + 4 * c;
                    dg[0] = a[q].x; dg[1] = a[q].y; dg[2] = a[q].z; dg[3] = a[q].w;
                    *(float4*)(eps_lds + pp * DE + 4 * c) = b[q];
                }
            }
        }
    }
    const int u = wave & 1, ht = wave >> 1;
    const int p = 32 * u + l31;
    const int64_t i = i0 + p;
    f32x16 acc = zero16();
    float4 wp[4];
    uint4 gcn = make_uint4(0, 0, 0, 0);                                 // group codes of the prefetched tile (this half)
    auto prefetch = [&](int tile) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int f = tid + ENC_THREADS * q;                        // 1024 float4 per 64-row tile
            const int r = tile * BHP_ROWS + (f >> 4);
            wp[q] = (r < Rp) ? *(const float4*)(Wp + (int64_t)r * H + 4 * (f & 15)) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const int g0i = tile * (BHP_ROWS / 8) + half * 4;               // gtab is padded by 8 entries past Rp / 8
        gcn = (8 * g0i < Rp) ? *(const uint4*)(gtab + g0i) : make_uint4(0, 0, 0, 0);
    };
    prefetch(0);
    const float* gx_p = gx_lds + p * DG;
    const float* eps_p = eps_lds + p * DE;
    for (int tile = 0; tile < n_tiles; ++tile) {
        __syncthreads();                                                // previous MFMA phase done with Wt
#pragma unroll
        for (int q = 0; q < 4; ++q) ((float4*)Wt)[tid + ENC_THREADS * q] = wp[q];
        __syncthreads();
        const uint32_t gcur[4] = {gcn.x, gcn.y, gcn.z, gcn.w};
        if (tile + 1 < n_tiles) prefetch(tile + 1);
        // K order: lane-half `half` walks packed rows 32*half + 0..31 of the tile = 4 groups of 8 rows
        const float* ap = Wt + (half * 32) * H + 32 * ht + l31;
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
            const uint32_t code = gcur[gg];                             // uniform per half-wave
            const uint32_t type = code >> 28, k = (code >> 12) & 0xFFFFu, l0 = code & 0xFFFu;
            float v[8];
            if (type == PK_OFF) {
                const float gk = gx_p[k];
                const float4 ea = *(const float4*)(eps_p + l0), eb = *(const float4*)(eps_p + l0 + 4);
                v[0] = gk * ea.x; v[1] = gk * ea.y; v[2] = gk * ea.z; v[3] = gk * ea.w;
                v[4] = gk * eb.x; v[5] = gk * eb.y; v[6] = gk * eb.z; v[7] = gk * eb.w;
            } else if (type == PK_LOC) {
#pragma unroll
                for (int jx = 0; jx < 8; ++jx) v[jx] = ((int)k + jx < D) ? gx_p[k + jx] : 0.f;
            } else if (type == PK_DIAG) {
#pragma unroll
                for (int jx = 0; jx < 8; ++jx) {
                    const int kk = (int)k + jx;
                    v[jx] = (kk < D && i < dm.nb) ? gx_p[kk] * eps_p[kk] * ldT[(int64_t)kk * dm.nb + i] + scale : 0.f;
                }
            } else {
#pragma unroll
                for (int jx = 0; jx < 8; ++jx) v[jx] = 0.f;
            }
#pragma unroll
            for (int jx = 0; jx < 8; ++jx) acc = mfma32(ap[(8 * gg + jx) * H], v[jx], acc);
        }
    }
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

// ------------------------------------------------------------------------------------------------------------
// NormEncoder forward for hidden_dim == 64, J % 4 == 0 (vi.py:417-435): the fc1 phase of k_mvn_enc_fwd_p (DMA-staged
// response rows, W1 on a 4-deep register ring, fp32 MFMA) followed by the two 1-row heads as per-lane dot products
// over the 32 hidden units a lane holds (+ the half-wave sum).  One wave = 32 persons, no workgroup barrier.
#define NE_THREADS 256
#define NE_WAVES 4
__host__ __device__ inline size_t norm_fast_wave_floats(int J) {
    const size_t a = (size_t)EP_WP * ef_ys(J) / 4, b = (size_t)(((32 * J + 1023) / 1024) * 256);
    return ((a > b ? a : b) + 3) & ~(size_t)3;
}
__global__ __launch_bounds__(NE_THREADS, 1) void k_norm_enc_fwd_fast(
    EncDims dm, const uint8_t* __restrict__ y, const int64_t* __restrict__ rows, const float* __restrict__ W1,
    const float* __restrict__ b1, const float* __restrict__ W21, const float* __restrict__ b21,
    const float* __restrict__ W22, const float* __restrict__ b22, float* __restrict__ h_out,
    float* __restrict__ loc_out, float* __restrict__ raw_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int H = 64;
    const int J = dm.J, YS = ef_ys(J);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    float* R1 = smem + wave * norm_fast_wave_floats(J);
    const int8_t* Yi = (const int8_t*)R1;
    const int64_t i0 = ((int64_t)blockIdx.x * NE_WAVES + wave) * EP_WP;
    const int p = l31;
    const int64_t i = i0 + p;
    if (i0 >= dm.nb) return;
    const int n_ydma = (32 * J + 1023) / 1024;
    const bool ydense = !rows && ((J >> 2) & 1) && i0 + EP_WP <= dm.nb && (i0 * J + (int64_t)n_ydma * 1024 <= dm.nb * (int64_t)J);
    const int ysr = ydense ? J : YS;
    if (ydense) {
        const uint8_t* src = y + i0 * J + 16 * lane;
        const uint32_t lb = lds_addr_uniform(R1);
        for (int d = 0; d < n_ydma; ++d) dma16(src + d * 1024, lb + (uint32_t)d * 1024u);
        vx_wait_vmem();
    } else {
        const int YW = YS / 4, JW = J / 4;
        uint32_t* Yw = (uint32_t*)R1;
        for (int e = lane; e < EP_WP * YW; e += 64) {
            const int pp = e / YW, wq = e - pp * YW;
            const int64_t ii = i0 + pp;
            uint32_t v = 0u;
            if (wq < JW && ii < dm.nb) {
                const int64_t row = rows ? rows[ii] : ii;
                v = *(const uint32_t*)(y + row * J + 4 * wq);              // bytes 0/1/255 == int8 0/1/-1 (vi.py:680-682)
            }
            Yw[e] = v;
        }
    }
    __builtin_amdgcn_wave_barrier();
    f32x16 acc0 = zero16(), acc1 = zero16();
    const int nfull = J / 32;
    auto loadA = [&](float4 (&A)[2][4], int c) {
        c = c < nfull ? c : nfull - 1;
        const int j0 = c * 32 + half * 16;
#pragma unroll
        for (int ht = 0; ht < 2; ++ht) {
            const float* src = W1 + (int64_t)(32 * ht + l31) * J + j0;
#pragma unroll
            for (int q = 0; q < 4; ++q) A[ht][q] = *(const float4*)(src + 4 * q);
        }
    };
    auto compute = [&](const float4 (&A)[2][4], int c) {
        const int8_t* yp = Yi + p * ysr + c * 32 + half * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int w = *(const int*)(yp + 4 * q);
            const float y0 = (float)((w << 24) >> 24), y1 = (float)((w << 16) >> 24);
            const float y2 = (float)((w << 8) >> 24), y3 = (float)(w >> 24);
            acc0 = mfma32(A[0][q].x, y0, acc0); acc1 = mfma32(A[1][q].x, y0, acc1);
            acc0 = mfma32(A[0][q].y, y1, acc0); acc1 = mfma32(A[1][q].y, y1, acc1);
            acc0 = mfma32(A[0][q].z, y2, acc0); acc1 = mfma32(A[1][q].z, y2, acc1);
            acc0 = mfma32(A[0][q].w, y3, acc0); acc1 = mfma32(A[1][q].w, y3, acc1);
        }
    };
    if (nfull > 0) {
        float4 A[4][2][4];
        loadA(A[0], 0); loadA(A[1], 1); loadA(A[2], 2);
        for (int c = 0; c < nfull; c += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                loadA(A[(u + 3) & 3], c + u + 3);
                if (c + u < nfull) compute(A[u], c + u);
            }
        }
    }
    if (nfull * 32 < J) {
        float4 At[2][4];
        const int j0 = nfull * 32 + half * 16;
#pragma unroll
        for (int ht = 0; ht < 2; ++ht) {
            const float* src = W1 + (int64_t)(32 * ht + l31) * J + j0;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                At[ht][q] = (j0 + 4 * q + 4 <= J) ? *(const float4*)(src + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        compute(At, nfull);
    }
    float sl = 0.f, sr = 0.f;                                 // partial loc / raw over this lane's 32 hidden units
#pragma unroll
    for (int ht = 0; ht < 2; ++ht)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int hh0 = 32 * ht + 8 * g + 4 * half;
            const float4 bb = *(const float4*)(b1 + hh0);
            // the head rows sit behind odd-length bias segments in the flat parameter buffer: 4-byte loads
            const float4 w21 = make_float4(W21[hh0], W21[hh0 + 1], W21[hh0 + 2], W21[hh0 + 3]);
            const float4 w22 = make_float4(W22[hh0], W22[hh0 + 1], W22[hh0 + 2], W22[hh0 + 3]);
            float4 hv;
            hv.x = softplusf_((ht ? acc1 : acc0)[4 * g + 0] + bb.x);               // vi.py:432
            hv.y = softplusf_((ht ? acc1 : acc0)[4 * g + 1] + bb.y);
            hv.z = softplusf_((ht ? acc1 : acc0)[4 * g + 2] + bb.z);
            hv.w = softplusf_((ht ? acc1 : acc0)[4 * g + 3] + bb.w);
            sl += hv.x * w21.x + hv.y * w21.y + hv.z * w21.z + hv.w * w21.w;
            sr += hv.x * w22.x + hv.y * w22.y + hv.z * w22.z + hv.w * w22.w;
            if (i < dm.nb) *(float4*)(h_out + i * H + hh0) = hv;
        }
    sl = half_sum32(sl);
    sr = half_sum32(sr);
    if (half == 0 && i < dm.nb) {
        loc_out[i] = sl + b21[0];
        raw_out[i] = sr + b22[0];
    }
}
