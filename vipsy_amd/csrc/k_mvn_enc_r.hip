// Amortized MVN guide forward, "wave-private persons" form (hidden_dim == 64, J % 4 == 0).
// Same mathematics / outputs as k_mvn_enc_fwd (k_mvn_enc.hip).  Structure:
//   * a wave owns 32 persons and walks ALL head rows for them, so x[p][k] is only ever touched by that
//     wave: the scatter  x[p,k] += M[p,(k,l)] * eps[p,l]  is a plain LDS read-modify-write done by the lower
//     half-wave in a fixed order -- no atomics (LDS float atomics were the bottleneck of the previous form:
//     profiles/r01_v2_pmc_sq_counters.json) and the result is run-to-run deterministic;
//   * fc1's output tile (h^T, rows = hidden units, lane = person) stays in registers and IS the B operand of
//     every head-row MFMA (K order of a step = the accumulator row map), so h never goes through LDS;
//   * head weights / fc1 weights stream global -> registers one tile ahead; no workgroup barrier anywhere
//     (the two waves of a workgroup share nothing); 26 KB LDS per wave -> 6 waves per CU.
#pragma once
#include "k_mvn_enc_fast.hip"

#define ER_THREADS 128
#define ER_WAVES 2
#define ER_WP 32                      // persons per wave

__host__ __device__ inline size_t enc_r_wave_floats(int D, int J) {
    const size_t a = (size_t)ER_WP * ef_ys(J) / 4;           // phase A: response bytes
    const size_t b = 2 * (size_t)ER_WP * enc_ds(D);          // phase B: eps | x
    return ((a > b ? a : b) + 64 + 3) & ~(size_t)3;          // + codes[32] + bias[32]
}
__host__ __device__ inline size_t enc_r_lds_floats(int D, int J) { return ER_WAVES * enc_r_wave_floats(D, J); }

__global__ __launch_bounds__(ER_THREADS, 2) void k_mvn_enc_fwd_r(
    EncDims dm, const uint8_t* __restrict__ y, const int64_t* __restrict__ rows, int64_t gid0,
    const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ W21,
    const float* __restrict__ b21, const float* __restrict__ W22, const float* __restrict__ b22,
    const float* __restrict__ eps_in, uint64_t seed, uint32_t step, const uint32_t* __restrict__ step_dev, uint32_t stream,
    float* __restrict__ h_out, float* __restrict__ x_out, float* __restrict__ eps_out,
    float* __restrict__ ldT, float* __restrict__ ent_out) {
    if (step_dev) step = *step_dev;                              // captured step: the Philox step lives in device memory
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int H = 64;
    const int D = dm.D, J = dm.J, DS = dm.DS, T = dm.T;
    const int YS = ef_ys(J);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    float* R1 = smem + wave * enc_r_wave_floats(D, J);
    int8_t* Yi = (int8_t*)R1;                                 // phase A
    float* eps_lds = R1;                                      // phase B  [32][DS]
    float* x_lds = R1 + ER_WP * DS;                           //          [32][DS]
    const size_t r1 = ((size_t)ER_WP * YS / 4 > 2 * (size_t)ER_WP * DS) ? (size_t)ER_WP * YS / 4 : 2 * (size_t)ER_WP * DS;
    uint32_t* codes = (uint32_t*)(R1 + r1);                   // [32]
    float* biasw = R1 + r1 + 32;                              // [32]
    const int64_t i0 = ((int64_t)blockIdx.x * ER_WAVES + wave) * ER_WP;
    const int p = l31;
    const int64_t i = i0 + p;
    if (i0 >= dm.nb) return;                                  // waves share nothing: no workgroup barrier below

    // ---------------------------------------------------------------- stage this wave's response rows (bytes)
    {
        const int YW = YS / 4, JW = J / 4;
        uint32_t* Yw = (uint32_t*)R1;
        for (int base = 0; base < ER_WP * YW; base += 64 * 8) {
            uint32_t v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int idx = base + q * 64 + lane;
                v[q] = 0u;
                if (idx < ER_WP * YW) {
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
                if (idx < ER_WP * YW) Yw[idx] = v[q];
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    // ---------------------------------------------------------------- phase A: fc1 (+ softplus), both hidden tiles
    f32x16 hreg[2];
    {
        f32x16 acc0 = zero16(), acc1 = zero16();
        const int nchunk = (J + 31) / 32;                     // 32 items per chunk: 16 k-steps x 2 hidden tiles
        auto loadA = [&](float4 (&A)[2][4], int c) {
            const int j0 = c * 32 + half * 16;
#pragma unroll
            for (int ht = 0; ht < 2; ++ht) {
                const float* src = W1 + (int64_t)(32 * ht + l31) * J + j0;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    A[ht][q] = (j0 + 4 * q + 4 <= J) ? *(const float4*)(src + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        auto compute = [&](const float4 (&A)[2][4], int c) {
            const int8_t* yp = Yi + p * YS + c * 32 + half * 16;
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
        float4 A0[2][4], A1[2][4];
        loadA(A0, 0);
        for (int c = 0; c < nchunk; c += 2) {
            if (c + 1 < nchunk) loadA(A1, c + 1);
            compute(A0, c);
            if (c + 2 < nchunk) loadA(A0, c + 2);
            if (c + 1 < nchunk) compute(A1, c + 1);
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
    }
    __builtin_amdgcn_wave_barrier();                          // response bytes no longer needed
    // ---------------------------------------------------------------- eps, x := 0
    {
        const int nblk = (D + 3) >> 2;
        for (int e = lane; e < ER_WP * nblk; e += 64) {
            const int pp = e / nblk, blk = e - pp * nblk;
            const int64_t ii = i0 + pp;
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            if (ii < dm.nb) {
                if (eps_in) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (4 * blk + q < D) z[q] = eps_in[ii * D + 4 * blk + q];
                } else {
                    const int64_t row = rows ? rows[ii] : ii;
                    z = philox_normal4(seed, step, stream, gid0 + row, (uint32_t)blk);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (4 * blk + q < D) {
                    eps_lds[pp * DS + 4 * blk + q] = z[q];
                    if (ii < dm.nb) eps_out[ii * D + 4 * blk + q] = z[q];
                }
        }
        for (int e = lane; e < ER_WP * DS; e += 64) x_lds[e] = 0.f;
        __builtin_amdgcn_wave_barrier();
        if (lane < ER_WP) eps_lds[lane * DS + D] = 1.0f;      // slot D: the "times one" of loc rows
    }
    __builtin_amdgcn_wave_barrier();
    // ---------------------------------------------------------------- phase B: all head rows, 32 per tile
    float ent_acc = 0.f;
    {
        const int64_t RT = (int64_t)T + D;
        const int n_rt = (int)((RT + 31) / 32);
        const float* ep = eps_lds + p * DS;
        float* xp = x_lds + p * DS;
        // A[ht][g] holds W[row][32ht + 8g + 4half .. +3]: exactly the hidden units this lane's hreg[ht][4g..4g+3] hold
        auto prefetch = [&](float4 (&A)[2][4], uint32_t& code, float& bias, int tt) {
            const int64_t r = (int64_t)tt * 32 + l31;
            const float* src = (r < T) ? W22 + r * H : (r < RT ? W21 + (r - T) * H : nullptr);
            code = enc_row_code_fast(r, T, D);
            bias = (r < T) ? b22[r] : (r < RT ? b21[r - T] : 0.f);
#pragma unroll
            for (int ht = 0; ht < 2; ++ht)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    A[ht][g] = src ? *(const float4*)(src + 32 * ht + 8 * g + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
        };
        auto tile = [&](const float4 (&A)[2][4], uint32_t code, float bias) {
            if (half == 0) { codes[l31] = code; biasw[l31] = bias; }
            __builtin_amdgcn_wave_barrier();
            f32x16 a = zero16();
#pragma unroll
            for (int ht = 0; ht < 2; ++ht)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    a = mfma32(A[ht][g].x, hreg[ht][4 * g + 0], a);
                    a = mfma32(A[ht][g].y, hreg[ht][4 * g + 1], a);
                    a = mfma32(A[ht][g].z, hreg[ht][4 * g + 2], a);
                    a = mfma32(A[ht][g].w, hreg[ht][4 * g + 3], a);
                }
            // epilogue: rows 8g + 4*half + j of the tile live in a[4g + j] of this lane (person p)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int rl = 8 * g + 4 * half;
                const uint4 cd = *(const uint4*)(codes + rl);
                const float4 bs = *(const float4*)(biasw + rl);
                const uint32_t cdv[4] = {cd.x, cd.y, cd.z, cd.w};
                const float bsv[4] = {bs.x, bs.y, bs.z, bs.w};
                float mine[4], other[4];
#pragma unroll
                for (int jx = 0; jx < 4; ++jx) {
                    const uint32_t cc = cdv[jx];
                    const float v = a[4 * g + jx] + bsv[jx];
                    float contrib = v * ep[cc & 0xFFFFu];                               // tril(M,-1) eps | loc | padding
                    if (cc & FC_DIAG) {                                                 // half-wave uniform, rare
                        const int k = (int)(cc & 0xFFFFu);
                        const float ld = expf(v);                                       // exp(diag M): vi.py:686
                        contrib = ld * ep[k];
                        ent_acc += v;
                        if (i < dm.nb) ldT[(int64_t)k * dm.nb + i] = ld;
                    }
                    mine[jx] = contrib;
                }
#pragma unroll
                for (int jx = 0; jx < 4; ++jx) other[jx] = __shfl_xor(mine[jx], 32, 64);
                if (half == 0) {
                    // rows 8g..8g+3 are mine, rows 8g+4..8g+7 came from the upper half; merge runs of equal k
                    const uint4 co = *(const uint4*)(codes + 8 * g + 4);
                    const uint32_t kk[8] = {(cd.x >> 16) & 0x7FFFu, (cd.y >> 16) & 0x7FFFu, (cd.z >> 16) & 0x7FFFu,
                                            (cd.w >> 16) & 0x7FFFu, (co.x >> 16) & 0x7FFFu, (co.y >> 16) & 0x7FFFu,
                                            (co.z >> 16) & 0x7FFFu, (co.w >> 16) & 0x7FFFu};
                    const float vv[8] = {mine[0], mine[1], mine[2], mine[3], other[0], other[1], other[2], other[3]};
                    uint32_t curk = kk[0];
                    float part = vv[0];
#pragma unroll
                    for (int e = 1; e < 8; ++e) {
                        if (kk[e] != curk) {                                             // wave-uniform (codes are)
                            xp[curk] += part;
                            curk = kk[e];
                            part = vv[e];
                        } else {
                            part += vv[e];
                        }
                    }
                    xp[curk] += part;
                }
            }
            __builtin_amdgcn_wave_barrier();
        };
        float4 A0[2][4], A1[2][4];
        uint32_t c0 = 0, c1 = 0;
        float bz0 = 0.f, bz1 = 0.f;
        prefetch(A0, c0, bz0, 0);
        for (int tt = 0; tt < n_rt; tt += 2) {
            if (tt + 1 < n_rt) prefetch(A1, c1, bz1, tt + 1);
            tile(A0, c0, bz0);
            if (tt + 2 < n_rt) prefetch(A0, c0, bz0, tt + 2);
            if (tt + 1 < n_rt) tile(A1, c1, bz1);
        }
    }
    __builtin_amdgcn_wave_barrier();
    // ---------------------------------------------------------------- write x, entropy part
    {
        const int pv = (int)((dm.nb - i0) < ER_WP ? (dm.nb - i0) : ER_WP);
        for (int e = lane; e < pv * D; e += 64) {
            const int pp = e / D, k = e - pp * D;
            x_out[i0 * D + e] = x_lds[pp * DS + k];
        }
        ent_acc += __shfl_xor(ent_acc, 32, 64);
        if (half == 0 && i < dm.nb) {
            float s = 0.f;
            for (int k = 0; k < D; ++k) { const float e = eps_lds[p * DS + k]; s += e * e; }
            ent_out[i] = 0.5f * s + ent_acc;                  // -log q + const = 0.5|eps|^2 + sum_k M_kk
        }
    }
}
