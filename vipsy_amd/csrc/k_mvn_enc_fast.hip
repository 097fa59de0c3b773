// Fast path of the amortized MVN guide forward for hidden_dim == 64 (the reference default, vi.py:661)
// and J % 4 == 0.  Same mathematics and outputs as k_mvn_enc_fwd (k_mvn_enc.hip); differences are
// purely about feeding the matrix pipe:
//   * head weights (fc22 | fc21 rows) go global -> registers, 128 contiguous bytes per lane, prefetched
//     one 32-row tile ahead; they are the MFMA A operand directly (K order = [half][s]), no LDS staging
//     and no workgroup barrier inside the head loop -- each wave walks its own row tiles;
//   * the response tile is staged once as raw bytes (0/1/255 reinterpreted as int8 0/1/-1 is exactly the
//     encoder input of vi.py:689-691) and fc1's weights also stream through registers;
//   * 70 KB of LDS and <= 256 VGPRs -> two workgroups (two waves per SIMD) per CU, so one wave's epilogue
//     (VALU + LDS atomics) overlaps the other's MFMAs.
#pragma once
#include "k_mvn_enc.hip"

#define EF_HS 65                       // h_lds stride (odd: conflict-free person-major reads)
#define FC_DIAG 0x80000000u

// Row code of the fast kernels: (kx << 16) | lx such that for every non-diagonal row the contribution
// is simply  x[p][kx] += v * eps[p][lx]  with no branch:
//   tril off-diagonal (k,l): kx = k, lx = l;   loc row k: kx = k, lx = D (slot D of every eps row holds 1);
//   padding row: kx = D (a dump slot never read back), lx = D.   Diagonal rows carry FC_DIAG | (k<<16) | k
//   and take the (rare, half-wave-uniform) exp() path.
__device__ __forceinline__ uint32_t enc_row_code_fast(int64_t r, int T, int D) {
    const uint32_t cc = enc_row_code(r, T, D);
    if (cc == ROW_NONE) return ((uint32_t)D << 16) | (uint32_t)D;
    if (cc & ROW_LOC) return ((cc & 0xFFFFu) << 16) | (uint32_t)D;
    if ((cc >> 16) == (cc & 0xFFFFu)) return FC_DIAG | cc;
    return cc;
}

__host__ __device__ inline int ef_ys(int J) { return ((J + 63) / 64) * 64 + 4; }     // bytes per person row
__host__ __device__ inline size_t enc_fwd_fast_lds_floats(int D, int J) {
    const size_t r1a = (size_t)ENC_P * ef_ys(J) / 4;
    const size_t r1b = 2 * (size_t)ENC_P * enc_ds(D);
    return (size_t)ENC_P * EF_HS + (r1a > r1b ? r1a : r1b) + ENC_P + 2 * 4 * 32;
}

__global__ __launch_bounds__(ENC_THREADS, 2) void k_mvn_enc_fwd_fast(
    EncDims dm, const uint8_t* __restrict__ y, const int64_t* __restrict__ rows, int64_t gid0,
    const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ W21,
    const float* __restrict__ b21, const float* __restrict__ W22, const float* __restrict__ b22,
    const float* __restrict__ eps_in, uint64_t seed, uint32_t step, uint32_t stream,
    float* __restrict__ h_out, float* __restrict__ x_out, float* __restrict__ eps_out,
    float* __restrict__ ldT, float* __restrict__ ent_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int H = 64;
    const int D = dm.D, J = dm.J, DS = dm.DS, T = dm.T;
    const int YS = ef_ys(J);
    float* h_lds = smem;                                    // [P][65]
    float* R1 = h_lds + ENC_P * EF_HS;                      // phase A: response bytes; phase B: eps | x
    const size_t r1a = (size_t)ENC_P * YS / 4, r1b = 2 * (size_t)ENC_P * DS;
    float* ent_lds = R1 + (r1a > r1b ? r1a : r1b);          // [P]
    uint32_t* codes_all = (uint32_t*)(ent_lds + ENC_P);     // [4][32]
    float* bias_all = (float*)(codes_all + 4 * 32);         // [4][32]
    int8_t* Yi = (int8_t*)R1;
    float* eps_lds = R1;
    float* x_lds = R1 + ENC_P * DS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int64_t i0 = (int64_t)blockIdx.x * ENC_P;

    // ---------------------------------------------------------------- stage the response tile (bytes)
    {
        const int YW = YS / 4, JW = J / 4;
        uint32_t* Yw = (uint32_t*)R1;
        for (int base = 0; base < ENC_P * YW; base += ENC_THREADS * 8) {
            uint32_t v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int idx = base + q * ENC_THREADS + tid;
                v[q] = 0u;
                if (idx < ENC_P * YW) {
                    const int p = idx / YW, wq = idx - p * YW;
                    const int64_t i = i0 + p;
                    if (wq < JW && i < dm.nb) {
                        const int64_t row = rows ? rows[i] : i;
                        v[q] = *(const uint32_t*)(y + row * J + 4 * wq);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int idx = base + q * ENC_THREADS + tid;
                if (idx < ENC_P * YW) Yw[idx] = v[q];
            }
        }
    }
    __syncthreads();
    // ---------------------------------------------------------------- phase A: fc1 (+ softplus)
    {
        const int u = wave & 1, ht = wave >> 1;
        const int hh_row = 32 * ht + l31;
        const int p = 32 * u + l31;
        f32x16 acc = zero16();
        const int nchunk = (J + 63) / 64;
        auto loadA = [&](float4 (&A)[8], int c) {
            const int j0 = c * 64 + half * 32;
            const float* src = W1 + (int64_t)hh_row * J + j0;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                A[q] = (j0 + 4 * q + 4 <= J) ? *(const float4*)(src + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        };
        auto compute = [&](const float4 (&A)[8], int c) {
            const int8_t* yp = Yi + p * YS + c * 64 + half * 32;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int w = *(const int*)(yp + 4 * q);
                acc = mfma32(A[q].x, (float)((w << 24) >> 24), acc);
                acc = mfma32(A[q].y, (float)((w << 16) >> 24), acc);
                acc = mfma32(A[q].z, (float)((w << 8) >> 24), acc);
                acc = mfma32(A[q].w, (float)(w >> 24), acc);
            }
        };
        float4 A0[8], A1[8];
        loadA(A0, 0);
        for (int c = 0; c < nchunk; c += 2) {
            if (c + 1 < nchunk) loadA(A1, c + 1);
            compute(A0, c);
            if (c + 2 < nchunk) loadA(A0, c + 2);
            if (c + 1 < nchunk) compute(A1, c + 1);
        }
        const int64_t i = i0 + p;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int hh0 = 32 * ht + 8 * g + 4 * half;
            const float4 bb = *(const float4*)(b1 + hh0);
            float4 hv;
            hv.x = softplusf_(acc[4 * g + 0] + bb.x);                    // vi.py:449
            hv.y = softplusf_(acc[4 * g + 1] + bb.y);
            hv.z = softplusf_(acc[4 * g + 2] + bb.z);
            hv.w = softplusf_(acc[4 * g + 3] + bb.w);
            h_lds[p * EF_HS + hh0 + 0] = hv.x;
            h_lds[p * EF_HS + hh0 + 1] = hv.y;
            h_lds[p * EF_HS + hh0 + 2] = hv.z;
            h_lds[p * EF_HS + hh0 + 3] = hv.w;
            if (i < dm.nb) *(float4*)(h_out + i * H + hh0) = hv;
        }
    }
    __syncthreads();                                         // h complete; response bytes no longer needed
    // ---------------------------------------------------------------- eps, x := 0
    {
        const int nblk = (D + 3) >> 2;
        for (int e = tid; e < ENC_P * nblk; e += ENC_THREADS) {
            const int p = e / nblk, blk = e - p * nblk;
            const int64_t i = i0 + p;
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            if (i < dm.nb) {
                if (eps_in) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (4 * blk + q < D) z[q] = eps_in[i * D + 4 * blk + q];
                } else {
                    const int64_t row = rows ? rows[i] : i;
                    z = philox_normal4(seed, step, stream, gid0 + row, (uint32_t)blk);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (4 * blk + q < D) {
                    eps_lds[p * DS + 4 * blk + q] = z[q];
                    if (i < dm.nb) eps_out[i * D + 4 * blk + q] = z[q];
                }
        }
        for (int e = tid; e < ENC_P * DS; e += ENC_THREADS) x_lds[e] = 0.f;
        if (tid < ENC_P) { ent_lds[tid] = 0.f; eps_lds[tid * DS + D] = 1.0f; }     // slot D: the "times one" of loc rows
    }
    __syncthreads();
    // ---------------------------------------------------------------- phase B: head rows, per wave
    {
        uint32_t* codes = codes_all + 32 * wave;
        float* biasw = bias_all + 32 * wave;
        const int64_t RT = (int64_t)T + D;
        const int n_rt = (int)((RT + 31) / 32);
        auto prefetch = [&](float4 (&A)[8], uint32_t& code, float& bias, int tt) {
            const int64_t r = (int64_t)tt * 32 + l31;
            const float* src = (r < T) ? W22 + r * H : (r < RT ? W21 + (r - T) * H : nullptr);
            code = enc_row_code_fast(r, T, D);
            bias = (r < T) ? b22[r] : (r < RT ? b21[r - T] : 0.f);
#pragma unroll
            for (int q = 0; q < 8; ++q)
                A[q] = src ? *(const float4*)(src + half * 32 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        };
        auto tile = [&](const float4 (&A)[8], uint32_t code, float bias) {
            if (half == 0) { codes[l31] = code; biasw[l31] = bias; }
            __builtin_amdgcn_wave_barrier();
            f32x16 a0 = zero16(), a1 = zero16();
            const float* bp0 = h_lds + l31 * EF_HS + half * 32;
            const float* bp1 = h_lds + (32 + l31) * EF_HS + half * 32;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                a0 = mfma32(A[q].x, bp0[4 * q + 0], a0); a1 = mfma32(A[q].x, bp1[4 * q + 0], a1);
                a0 = mfma32(A[q].y, bp0[4 * q + 1], a0); a1 = mfma32(A[q].y, bp1[4 * q + 1], a1);
                a0 = mfma32(A[q].z, bp0[4 * q + 2], a0); a1 = mfma32(A[q].z, bp1[4 * q + 2], a1);
                a0 = mfma32(A[q].w, bp0[4 * q + 3], a0); a1 = mfma32(A[q].w, bp1[4 * q + 3], a1);
            }
#pragma unroll
            for (int uu = 0; uu < 2; ++uu) {
                const int p = 32 * uu + l31;
                const int64_t i = i0 + p;
                const float* ep = eps_lds + p * DS;
                float* xp = x_lds + p * DS;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int rl = 8 * g + 4 * half;
                    const uint4 cd = *(const uint4*)(codes + rl);
                    const float4 bs = *(const float4*)(biasw + rl);
                    const uint32_t cdv[4] = {cd.x, cd.y, cd.z, cd.w};
                    const float bsv[4] = {bs.x, bs.y, bs.z, bs.w};
                    // LDS float atomics are expensive (~100+ cycles each): merge the rows of a group that
                    // share kx (the common case: four consecutive entries of one tril row) into one add
                    int curk = -1;
                    float part = 0.f;
#pragma unroll
                    for (int jx = 0; jx < 4; ++jx) {
                        const uint32_t cc = cdv[jx];
                        const float v = (uu == 0 ? a0[4 * g + jx] : a1[4 * g + jx]) + bsv[jx];
                        const int k = (int)((cc >> 16) & 0x7FFFu);
                        float contrib;
                        if (cc & FC_DIAG) {                                           // half-wave uniform, rare
                            const float ld = expf(v);                                 // exp(diag M): vi.py:686
                            contrib = ld * ep[k];
                            atomicAdd(&ent_lds[p], v);
                            if (i < dm.nb) ldT[(int64_t)k * dm.nb + i] = ld;
                        } else {
                            contrib = v * ep[cc & 0xFFFFu];
                        }
                        if (k != curk) {
                            if (curk >= 0) atomicAdd(&xp[curk], part);
                            curk = k;
                            part = contrib;
                        } else {
                            part += contrib;
                        }
                    }
                    atomicAdd(&xp[curk], part);
                }
            }
            __builtin_amdgcn_wave_barrier();
        };
        float4 A0[8], A1[8];
        uint32_t c0 = ROW_NONE, c1 = ROW_NONE;
        float bz0 = 0.f, bz1 = 0.f;
        int tt = wave;
        if (tt < n_rt) prefetch(A0, c0, bz0, tt);
        for (; tt < n_rt; tt += 8) {
            if (tt + 4 < n_rt) prefetch(A1, c1, bz1, tt + 4);
            tile(A0, c0, bz0);
            if (tt + 8 < n_rt) prefetch(A0, c0, bz0, tt + 8);
            if (tt + 4 < n_rt) tile(A1, c1, bz1);
        }
    }
    __syncthreads();
    // ---------------------------------------------------------------- phase C: write x, entropy part
    for (int e = tid; e < ENC_P * D; e += ENC_THREADS) {
        const int p = e / D, k = e - p * D;
        const int64_t i = i0 + p;
        if (i < dm.nb) x_out[i * D + k] = x_lds[p * DS + k];
    }
    if (tid < ENC_P) {
        const int64_t i = i0 + tid;
        if (i < dm.nb) {
            float s = 0.f;
            for (int k = 0; k < D; ++k) { const float e = eps_lds[tid * DS + k]; s += e * e; }
            ent_out[i] = 0.5f * s + ent_lds[tid];
        }
    }
}
