// fc1 weight gradient, operands staged through LDS (round 4): the same mathematics, arguments and slab format as
// k_fc1_bwd_b (k_fc1_bwd_b.hip),
//     GW1[hh][j] = sum_p ghpreT[hh][p] yin[p][j],   Gb1[hh] = sum_p ghpreT[hh][p]         (yin = int8 -1 / 0 / 1)
// k_fc1_bwd_b reads its operands global -> registers with every lane on a row of its own: 8 load instructions a k-step
// and wave, each touching 64 different cache lines.  That is what it waits for (0.36 ms for 0.9 GB: 17 % of the matrix pipe,
// 2.5 TB/s; the same with two waves a SIMD) -- the texture path takes a line a cycle.  Here a workgroup moves the operands
// of 64 persons as whole 1 KB transfers (LDS-DMA, rows of 256 / 64 contiguous bytes: 16 KB of ghpreT, 32 KB of yT), double
// buffered, and the four waves read their fragments from LDS: the per-k-step arithmetic (split of ghpre, response bytes
// to fp16 / bf16, MFMAs, their order) is k_fc1_bwd_b's.
//   G tile [64 rows][16 chunks of 16 B] (64 persons fp32): chunk c of row r at position c ^ (r & 15)
//   Y tile [512 rows][4 chunks of 16 B] (64 persons, bytes): chunk c of row r at position c ^ ((r >> 2) & 3)
// (the swizzles are applied to the per-lane SOURCE address of a transfer: the LDS side of a transfer is lane-linear)
// A workgroup's persons are a whole number of 64-person chunks (the last workgroup takes the ragged end): the slabs are
// cut at other places than k_fc1_bwd_b's, so the two kernels' results differ in the last bits (summation order).
// (included by vx_abi.hip after k_fc1_bwd_b.hip)
#define F1C_PC 64                                                       // persons a chunk (four k-steps)
#define F1C_GBYTES (64 * F1C_PC * 4)                                    // 16 384
#define F1C_YBYTES (512 * F1C_PC)                                       // 32 768
#define F1C_BUF (F1C_GBYTES + F1C_YBYTES)
__host__ __device__ inline size_t f1c_lds_bytes() { return 2 * (size_t)F1C_BUF; }

template <bool F16>
__global__ __launch_bounds__(256, 1) void k_fc1_bwd_c(
    EncDims dm, const uint8_t* __restrict__ yT, int64_t ystride, const float* __restrict__ ghpreT,
    float* __restrict__ slabs, int64_t slab_len, const uint32_t* __restrict__ maxw = nullptr /*F16: float bits, [3] = max |ghpre|*/) {
    extern __shared__ __attribute__((aligned(16))) char smem_f1[];
    typedef uint32_t u32x4w __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x2w __attribute__((ext_vector_type(2)));
    constexpr int NT = 4;                                               // item tiles a wave: 4 waves x 4 x 32 = 512 items a workgroup
    const int J = dm.J;
    const int64_t nb = dm.nb;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int jb0 = blockIdx.x * 512;                                   // first item of the workgroup (item J = the bias column)
    const int jw0 = jb0 + 32 * NT * wave;
    int kind[NT];                                                       // 0: response row, 1: ones (the bias column), 2: nothing
    uint32_t yoff[NT];                                                  // byte offset of this lane's row in a Y tile (+ 8 half)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int j = jw0 + 32 * t + l31;
        kind[t] = j < J ? 0 : (j == J ? 1 : 2);
        yoff[t] = (uint32_t)((j - jb0) * F1C_PC + 8 * half);
    }
    const uint32_t ysw = (uint32_t)(((jw0 - jb0 + l31) >> 2) & 3);      // (32 t is a multiple of 16: the swizzle of a lane's rows is one value)
    float g_scale = 1.0f, g_inv = 1.0f;
    if constexpr (F16) {
        const int e = f16_scale_exp(__builtin_bit_cast(float, maxw[3]));
        g_scale = ldexpf(1.0f, e);
        g_inv = ldexpf(1.0f, -e);
    }
    f32x16 acc[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) { acc[t][0] = zero16(); acc[t][1] = zero16(); }

    // ---- the workgroup's persons: whole chunks
    const int64_t n_ch = (nb + F1C_PC - 1) / F1C_PC, per = (n_ch + gridDim.y - 1) / gridDim.y;
    const int64_t ch0 = (int64_t)blockIdx.y * per, ch1 = (ch0 + per < n_ch) ? ch0 + per : n_ch;

    // ---- transfers of a chunk: 16 of the G tile (4 rows each), 32 of the Y tile (16 rows each); wave w moves d = w + 4 u
    const char* gsrc[4];                                                // per-lane source at person 0 of the chunk
    const char* ysrc[8];
    bool ylive[8];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int d = wave + 4 * u, row = 4 * d + (lane >> 4), c = (lane & 15) ^ (row & 15);
        gsrc[u] = (const char*)(ghpreT + (int64_t)row * nb + 4 * c);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int d = wave + 4 * u, row = 16 * d + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
        ylive[u] = jb0 + row < J;                                       // rows of the bias column and past it are never read as data
        ysrc[u] = (const char*)(yT + (int64_t)(jb0 + row < J ? jb0 + row : 0) * ystride + 16 * c);
    }
    auto stage = [&](int64_t ch, int b) __attribute__((always_inline)) {
        const int64_t p0 = ch * F1C_PC;
        const uint32_t lb = lds_addr_uniform(smem_f1 + b * F1C_BUF) + (uint32_t)wave * 1024u;
        const bool whole = p0 + F1C_PC <= nb;                           // block-uniform
        if (!whole) {                                                   // the ragged last chunk: absent persons are zeros of ghpre
            for (int e = tid; e < F1C_GBYTES / 16; e += 256) *(f32x4*)(smem_f1 + b * F1C_BUF + 16 * e) = f32x4{0.f, 0.f, 0.f, 0.f};
            __syncthreads();
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = 4 * (wave + 4 * u) + (lane >> 4), c = (lane & 15) ^ (row & 15);
            if (whole || p0 + 4 * c + 4 <= nb) dma16(gsrc[u] + p0 * 4, lb + (uint32_t)u * 4096u);      // nb % 4 == 0
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {                                   // (a yT row is ystride bytes long: nothing is read past it)
            const int row = 16 * (wave + 4 * u) + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
            if (ylive[u] && (whole || p0 + 16 * c + 16 <= ystride)) dma16(ysrc[u] + p0, lb + (uint32_t)F1C_GBYTES + (uint32_t)u * 4096u);
        }
    };

    struct Ops { f32x4 g[2][2]; u32x2w y[NT]; };
    // operands of k-step ks (0..3) of the chunk in buffer b
    auto read_ops = [&](Ops& o, int b, int ks) __attribute__((always_inline)) {
        const char* gb = smem_f1 + b * F1C_BUF;
        const char* yb = gb + F1C_GBYTES;
#pragma unroll
        for (int ht = 0; ht < 2; ++ht) {
            const int row = 32 * ht + l31;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int c = 4 * ks + 2 * half + q;
                o.g[ht][q] = *(const f32x4*)(gb + row * 256 + ((c ^ (row & 15)) << 4));
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) o.y[t] = *(const u32x2w*)(yb + yoff[t] + (((uint32_t)ks ^ ysw) << 4));
    };
    auto compute = [&](const Ops& o) __attribute__((always_inline)) {   // (k_fc1_bwd_b's k-step)
        bf16x8 a[2][3];
        f16x8 ah[2][2];
#pragma unroll
        for (int ht = 0; ht < 2; ++ht) {
            const float v[8] = {o.g[ht][0][0], o.g[ht][0][1], o.g[ht][0][2], o.g[ht][0][3],
                                o.g[ht][1][0], o.g[ht][1][1], o.g[ht][1][2], o.g[ht][1][3]};
            if constexpr (F16) split2h_frag(v, g_scale, ah[ht][0], ah[ht][1]);
            else fb_split8(v, a[ht][0], a[ht][1], a[ht][2]);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            u32x4w q;
            constexpr uint32_t ONE = F16 ? 0x3C00u : 0x3F80u;          // 1.0 in fp16 / bf16
#pragma unroll
            for (int d = 0; d < 4; ++d) {                              // byte b in {0, 1, 255} -> {0, 1, -1}
                const uint32_t src = o.y[t][d >> 1];
                const uint32_t w = (d & 1) ? __builtin_amdgcn_perm(0u, src, 0x0c030c02u) : __builtin_amdgcn_perm(0u, src, 0x0c010c00u);
                q[d] = (w & 0x00010001u) * ONE | ((w & 0x00800080u) << 8);
                if (kind[t] == 1) q[d] = ONE * 0x00010001u;
                if (kind[t] == 2) q[d] = 0u;
            }
#pragma unroll
            for (int ht = 0; ht < 2; ++ht) {
                if constexpr (F16) {
                    const f16x8 yh = __builtin_bit_cast(f16x8, q);
                    acc[t][ht] = mfma_f16(ah[ht][1], yh, acc[t][ht]);
                    acc[t][ht] = mfma_f16(ah[ht][0], yh, acc[t][ht]);
                } else {
                    const bf16x8 yb = __builtin_bit_cast(bf16x8, q);
                    acc[t][ht] = mfma_bf16(a[ht][2], yb, acc[t][ht]);
                    acc[t][ht] = mfma_bf16(a[ht][1], yb, acc[t][ht]);
                    acc[t][ht] = mfma_bf16(a[ht][0], yb, acc[t][ht]);
                }
            }
        }
    };

    if (ch0 < ch1) {
        stage(ch0, 0);
        vx_wait_vmem();
        __syncthreads();
        int b = 0;
        for (int64_t ch = ch0; ch < ch1; ++ch) {
            if (ch + 1 < ch1) stage(ch + 1, b ^ 1);                    // (its buffer was released by the barrier below)
            Ops o0, o1;
            read_ops(o0, b, 0);
            read_ops(o1, b, 1);
            compute(o0);
            read_ops(o0, b, 2);
            compute(o1);
            read_ops(o1, b, 3);
            compute(o0);
            compute(o1);
            vx_wait_vmem();                                            // this wave's transfers of the next chunk
            __syncthreads();                                           // ... everybody's; and this chunk's buffer is free
            b ^= 1;
        }
    }
    // slab: [W1-grad: 64 * J | b1-grad: 64]
    float* slab = slabs + (int64_t)blockIdx.y * slab_len;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int j = jw0 + 32 * t + l31;
        if (j <= J) {
#pragma unroll
            for (int ht = 0; ht < 2; ++ht)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int hh = 32 * ht + crow32(r, half);
                    if (j < J) slab[(int64_t)hh * J + j] = acc[t][ht][r] * g_inv;
                    else slab[(int64_t)64 * J + hh] = acc[t][ht][r] * g_inv;
                }
        }
    }
}
