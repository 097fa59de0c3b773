// fc1 weight gradient on the 16-bit MFMA, from dimension-major operands (full batch, no row gather): the same result and
// slab format as k_fc1_bwd_t (k_mvn_bwd_t.hip; autograd of vi.py:417-455's fc1),
//     GW1[hh][j] = sum_p ghpreT[hh][p] yin[p][j],   Gb1[hh] = sum_p ghpreT[hh][p]         (yin = int8 -1 / 0 / 1)
// MFMA 32x32x16: C rows = hidden units, columns = items, contraction index = persons (16 per k-step).
//   B = the response bytes of 8 consecutive persons of an item row of yT: exact in bf16 and in fp16, converted in registers;
//   A = 8 consecutive persons of a ghpreT row, split in registers:
//       F16 (the step's largest |ghpre| is known: the hidden-gradient kernel collected it, maxw[3]): two fp16 terms of
//       ghpre 2^s -> 2 products per k-step, the power of two taken off the accumulators at the end;
//       otherwise (the 1-D encoder's caller has no such maximum): three bf16 terms -> 3 products.
//   The bias gradient is the column of a virtual item J whose responses are all 1.
// Operands are staged through LDS (round 4).  The form before it (k_fc1_bwd_b, retired in round 5) read them global ->
// registers with every lane on a row of its own: 8 load instructions a k-step and wave, each touching 64 different cache
// lines.  That is what it waited for (0.36 ms for 0.9 GB: 17 % of the matrix pipe, 2.5 TB/s; the same with two waves a SIMD)
// -- the texture path takes a line a cycle (docs/HARDWARE.md, rule 33).  Here a workgroup moves the operands of 64 persons as
// whole 1 KB transfers (LDS-DMA, rows of 256 / 64 contiguous bytes: 16 KB of ghpreT, 32 KB of yT), three chunks deep (a
// wave's transfers a chunk are always the same number: the wait for a chunk is counted and leaves the next one's in flight),
// and the waves read their fragments from LDS.
//   G tile [64 rows][16 chunks of 16 B] (64 persons fp32): chunk c of row r at position c ^ (r & 15)
//   Y tile [512 rows][4 chunks of 16 B] (64 persons, bytes): chunk c of row r at position c ^ ((r >> 2) & 3)
// (the swizzles are applied to the per-lane SOURCE address of a transfer: the LDS side of a transfer is lane-linear)
// Measured (the 1M x 500 step, same box): 0.354-0.361 ms for the register form, 0.232-0.244 here -- and the same for four
// waves of four item tiles or eight of two, two chunks deep or three, the eight-instruction split of ghpre or the
// four-instruction one: what is left is the 0.9 GB themselves, read as 64-byte pieces of 500 rows a megabyte apart (3.8 TB/s).
// A workgroup's persons are a whole number of 64-person chunks (the last workgroup takes the ragged end).
// The ragged last chunk (nb % 64 != 0; nb % 4 == 0 and ystride % 16 == 0 are the caller's contract): BOTH tiles are cleared
// before the transfers that are still inside the batch are issued, so an absent person is ghpre = 0 AND response byte 0 -- no
// product of the chunk depends on bytes that were not written this chunk.
// (included by vx_abi.hip after k_mvn_fwd_b.hip)
#define F1C_PC 64                                                       // persons a chunk (four k-steps)
#define F1C_GBYTES (64 * F1C_PC * 4)                                    // 16 384
#define F1C_YBYTES (512 * F1C_PC)                                       // 32 768
#define F1C_BUF (F1C_GBYTES + F1C_YBYTES)
#define F1C_NBUF 3                                                      // chunks in LDS: one in the MFMAs, two in flight
__host__ __device__ inline size_t f1c_lds_bytes() { return F1C_NBUF * (size_t)F1C_BUF; }

#ifndef F1C_NT
#define F1C_NT 2                                                        // item tiles a wave; 16 / F1C_NT waves a workgroup (512 items)
#endif
#define F1C_WAVES (16 / F1C_NT)
#define F1C_THREADS (64 * F1C_WAVES)
// TILED: the responses come from a TILE-MAJOR copy, [chunk of 64 persons][J][64 bytes] (ystride unused): a workgroup's 512 item
// rows of a chunk are one contiguous 32 KB instead of 64-byte pieces of rows a whole batch apart.
template <bool F16, bool TILED = false>
__global__ __launch_bounds__(F1C_THREADS, 1) void k_fc1_bwd_c(
    EncDims dm, const uint8_t* __restrict__ yT, int64_t ystride, const float* __restrict__ ghpreT,
    float* __restrict__ slabs, int64_t slab_len, const uint32_t* __restrict__ maxw = nullptr /*F16: float bits, [3] = max |ghpre|*/) {
    extern __shared__ __attribute__((aligned(16))) char smem_f1[];
    typedef uint32_t u32x4w __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x2w __attribute__((ext_vector_type(2)));
    constexpr int NT = F1C_NT, NW = F1C_WAVES;                          // item tiles a wave, waves: NW x NT x 32 = 512 items a workgroup
    constexpr int NG = 16 / NW, NY = 32 / NW, NOWN = NG + NY;          // transfers a wave and chunk: G tile, Y tile, both
    static_assert(NOWN <= 15, "counted wait: vmcnt's low four bits");
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

    // ---- transfers of a chunk: 16 of the G tile (4 rows each), 32 of the Y tile (16 rows each); wave w moves d = w + NW u
    const char* gsrc[NG];                                               // per-lane source at person 0 of the chunk
    const char* ysrc[NY];
#pragma unroll
    for (int u = 0; u < NG; ++u) {
        const int d = wave + NW * u, row = 4 * d + (lane >> 4), c = (lane & 15) ^ (row & 15);
        gsrc[u] = (const char*)(ghpreT + (int64_t)row * nb + 4 * c);
    }
#pragma unroll
    for (int u = 0; u < NY; ++u) {
        const int d = wave + NW * u, row = 16 * d + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
        // (rows of the bias column and past it are never read as data: their transfers re-read row 0, so that every wave
        // issues the same number of transfers a chunk)
        ysrc[u] = (const char*)(yT + (int64_t)(jb0 + row < J ? jb0 + row : 0) * (TILED ? (int64_t)F1C_PC : ystride) + 16 * c);
    }
    auto stage = [&](int64_t ch, int b) __attribute__((always_inline)) {
        const int64_t p0 = ch * F1C_PC;
        const uint32_t lb = lds_addr_uniform(smem_f1 + b * F1C_BUF) + (uint32_t)wave * 1024u;
        const bool whole = p0 + F1C_PC <= nb;                           // block-uniform
        if (!whole) {                                                   // the ragged last chunk: absent persons are zeros of both tiles
            for (int e = tid; e < F1C_BUF / 16; e += F1C_THREADS) *(f32x4*)(smem_f1 + b * F1C_BUF + 16 * e) = f32x4{0.f, 0.f, 0.f, 0.f};
            __syncthreads();
        }
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            const int row = 4 * (wave + NW * u) + (lane >> 4), c = (lane & 15) ^ (row & 15);
            if (whole || p0 + 4 * c + 4 <= nb) dma16(gsrc[u] + p0 * 4, lb + (uint32_t)u * (1024u * NW));      // nb % 4 == 0
        }
#pragma unroll
        for (int u = 0; u < NY; ++u) {                                  // (a yT row is ystride bytes long: nothing is read past it)
            const int row = 16 * (wave + NW * u) + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
            if constexpr (TILED) dma16(ysrc[u] + ch * (int64_t)J * F1C_PC, lb + (uint32_t)F1C_GBYTES + (uint32_t)u * (1024u * NW));
            else if (whole || p0 + 16 * c + 16 <= ystride) dma16(ysrc[u] + p0, lb + (uint32_t)F1C_GBYTES + (uint32_t)u * (1024u * NW));
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
            if constexpr (F16) split2h_frag_mix(v, g_scale, ah[ht][0], ah[ht][1]);
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
        // whole(ch): the chunk's transfers are the full NOWN a wave (the ragged last chunk skips some)
        auto whole = [&](int64_t ch) -> bool { return (ch + 1) * F1C_PC <= nb; };
        stage(ch0, 0);
        if (ch0 + 1 < ch1) stage(ch0 + 1, 1);
        if (ch0 + 1 < ch1 && whole(ch0 + 1)) __builtin_amdgcn_s_waitcnt(0x0F70 | NOWN); else vx_wait_vmem();
        __syncthreads();
        int b = 0;
        for (int64_t ch = ch0; ch < ch1; ++ch) {
            const int b2 = b + 2 >= F1C_NBUF ? b + 2 - F1C_NBUF : b + 2;
            if (ch + 2 < ch1) stage(ch + 2, b2);                       // (its buffer was released by the barrier of the chunk before)
            Ops o0, o1;
            read_ops(o0, b, 0);
            read_ops(o1, b, 1);
            compute(o0);
            read_ops(o0, b, 2);
            compute(o1);
            read_ops(o1, b, 3);
            compute(o0);
            compute(o1);
            // this wave's transfers of the next chunk (those of the one after it may stay in flight: loads complete in order)
            if (ch + 2 < ch1 && whole(ch + 2)) __builtin_amdgcn_s_waitcnt(0x0F70 | NOWN); else vx_wait_vmem();
            __syncthreads();                                           // ... everybody's; and this chunk's buffer is free
            b = b + 1 >= F1C_NBUF ? 0 : b + 1;
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
