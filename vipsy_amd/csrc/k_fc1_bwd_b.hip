// fc1 weight gradient on the 16-bit MFMA, from dimension-major operands (full batch, no row gather): the same result and
// slab format as k_fc1_bwd_t (k_mvn_bwd_t.hip),
//     GW1[hh][j] = sum_p ghpreT[hh][p] yin[p][j],   Gb1[hh] = sum_p ghpreT[hh][p]         (yin = int8 -1 / 0 / 1)
// MFMA 32x32x16: C rows = hidden units, columns = items, contraction index = persons (16 per k-step).
//   B = the response bytes of 8 consecutive persons of an item row of yT: exact in bf16 and in fp16, converted in registers;
//   A = 8 consecutive persons of a ghpreT row, split in registers:
//       F16 (round 3; the step's largest |ghpre| is known: the hidden-gradient kernel collected it, maxw[3]): two fp16 terms
//       of ghpre 2^s -> 2 products per k-step, the power of two taken off the accumulators at the end;
//       otherwise (the 1-D encoder's caller has no such maximum): three bf16 terms -> 3 products.
//   The bias gradient is the column of a virtual item J whose responses are all 1.
// A wave owns 4 item tiles x both hidden tiles (128 accumulator registers); operands go global -> registers one
// k-step ahead (consecutive k-steps of a lane continue in the same cache lines); no LDS.
// (included by vx_abi.hip after k_mvn_fwd_b.hip)

// F1B_NT item tiles a wave, 512 / (32 F1B_NT) waves a workgroup (512 items either way).  Four tiles and four waves; two tiles
// and eight waves (64 accumulator registers, two waves per SIMD) were slower beside k_mvn_enc_bwd_w_b (the pair 3.04 instead of
// 2.82 ms): the split of ghpre is then shared by half as many MFMAs.
#ifndef F1B_NT
#define F1B_NT 4
#endif
#define F1B_THREADS (64 * (16 / F1B_NT))
#define F1B_NS 6                                                     // operand stages in registers (k-steps of look-ahead + 1)

template <bool F16>
__global__ __launch_bounds__(F1B_THREADS, 1) void k_fc1_bwd_b(
    EncDims dm, const uint8_t* __restrict__ yT, int64_t ystride, const float* __restrict__ ghpreT,
    float* __restrict__ slabs, int64_t slab_len, const uint32_t* __restrict__ maxw = nullptr /*F16: float bits, [3] = max |ghpre|*/) {
    typedef uint32_t u32x4w __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x2w __attribute__((ext_vector_type(2)));
    const int J = dm.J;
    const int64_t nb = dm.nb;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int jw0 = blockIdx.x * 512 + 32 * F1B_NT * wave;             // items of this wave: jw0 .. jw0 + 32 F1B_NT - 1 (item J = bias)
    if (jw0 > J) return;                                               // waves share nothing
    const uint8_t* yrow[F1B_NT];
    int kind[F1B_NT];                                                       // 0: response row, 1: ones (the bias column), 2: nothing
#pragma unroll
    for (int t = 0; t < F1B_NT; ++t) {
        const int j = jw0 + 32 * t + l31;
        kind[t] = j < J ? 0 : (j == J ? 1 : 2);
        yrow[t] = yT + (int64_t)(j < J ? j : 0) * ystride + 8 * half;
    }
    const float* grow[2] = {ghpreT + (int64_t)l31 * nb + 8 * half, ghpreT + (int64_t)(32 + l31) * nb + 8 * half};

    float g_scale = 1.0f, g_inv = 1.0f;
    if constexpr (F16) {
        const int e = f16_scale_exp(__builtin_bit_cast(float, maxw[3]));
        g_scale = ldexpf(1.0f, e);
        g_inv = ldexpf(1.0f, -e);
    }
    f32x16 acc[F1B_NT][2];
#pragma unroll
    for (int t = 0; t < F1B_NT; ++t) { acc[t][0] = zero16(); acc[t][1] = zero16(); }

    struct Ops { f32x4 g[2][2]; u32x2w y[F1B_NT]; };
    auto load = [&](Ops& o, int64_t ks) __attribute__((always_inline)) {
        const int64_t p0 = 16 * ks;                                    // this lane: persons p0 + 8 half + 0..7
#pragma unroll
        for (int ht = 0; ht < 2; ++ht)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (p0 + 8 * half + 4 * q + 4 <= nb) v = *(const f32x4*)(grow[ht] + p0 + 4 * q);   // nb % 4 == 0
                o.g[ht][q] = v;
            }
#pragma unroll
        for (int t = 0; t < F1B_NT; ++t) o.y[t] = *(const u32x2w*)(yrow[t] + p0);     // ystride % 16 == 0, >= nb rounded to 16
    };
    auto compute = [&](const Ops& o) __attribute__((always_inline)) {
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
        for (int t = 0; t < F1B_NT; ++t) {
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
    // a workgroup walks a CONTIGUOUS range of k-steps: consecutive loads of a lane continue in the same cache lines
    const int64_t n_ks = (nb + 15) / 16, per = (n_ks + gridDim.y - 1) / gridDim.y;
    int64_t ks = (int64_t)blockIdx.y * per;
    const int64_t ks_end = (ks + per < n_ks) ? ks + per : n_ks;
    // operands run F1B_NS k-steps ahead of the MFMAs: the loads come from HBM (a wave streams its own rows of yT and
    // ghpreT), one k-step is ~770 cycles of matrix time, so a single stage of look-ahead left the matrix pipe waiting
    // (busy 0.21)
    if (ks < ks_end) {
        Ops st[F1B_NS];
#pragma unroll
        for (int u = 0; u < F1B_NS - 1; ++u)
            if (ks + u < ks_end) load(st[u], ks + u);
        while (ks < ks_end) {
#pragma unroll
            for (int u = 0; u < F1B_NS; ++u) {
                if (ks + F1B_NS - 1 < ks_end) load(st[(u + F1B_NS - 1) % F1B_NS], ks + F1B_NS - 1);
                if (ks < ks_end) compute(st[u]);
                ++ks;
            }
        }
    }
    // slab: [W1-grad: 64 * J | b1-grad: 64]
    float* slab = slabs + (int64_t)blockIdx.y * slab_len;
#pragma unroll
    for (int t = 0; t < F1B_NT; ++t) {
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
