// Common device helpers for the vipsy_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vipsy_amd.h"

#define VX_WAVE 64
#define VX_EPS32 1.1920928955078125e-07f          // torch clamp_probs epsilon (float32)
#define VX_LOGP_MISSING (-1.1920928244535389e-07f) // log Bern(0 | clamp(0)) = -log1p(eps) (vi.py:621-624)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// fp32 MFMA 32x32x2 (v_mfma_f32_32x32x2_f32): exact fp32 fma chain, 64 cycles / SIMD.
//   A: lane l holds A[m = l & 31][k = l >> 5]
//   B: lane l holds B[k = l >> 5][n = l & 31]
//   C: lane l, reg r holds C[m = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][n = l & 31]
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int crow32(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// ---------------------------------------------------------------------------------------------
// bf16 MFMA 32x32x16 (v_mfma_f32_32x32x16_bf16, 32 cycles / SIMD), fp32 accumulate:
//   A: lane l holds A[m = l & 31][k = 8 (l >> 5) + j], B: lane l holds B[k = 8 (l >> 5) + j][n = l & 31], j = 0..7
//   C: as the fp32 form above.
// bf16x3: an fp32 operand as three bf16 terms v = h + m + l (round-to-nearest at each stage, exact to 2^-24);
// six cross products (h h, h m, m h, h l, l h, m m) reproduce the fp32 product chain.
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// eight fp32 values -> three bf16 fragments (h, m, l)
__device__ __forceinline__ void split3_frag(const float (&v)[8], bf16x8& fh, bf16x8& fm, bf16x8& fl) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)v[j];
        const float r1 = v[j] - (float)h;
        const __bf16 m = (__bf16)r1;
        fh[j] = h; fm[j] = m; fl[j] = (__bf16)(r1 - (float)m);
    }
}

// ---------------------------------------------------------------------------------------------
// fp16 MFMA 32x32x16 (v_mfma_f32_32x32x16_f16, 32 cycles / SIMD, fp32 accumulate; fragment layout as the bf16 form).
// f16x2: an fp32 operand, scaled by a power of two into the fp16 range, as TWO fp16 terms v 2^s = hi + lo (round to
// nearest at both stages: 11 + 11 significand bits, |v 2^s - hi - lo| <= max(2^-22 |v 2^s|, 2^-25) -- fp16 subnormals are
// honoured by the MFMA, tools/f16_ubench.hip); THREE cross products (lo hi, hi lo, hi hi) reproduce the fp32 product
// chain (tools/sim16.py: 2.2e-7 of sum |a b| against 2.4e-7 for the fp32 chain itself) in half the matrix-pipe time of
// the six bf16 products.  The power of two comes off the fp32 accumulator (exact).
// ---------------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mfma_f16(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// two fp32 values (already scaled) -> packed fp16 heads and packed fp16 remainders: v_cvt_pk_f16_f32 (round to nearest),
// the two halves back as floats, the two exact remainders, one more packed conversion
__device__ __forceinline__ void split2h_pair(float a, float b, uint32_t& ph, uint32_t& pl) {
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f16x2 h = __builtin_convertvector((f32x2){a, b}, f16x2);
    float ra = a - (float)h[0], rb = b - (float)h[1];
    asm("" : "+v"(ra), "+v"(rb));                      // scalar subtractions: packed f32 adds cost more beside MFMAs
    ph = __builtin_bit_cast(uint32_t, h);
    pl = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){ra, rb}, f16x2));
}
// eight fp32 values times `scale` (a power of two) -> the two fp16 fragments
__device__ __forceinline__ void split2h_frag(const float (&v)[8], float scale, f16x8& fh, f16x8& fl) {
    typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));
    u32x4s ph, pl;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t a, b;
        split2h_pair(v[2 * q] * scale, v[2 * q + 1] * scale, a, b);
        ph[q] = a; pl[q] = b;
    }
    fh = __builtin_bit_cast(f16x8, ph);
    fl = __builtin_bit_cast(f16x8, pl);
}
// The same two fragments in FOUR instructions an element pair instead of eight: v_fma_mixlo_f16 / v_fma_mixhi_f16 compute an
// fp32 fma of operands that are fp32 or one half of a register read as fp16 and write the result, rounded to fp16, into one
// half of the destination -- head = rn16(v * scale + 0), remainder = rn16(v * scale - head).  Both fmas are exact in fp32 (a
// power-of-two scaling; the difference of a value and its own fp16 rounding), so every result is rounded once, to fp16, as in
// split2h_frag: the same bits, except that a value of -0 splits into (+0, -0) instead of (-0, +0).  `scale` must be
// wave-uniform (it is read from a scalar register).
__device__ __forceinline__ void split2h_frag_mix(const float (&v)[8], float scale, f16x8& fh, f16x8& fl) {
    typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));
    const float sc_ = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, scale)));
    // The hazard recogniser does not look into asm statements (docs/HARDWARE.md rule 31): a value that a TRANSCENDENTAL
    // instruction has just written (v_exp_f32 in k_hodina_m: the posterior weights) needs a wait state before a vector
    // instruction reads it, and nothing inserts it in front of the v_fma_mix below -- one build of k_hodina_m returned garbage
    // for ~3 % of the persons, the next one, with another schedule, did not.  The eight values pass through one s_nop that the
    // asm statements below depend on.
    float w[8] = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
    asm volatile("s_nop 0" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]));
    u32x4s ph, pl;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t h, l;
        asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(w[2 * q]), "s"(sc_));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(w[2 * q + 1]), "s"(sc_));
        asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(w[2 * q]), "s"(sc_), "v"(h));
        asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(w[2 * q + 1]), "s"(sc_), "v"(h));
        ph[q] = h; pl[q] = l;
    }
    // ... and out: a register that a vector instruction INSIDE an asm statement has just written needs two wait states before an
    // MFMA reads it as an operand (cdna_hip_programming.md section 5.7, item 2: "wrong values on some waves of some launches")
    asm volatile("s_nop 1" : "+v"(ph[0]), "+v"(ph[1]), "+v"(ph[2]), "+v"(ph[3]), "+v"(pl[0]), "+v"(pl[1]), "+v"(pl[2]), "+v"(pl[3]));
    fh = __builtin_bit_cast(f16x8, ph);
    fl = __builtin_bit_cast(f16x8, pl);
}
// one value for the image builders (host-rate code): bit patterns of the two terms
__device__ __forceinline__ void split2h_bits(float v, uint16_t& bh, uint16_t& bl) {
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)h);
    bh = __builtin_bit_cast(uint16_t, h);
    bl = __builtin_bit_cast(uint16_t, l);
}
// the power of two that brings |v| <= vmax under 2^15 (fp16 overflows at 65504): 2^(15 - e), vmax = m 2^e, 0.5 <= m < 1;
// vmax = 0 or not finite: 1
__host__ __device__ __forceinline__ int f16_scale_exp(float vmax) {
    if (!(vmax > 0.f) || !(vmax < 3.0e38f)) return 0;
    int e;
    (void)frexpf(vmax, &e);
    return 15 - e;
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (same spec as oracle/vi_oracle.py::philox4x32_10)
// ---------------------------------------------------------------------------------------------
struct u32x4 { uint32_t x, y, z, w; };

__host__ __device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                        uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        uint64_t p0 = (uint64_t)c0 * 0xD2511F53u;
        uint64_t p1 = (uint64_t)c2 * 0xCD9E8D57u;
        uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return u32x4{c0, c1, c2, c3};
}

__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * 5.9604644775390625e-08f; }

// four standard normals for (person gid, block) -- dims 4*block .. 4*block+3
__device__ __forceinline__ f32x4 philox_normal4(uint64_t seed, uint32_t step, uint32_t stream, int64_t gid,
                                                uint32_t block) {
    u32x4 w = philox4x32_10((uint32_t)gid, (uint32_t)((uint64_t)gid >> 32), step, (stream << 16) | block,
                            (uint32_t)seed, (uint32_t)(seed >> 32));
    // Box-Muller on the hardware transcendentals: v_sin/v_cos take the angle in revolutions (the uniform itself, no
    // range reduction), v_log is log2.  Absolute error of a normal ~1e-6, far inside the parity tolerance (2e-5);
    // the library sincosf/logf cost ~300 VALU issues per call, and VALU time adds to MFMA time on this chip.
    const float u0 = u01(w.x), u1 = u01(w.y), u2 = u01(w.z), u3 = u01(w.w);
    const float r0 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u0));   // sqrt(-2 ln u)
    const float r1 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u2));
    const float s0 = __builtin_amdgcn_sinf(u1), c0 = __builtin_amdgcn_cosf(u1);
    const float s1 = __builtin_amdgcn_sinf(u3), c1 = __builtin_amdgcn_cosf(u3);
    f32x4 o;
    o[0] = r0 * c0; o[1] = r0 * s0; o[2] = r1 * c1; o[3] = r1 * s1;
    return o;
}

// ---------------------------------------------------------------------------------------------
// scalar math
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
// softplus = max(x, 0) + log(1 + exp(-|x|)) on the hardware exp2 / log2 (absolute error <= 1e-7)
__device__ __forceinline__ float softplusf_(float x) {
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * fabsf(x));
    return fmaf(__builtin_amdgcn_logf(1.0f + e), 0.6931471805599453f, fmaxf(x, 0.f));
}

// One response cell: given z = Dc*(x.a + b) returns the log-lik term `lp`
// and dlp/dz; for 3PL/4PL also dlp/dc_un, dlp/dd_un.  y: 0/1/255(missing).
// 2PL: P = sigmoid(z) clamped to [eps, 1-eps]  ==  z clamped to +-logit(1-eps), zero gradient outside.
// 3PL/4PL: P = c + (d-c) s(z) and Q = 1-P = (1-d) + (d-c) s(-z) are both formed from positive terms
// (omd = 1-d = sigmoid(-d_un) comes from the leaf), so (y-P)/(P(1-P)) = y ? 1/P : -1/Q keeps full
// float32 accuracy where the reference's own float32 chain (sigmoid -> clamp -> log / log1p) loses it.
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// ln of a NORMAL positive float on the hardware log2 (one transcendental + one multiply; the library's __logf expands to ten
// instructions of denormal handling and an extended-precision multiply by ln 2 -- the clamped probabilities of the 3PL / 4PL
// cell lie in [eps32, 1 - eps32]).  Measured against double-precision log (tools/fast_log_check.hip, profiles/r06_fast_log_check.txt):
// relative error <= 1.44e-7 over (0, 1) and <= 1.36e-7 for 1 - 1e-3 <= x <= 1 - 1.2e-7, where log is small (absolute 1.1e-10) --
// the hardware log2 is accurate relative to its RESULT near 1, as the library call is (1.58e-7 there))
__device__ __forceinline__ float fast_log(float x) { return 0.6931471805599453f * __builtin_amdgcn_logf(x); }

template <int MODEL>
__device__ __forceinline__ void irt_cell(float z, unsigned y, float c, float d, float omd, float& lp, float& dz,
                                         float& dc, float& dd) {
    // y >= 254: missing cell (255) or a cell outside the problem (254): no gradient; only 255 carries the
    // reference's constant log Bern(0 | clamp(0))
    const bool obs = y < 2u;
    if (MODEL <= 2) {
        // 19 VALU + 3 transcendental issues (VALU issue time adds to fp32-MFMA time on gfx950: count them)
        const float yf = (float)y;
        const float ZL = 15.942384719848633f;       // logit(1 - eps32)
        const float zc = __builtin_amdgcn_fmed3f(z, -ZL, ZL);
        const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * fabsf(zc));   // exp(-|zc|) in (1e-7, 1]
        const float t = 1.0f + e;                                                   // in (1, 2]: no denormal care
        const float r = fast_rcp(t);
        const float sp = fmaf(__builtin_amdgcn_logf(t), 0.6931471805599453f, fmaxf(zc, 0.f));   // softplus(zc)
        const float sg = (zc >= 0.f) ? r : e * r;
        // not observed: 255 (missing) carries the reference's constant, 254 (outside the problem) nothing
        const float mval = fmaf(yf, VX_LOGP_MISSING, -254.0f * VX_LOGP_MISSING);
        float lp0 = fmaf(yf, zc, -sp), d0 = yf - sg;
        asm("" : "+v"(lp0), "+v"(d0));                       // keep the chain branch-free (no exec-mask skip per cell)
        lp = obs ? lp0 : mval;
        dz = (obs && zc == z) ? d0 : 0.f;                    // zero gradient where the clamp is active
        dc = 0.f; dd = 0.f;
    } else {
        const float e = __expf(-fabsf(z));
        const float r = fast_rcp(1.0f + e);
        const float sg = (z >= 0.f) ? r : e * r;             // sigmoid(z)
        const float sn = (z >= 0.f) ? e * r : r;             // sigmoid(-z) = 1 - sg, no cancellation
        const float dmc = d - c;
        const float P = c + dmc * sg;
        const float Q = omd + dmc * sn;
        const bool inside = obs && (P >= VX_EPS32) && (Q >= VX_EPS32);
        const float Pc = fminf(fmaxf(P, VX_EPS32), 1.0f - VX_EPS32);
        const float Qc = fminf(fmaxf(Q, VX_EPS32), 1.0f - VX_EPS32);
        const float sel = (y == 1u) ? Pc : Qc;
        lp = obs ? fast_log(sel) : (y == 255u ? VX_LOGP_MISSING : 0.f);
        const float inv = fast_rcp(sel);
        const float dP = inside ? ((y == 1u) ? inv : -inv) : 0.f;
        dz = dP * dmc * sg * sn;
        dc = dP * sn * c * (1.0f - c);               // w.r.t. unconstrained c (sigmoid transform)
        dd = (MODEL == 4) ? dP * sg * d * omd : 0.f;
    }
}

// The same cell with the response handed over as a float (0, 1, 254 = outside the problem, 255 = missing): the byte comes
// out of a packed word by v_cvt_f32_ubyteN, no integer copy of it is needed.
template <int MODEL>
__device__ __forceinline__ void irt_cell_f(float z, float yf, float c, float d, float omd, float& lp, float& dz, float& dc,
                                           float& dd) {
    const bool obs = yf < 1.5f;
    if (MODEL <= 2) {
        const float ZL = 15.942384719848633f;       // logit(1 - eps32)
        const float zc = __builtin_amdgcn_fmed3f(z, -ZL, ZL);
        const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * fabsf(zc));   // exp(-|zc|) in (1e-7, 1]
        const float t = 1.0f + e;
        const float r = fast_rcp(t);
        const float sp = fmaf(__builtin_amdgcn_logf(t), 0.6931471805599453f, fmaxf(zc, 0.f));   // softplus(zc)
        const float sg = (zc >= 0.f) ? r : e * r;
        const float mval = fmaf(yf, VX_LOGP_MISSING, -254.0f * VX_LOGP_MISSING);
        float lp0 = fmaf(yf, zc, -sp), d0 = yf - sg;
        asm("" : "+v"(lp0), "+v"(d0));
        lp = obs ? lp0 : mval;
        dz = (obs && zc == z) ? d0 : 0.f;
        dc = 0.f; dd = 0.f;
    } else {
        const bool one = yf == 1.0f;
        const float e = __expf(-fabsf(z));
        const float r = fast_rcp(1.0f + e);
        const float sg = (z >= 0.f) ? r : e * r;
        const float sn = (z >= 0.f) ? e * r : r;
        const float dmc = d - c;
        const float P = c + dmc * sg;
        const float Q = omd + dmc * sn;
        const bool inside = obs && (P >= VX_EPS32) && (Q >= VX_EPS32);
        const float Pc = fminf(fmaxf(P, VX_EPS32), 1.0f - VX_EPS32);
        const float Qc = fminf(fmaxf(Q, VX_EPS32), 1.0f - VX_EPS32);
        const float sel = one ? Pc : Qc;
        lp = obs ? fast_log(sel) : (yf > 254.5f ? VX_LOGP_MISSING : 0.f);
        const float inv = fast_rcp(sel);
        const float dP = inside ? (one ? inv : -inv) : 0.f;
        dz = dP * dmc * sg * sn;
        dc = dP * sn * c * (1.0f - c);
        dd = (MODEL == 4) ? dP * sg * d * omd : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------
// Order-independent accumulation in LDS: a float is added as a 64-bit fixed-point integer (2^-32 resolution, +-2^31
// range).  Integer adds commute, so the sum does not depend on the order in which the hardware retires the atomics --
// block reductions stay bit-reproducible from run to run at the cost of a float atomic.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void fx_add(long long* slot, float v) {
    atomicAdd((unsigned long long*)slot, (unsigned long long)(long long)(v * 4294967296.0f));
}
__device__ __forceinline__ float fx_get(long long s) { return (float)((double)s * 2.3283064365386963e-10); }

// ---------------------------------------------------------------------------------------------
// wave / block reductions
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// DPP wave reductions (VALU speed; __shfl_xor lowers to ds_bpermute, an LDS-pipe round trip per step).
// After the six steps lane 63 holds the wave total; v_readlane broadcasts it through an SGPR.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov0(float v) {           // lanes not written (or reading out of range) get 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, true));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_movv(float v, float old) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                                 CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += dpp_mov0<0xB1, 0xF>(v);          // quad_perm [1,0,3,2]
    v += dpp_mov0<0x4E, 0xF>(v);          // quad_perm [2,3,0,1]
    v += dpp_mov0<0x141, 0xF>(v);         // row_half_mirror
    v += dpp_mov0<0x140, 0xF>(v);         // row_mirror          -> every lane holds its 16-lane row sum
    v += dpp_mov0<0x142, 0xA>(v);         // row_bcast:15 into rows 1 and 3
    v += dpp_mov0<0x143, 0xC>(v);         // row_bcast:31 into rows 2 and 3 -> lane 63 = wave sum
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max_dpp(float v) {
    v = fmaxf(v, dpp_movv<0xB1, 0xF>(v, v));
    v = fmaxf(v, dpp_movv<0x4E, 0xF>(v, v));
    v = fmaxf(v, dpp_movv<0x141, 0xF>(v, v));
    v = fmaxf(v, dpp_movv<0x140, 0xF>(v, v));
    v = fmaxf(v, dpp_movv<0x142, 0xA>(v, v));
    v = fmaxf(v, dpp_movv<0x143, 0xC>(v, v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// *w = max(*w, float bits of v) for a non-negative v, by integer atomicMax (order-free) -- but only when it would RAISE the
// word: atomics on one address serialise in the L2 (15 625 blocks x 12 of them: +1.5 ms on a 0.7 ms kernel, measured), and
// after the first few waves the word already holds a value no later wave exceeds.  The word is read first; a stale read can
// only cost a redundant atomic.
__device__ __forceinline__ void atomic_max_raise(uint32_t* w, float v) {
    const uint32_t b = __builtin_bit_cast(uint32_t, v);
    if (b > *(volatile const uint32_t*)w) atomicMax(w, b);
}
// The partner half's value: after `v_permlane32_swap_b32 a, b` lanes 0..31 of `b` hold what lanes 32..63 of `a` held and lanes
// 32..63 of `a` what lanes 0..31 of `b` held (gfx950, VALU: no LDS round trip).  The clang builtin mis-assigns its second
// result on this toolchain, hence the asm -- and the asm must not OWN a register: hipcc checks nothing an asm statement
// writes against MFMAs in flight (docs/HARDWARE.md rule 40).  A scratch output ("=&v") was free to land in the dead upper
// registers of an accumulator whose MFMA had just been issued (k_hodina_m reads 5 of the 16 registers of its attribute
// product); the MFMA then wrote its result over the copy several passes later and the swap returned garbage for those
// lanes -- in one schedule, not in the next.  Both operands are therefore read-write and initialised by compiler-visible
// code: the compiler pads ITS write against the MFMA (12 states behind a 32x32x16 product), after which the register is
// live and nothing in flight can still write it.  The two wait states a vector write needs in front of a permlane swap
// open the statement.
__device__ __forceinline__ void permlane32_swap(unsigned& a, unsigned& b) {
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
// v[lane & 31] + v[(lane & 31) + 32] in every lane
__device__ __forceinline__ float half_sum32(float v) {
    unsigned a = __builtin_bit_cast(unsigned, v), b = a;
    permlane32_swap(a, b);
    return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
// the partner lane's value (lane ^ 32)
__device__ __forceinline__ float half_swap32(float v) {
    unsigned a = __builtin_bit_cast(unsigned, v), b = a;
    permlane32_swap(a, b);
    return __builtin_bit_cast(float, (threadIdx.x & 32) ? a : b);
}
__device__ __forceinline__ float lane_bcast(float v, int lane_uniform) {      // lane index must be wave-uniform
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane_uniform));
}

// ---------------------------------------------------------------------------------------------
// Global -> LDS DMA (gfx950 global_load_lds_dword / dwordx4): lane i's 4 / 16 bytes land at lds + i * 4 / 16; lanes
// switched off by EXEC move nothing.  `lds` must be wave-uniform.  Issued through inline asm on purpose: the compiler
// then neither knows nor waits (vmcnt) for it at the next LDS read, so the transfer overlaps the MFMA phases; the
// kernel owns the `s_waitcnt vmcnt(0)` (vx_wait_vmem) that must precede the barrier publishing the data.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lds_addr_uniform(const void* lds) {
    return __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) const void*)lds);
}
__device__ __forceinline__ void dma16(const void* gptr, uint32_t lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(lds_byte_addr) : "memory", "m0");
}
// scalar base + 32-bit per-lane byte offset (no 64-bit VALU address arithmetic per transfer)
__device__ __forceinline__ void dma16s(const void* sbase_uniform, uint32_t voff, uint32_t lds_byte_addr) {
    const uint64_t a = (uint64_t)sbase_uniform;
    const uint64_t sa = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sa),
                 "s"(__builtin_amdgcn_readfirstlane((int)lds_byte_addr)) : "memory", "m0");
}
__device__ __forceinline__ void dma4(const void* gptr, uint32_t lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(gptr), "s"(lds_byte_addr) : "memory", "m0");
}
__device__ __forceinline__ void vx_wait_vmem() { __builtin_amdgcn_s_waitcnt(0x0F70); }   // vmcnt(0) only

#define VX_CHECK_LAUNCH()                                  \
    do {                                                   \
        hipError_t e__ = hipGetLastError();                \
        if (e__ != hipSuccess) return (int)e__;            \
    } while (0)
