// Shared pieces of the hidden_dim == 64 guide kernels (k_mvn_enc_r.hip, k_mvn_enc_bwd_fast.hip, k_mvn_packed.hip,
// k_mvn_fwd_b.hip): the branch-free row code of the reference row order and the LDS row stride of the staged responses.
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
