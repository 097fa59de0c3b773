// The per-step weight images of the f16x2 guide kernels in TWO launches instead of seven (k_pack_heads, k_clear_words,
// k_enc_scales_max, k_enc_scales, k_pack_w1_b, k_pack_heads_b and -- in the backward call -- k_pack_heads_hb): a launch costs
// a step 4-5 us however little it does (tools/minibatch_probe.py: 13 such launches were 61 of the 350 us of a B = 100 step,
// 60 of the 1 480 us of a 125 k-person shard), and the first two kernels of the old chain ran one after the other for no
// reason.  Same arithmetic, same images, bit for bit: the bodies are the ones the single kernels run.
//   stage 1   blocks [0, Rp / 4): four packed rows each (k_pack_heads);  the next FB_SC_BLOCKS blocks: the partial maxima of
//             the parameters (k_enc_scales_max), ONE set of four floats per block into sc[FB_SC_PART ..] -- no atomics,
//             nothing to clear.
//   stage 2   every block first turns the partial maxima into the scale words (a fixed-order maximum: identical in every
//             block), block 0 files them in sc[0 .. 10] for the kernels of the step and clears sc[11 .. 14], the words that
//             collect the step's largest |gx|, |gd|, |eps|, |ghpre| (k_mvn_enc_bwd_h_b*);  then by role: the fc1 k-step images
//             (k_pack_w1_b), the head tile images (k_pack_heads_b), the unit images of the hidden gradient (k_pack_heads_hb).
// (included by vx_abi.hip after k_mvn_bwd_hb.hip)
#pragma once

__device__ __forceinline__ void rows_from_ring(const int64_t* __restrict__ ring, int64_t ring_stride, int ring_slots,
                                               const uint32_t* __restrict__ step_dev, int64_t* __restrict__ rows_out, int64_t nb) {
    const int64_t* src = ring + (int64_t)(*step_dev % (uint32_t)ring_slots) * ring_stride;
    for (int64_t i = threadIdx.x; i < nb; i += blockDim.x) rows_out[i] = __builtin_nontemporal_load(src + i);
}
// (the same as a launch of its own: a forward path without the fused pack)
__global__ __launch_bounds__(256) void k_rows_from_ring(const int64_t* __restrict__ ring, int64_t ring_stride, int ring_slots,
                                                        const uint32_t* __restrict__ step_dev, int64_t* __restrict__ rows_out,
                                                        int64_t nb) {
    rows_from_ring(ring, ring_stride, ring_slots, step_dev, rows_out, nb);
}

__global__ __launch_bounds__(256) void k_pack_stage1(int D, int J, const float* __restrict__ W1, const float* __restrict__ b1,
                                                     const float* __restrict__ W21, const float* __restrict__ b21,
                                                     const float* __restrict__ W22, const float* __restrict__ b22,
                                                     float* __restrict__ Wp, float* __restrict__ bp, uint32_t* __restrict__ gtab,
                                                     float* __restrict__ WpT /*or null: no kernel of this step reads it*/,
                                                     float* __restrict__ sc, const int64_t* __restrict__ ring = nullptr,
                                                     int64_t ring_stride = 0, int ring_slots = 1,
                                                     const uint32_t* __restrict__ step_dev = nullptr,
                                                     int64_t* __restrict__ rows_out = nullptr, int64_t nb = 0,
                                                     int n_row_blocks = -1 /*(Rp + 3) / 4, or 0: stage 2 reads the parameters
                                                     themselves (direct) and nothing of this step needs Wp / bp / WpT*/) {
    if (n_row_blocks < 0) n_row_blocks = (pk_rows(D) + 3) / 4;
    const int Rp = pk_rows(D);
    const int blk = blockIdx.x, tid = threadIdx.x;
    if (blk == n_row_blocks + FB_SC_BLOCKS) {
        // one more block, with a ring: the step's row indices from the pinned host ring (vx_irt_cfg.rows_ring) into the
        // device buffer the kernels behind this launch read -- no copy and no gap in front of the replay
        rows_from_ring(ring, ring_stride, ring_slots, step_dev, rows_out, nb);
        return;
    }
    if (blk < n_row_blocks) {
        const int T = D * (D + 1) / 2;
        const int pr = 4 * blk + (tid >> 6), hh = tid & 63;
        if (pr >= Rp) return;
        int src;
        uint32_t gcode;
        pk_decode(pr, D, T, src, gcode);
        const float v = (src < 0) ? 0.f : (src < T ? W22[(int64_t)src * 64 + hh] : W21[(int64_t)(src - T) * 64 + hh]);
        Wp[(int64_t)pr * 64 + hh] = v;
        if (WpT) WpT[(int64_t)hh * Rp + pr] = v;
        if (hh == 0) {
            bp[pr] = (src < 0) ? 0.f : (src < T ? b22[src] : b21[src - T]);
            if ((pr & 7) == 0) gtab[pr >> 3] = gcode;
        }
        return;
    }
    const int b = blk - n_row_blocks;                                  // 0 .. FB_SC_BLOCKS - 1
    __shared__ float red[4][4];
    float mw, mb, m1, l1;
    enc_scales_block_max(b, D, J, W1, b1, W21, b21, W22, b22, mw, mb, m1, l1);
    if ((tid & 63) == 0) { red[tid >> 6][0] = mw; red[tid >> 6][1] = mb; red[tid >> 6][2] = m1; red[tid >> 6][3] = l1; }
    __syncthreads();
    if (tid < 4) sc[FB_SC_PART + 4 * b + tid] = fmaxf(fmaxf(red[0][tid], red[1][tid]), fmaxf(red[2][tid], red[3][tid]));
}

__global__ __launch_bounds__(256) void k_pack_stage2(int D, int J, int n_tiles, int n_off_groups, const float* __restrict__ W1,
                                                     const float* __restrict__ W21, const float* __restrict__ W22,
                                                     const float* __restrict__ Wp, const float* __restrict__ bp,
                                                     const uint32_t* __restrict__ gtab, float* __restrict__ sc,
                                                     uint8_t* __restrict__ w1img, uint8_t* __restrict__ img, uint32_t* __restrict__ gt2,
                                                     uint8_t* __restrict__ himg /*or null*/,
                                                     const float* __restrict__ b21 = nullptr /*direct: the tile images from the
                                                     parameters themselves, gtab written here (stage 1 made no packed copy)*/,
                                                     const float* __restrict__ b22 = nullptr, uint32_t* __restrict__ gtab_out = nullptr) {
    __shared__ float scl[16];
    const int tid = threadIdx.x, blk = blockIdx.x;
    if (tid < 64) {
        const f32x4 v = *(const f32x4*)(sc + FB_SC_PART + 4 * tid);    // FB_SC_BLOCKS == 64: one block's four maxima per lane
        const float mw = wave_max_dpp(v[0]), mb = wave_max_dpp(v[1]), m1 = wave_max_dpp(v[2]), l1 = wave_max_dpp(v[3]);
        if (tid == 0) enc_scales_from_max(mw, mb, m1, l1, scl);
    }
    __syncthreads();
    if (blk == 0) {
        if (tid < 11) sc[tid] = scl[tid];
        else if (tid < 15) ((uint32_t*)sc)[tid] = 0u;                  // the step's operand maxima start from zero
    }
    const int n_w1 = (J + 15) / 16;
    if (blk < n_w1) { pack_w1_b_kstep(blk, J, W1, scl[0], w1img); return; }
    if (blk < n_w1 + n_tiles) {
        if (gtab_out) pack_heads_b_tile<true>(blk - n_w1, n_off_groups, Wp, bp, gtab, scl[2], scl[5], img, gt2, D, W21, b21, W22, b22, gtab_out);
        else pack_heads_b_tile<false>(blk - n_w1, n_off_groups, Wp, bp, gtab, scl[2], scl[5], img, gt2);
        return;
    }
    if (himg) pack_heads_hb_unit(blk - n_w1 - n_tiles, D, W21, W22, scl[2], himg);
}

// The last launch of vx_mvn_enc_backward on the packed row space, three kernels of ~5 us in one:
//   blocks [0, n_unpack)            the head gradients summed over their slabs, back in the reference layout (k_unpack_head_grads)
//   blocks [n_unpack, + n_f1)       the fc1 slabs summed (k_reduce_slabs) -- none when the side stream's kernel did it
//   one more block, with ll         the loss of a small batch, out[0] = alpha (sum ll + sum ent) in k_sum_stage1's one-block
//                                   order; the device step counter of a captured step advances here (no block of this launch
//                                   reads it)
// Same bodies, same sums, bit for bit.
__global__ __launch_bounds__(256) void k_enc_bwd_tail(int D, int H, const float* __restrict__ slabs_w, int n_prw, int64_t lenw,
                                                      float alpha, float* __restrict__ out_w, int n_unpack,
                                                      const float* __restrict__ slabs_f, int64_t n_prf, int64_t lenf,
                                                      float* __restrict__ out_f, int n_f1, const float* __restrict__ ll,
                                                      const float* __restrict__ ent, int64_t nb, float loss_alpha,
                                                      float* __restrict__ loss, float* __restrict__ sum_ws,
                                                      uint32_t* __restrict__ tick) {
    __shared__ float part[4][64];
    const int blk = blockIdx.x;
    if (blk < n_unpack) { unpack_head_rows(blk, D, H, slabs_w, n_prw, lenw, alpha, out_w); return; }
    if (blk < n_unpack + n_f1) { reduce_slabs_cols(blk - n_unpack, n_f1, part, slabs_f, n_prf, lenf, lenf, alpha, out_f); return; }
    if (ll) sum_block(0, 1, &part[0][0], ll, nb, sum_ws, ent, loss_alpha, loss, tick);
}

// A small batch's head weight gradient (k_mvn_enc_bwd_w_t: 44 workgroups at B = 100) and its fc1 weight gradient
// (k_fc1_bwd<2, 1>: 16) need nothing from each other and both run behind the hidden gradient: ONE launch, the first n_wt
// workgroups the one, the rest the other (15 + 9 us one after the other on a chip they fill to a sixth).  Same bodies, same
// slabs.  (Registers and LDS are the larger kernel's: one wave a SIMD, as k_mvn_enc_bwd_w_t alone.)
__global__ __launch_bounds__(BT_THREADS, 1) void k_bwd_wt_fc1(
    EncDims dm, const float* __restrict__ hT, const float* __restrict__ epsT, const float* __restrict__ gdT,
    const float* __restrict__ gxT, const uint32_t* __restrict__ gtab, float* __restrict__ slabs_w, int64_t lenw, int wt_gx, int wt_gy,
    const uint8_t* __restrict__ y, const int64_t* __restrict__ rows, const float* __restrict__ ghpre, float* __restrict__ slabs_f,
    int64_t lenf, int fast, int f_gx, int f_gy) {
    extern __shared__ __attribute__((aligned(16))) char smem_wf[];
    const int L = blockIdx.x, n_wt = wt_gx * wt_gy;
    if (L < n_wt) {
        bwd_w_t_body(dm, hT, epsT, gdT, gxT, gtab, slabs_w, lenw, smem_wf, VGrid{L, wt_gx, wt_gy});
        return;
    }
    const int Lf = L - n_wt;
    fc1_bwd_body<2, 1>(dm, y, rows, ghpre, slabs_f, lenf, fast, (float*)smem_wf, Lf % f_gx, Lf / f_gx, f_gy);
}
