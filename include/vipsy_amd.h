/*
 * vipsy_amd C ABI -- the drop-in boundary for the ELBO-gradient hot path on MI355X (gfx950).
 *
 * The reference (inuyasha2012/vipsy) has no native boundary: the op being replaced is
 *     loss = svi.step(data)                                   vi.py:503-516, called from
 *     BaseIRT.fit vi.py:634-638 / VCHoDina.fit vi.py:937-942
 * whose arithmetic lives in pyro-ppl 1.4.0 (Trace_ELBO / TraceEnum_ELBO + pyro.optim.Adam).
 * This header is what a binding for that op links against: plain C, device pointers and sizes,
 * no torch types.  Every entry point
 *   - is asynchronous on the hipStream_t passed as `void* stream`,
 *   - allocates nothing and never synchronises (graph-capturable),
 *   - borrows device memory owned by the caller (PyTorch-ROCm in our host code),
 *   - returns 0 on success, a hipError_t (>0) from the launch, or a VX_E* code (<0) for bad
 *     arguments.
 *
 * One step of the reference loop maps to, in order (vipsy_amd/engine.py drives exactly this):
 *   guide forward  (encoder / per-person variational rows -> x, entropy)   vi.py:673-723
 *   model likelihood + gradients                                             vi.py:574-625
 *   guide backward (encoder weight grads / per-person grads)
 *   [multi-GPU: one RCCL all-reduce of the flat gradient buffer -- done by the host]
 *   optimiser on the unconstrained leaves, `free` mask applied               vi.py:508-514
 *
 * Data contract: responses are uint8, 1 byte per cell: 0, 1, or 255 = missing (the reference
 * holds float32 with NaN, vi.py:621).  All parameters / state are float32, unconstrained
 * (vi.py:510).  Persons are rows; a rank owns the contiguous shard [gid0, gid0 + n_local).
 */
#ifndef VIPSY_AMD_H
#define VIPSY_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VX_OK 0
#define VX_EINVAL (-1)       /* bad argument / unsupported shape */
#define VX_EUNIMPL (-2)
#define VX_ABI_VERSION 5

enum vx_model { VX_IRT_1PL = 1, VX_IRT_2PL = 2, VX_IRT_3PL = 3, VX_IRT_4PL = 4 }; /* vi.py:538-543 */

/* Problem description shared by the IRT entry points (BaseIRT.__init__, vi.py:545-572). */
typedef struct vx_irt_cfg {
    int32_t model;       /* enum vx_model */
    int32_t D;           /* x_feature: latent dimensions */
    int32_t J;           /* item_size */
    int32_t H;           /* encoder hidden_dim (amortized guides; <= 128), else 0 */
    float   Dc;          /* the scalar `D` of vi.py:549 (1 or 1.702) */
    float   scale;       /* plate scale N_global / B_global */
    uint64_t seed;       /* Philox key */
    uint32_t step;       /* Philox counter word 2: optimisation step */
    uint32_t stream;     /* Philox counter word 3 high half: particle index */
    const uint32_t* step_dev; /* or NULL: the step counter in DEVICE memory -- vx_mvn_enc_forward / vx_mvn_bbvi_forward read the Philox step from it
                            instead of `step`, so that a whole step can be captured once in a HIP graph and replayed
                            (vx_sum2 advances it, vx_adam_step reads it; the D = 1 entry points take it as an argument) */
    const int64_t* rows_ring; /* or NULL.  A captured subsampled step (test.py:338-343: a new draw of B rows every step) needs its
                            row indices on the device without a copy of its own in front of every replay: the host writes
                            the draw of step t into slot t % rows_ring_slots of this PINNED HOST buffer
                            ([rows_ring_slots][rows_ring_stride] int64), and vx_mvn_enc_forward -- with step_dev, and `rows`
                            then being the DEVICE buffer the kernels of the step read -- first copies slot
                            *step_dev % rows_ring_slots into `rows` (nb indices), inside its first launch where it can.  The
                            host must not rewrite a slot before the step that read it has finished. */
    int64_t rows_ring_stride;
    int32_t rows_ring_slots;
    int32_t _pad;
} vx_irt_cfg;

int vx_abi_version(void);
const char* vx_build_info(void);

/* ---- Measurement aid (no reference counterpart; used by bench.py only).  While enabled, the entry points record a
 * pair of HIP events on their launch stream around each of the large kernels; vx_prof_read synchronises on them and
 * returns the kernel's name, its mean duration and the number of launches seen since vx_prof_enable(1).  Disabled
 * (the default) nothing is recorded and no entry point synchronises.
 *   on = 0: off, what was recorded is dropped;  1: on, what was recorded is dropped;
 *   on = 3: PAUSE (off, the records stay);  2: RESUME (on, the records stay) -- bench.py brackets a SAMPLE of the timed steps:
 *   an event pair around a kernel costs the stream a few microseconds (25 pairs a step: 0.11 ms of the 9.0 ms headline step). */
int vx_prof_enable(int on);
int vx_prof_count(void);
int vx_prof_read(int slot, char* name, int name_cap, float* mean_ms, int* launches);
/* persons the last launch of that kernel took when the batch is divided between two kernels (the amortized forward gives
 * whole chip rounds to k_mvn_enc_fwd_b2 and a short last round to k_mvn_enc_fwd_b); 0 = the whole batch */
int vx_prof_units(int slot, int64_t* units);

/* ---- RNG: eps[i, d] = N(0,1) keyed by the GLOBAL person id (Philox4x32-10 + Box-Muller).
 * gids == NULL means gid = gid0 + i.  Also used by tests to hand the oracle identical eps. */
int vx_philox_normals(float* eps /*[n][D]*/, const int64_t* gids, int64_t gid0, int64_t n, int32_t D,
                      uint64_t seed, uint32_t step, uint32_t stream, void* hip_stream);
/* raw Philox words for bit-exact checks against the oracle: out[i][4] for counter (i, 0, step, stream<<16) */
int vx_philox_raw(uint32_t* out /*[n][4]*/, int64_t gid0, int64_t n, uint64_t seed, uint32_t step,
                  uint32_t stream, void* hip_stream);

/* ---- amortized multivariate-normal guide, forward (MvnEncoder.forward + rsample of
 * MultivariateNormal(loc, scale_tril=LowerCholeskyTransform(M)); vi.py:438-455, 685-693).
 *   rows   : local row index of each batch member (NULL = 0..nb-1), gids for the RNG = gid0+row
 *   eps_in : if non-NULL use these draws instead of generating them (parity tests)
 * Outputs (all [nb][...], float32): h = softplus(fc1), x = loc + L eps, eps, ldT[D][nb] =
 * exp(diag M) stored dimension-major, ent[i] = 0.5|eps_i|^2 + sum_k M_kk (= log p/q constants
 * cancelled). */
int vx_mvn_enc_forward(const vx_irt_cfg* cfg, const uint8_t* y /*[n_local][J]*/, const int64_t* rows,
                       int64_t nb, int64_t gid0,
                       const float* W1 /*[H][J]*/, const float* b1, const float* W21 /*[D][H]*/,
                       const float* b21, const float* W22 /*[T][H]*/, const float* b22,
                       const float* eps_in, float* h /*[nb][H]*/, float* x /*[nb][D]*/,
                       float* eps /*[nb][D]*/, float* ldT /*[D][nb]*/, float* ent /*[nb]*/,
                       float* hT /*[H][nb] or NULL*/, float* epsT /*[D][nb] or NULL*/,
                       float* packws /*vx_mvn_pack_floats(cfg) floats or NULL*/,
                       uint8_t* ximg /*vx_irt_lik_ximg_bytes(cfg, nb) bytes or NULL*/,
                       uint16_t* hs /*[2][64][nb] fp16 or NULL*/, void* hip_stream);
/* hs (optional, with hT, hidden_dim 64): the two fp16 terms of hT 2^sh (the power of two that `packws` carries for h), the
 * operand of the head weight-gradient kernel -- point it at workspace + vx_mvn_enc_bwd_hs_offset(cfg, nb) of the backward
 * call and set bit 1 of its gd_ready. */
/* ximg (optional, vx_irt_lik_ximg_bytes(cfg, nb) > 0 only: 96 <= D <= 111, 1PL / 2PL link): x once more, as the pre-split
 * operand image of the fp16-MFMA likelihood kernel (two fp16 terms of x 2^7 per value, in that kernel's LDS tile order,
 * followed by one overflow word that the writers raise for a latent of |x| >= 511.75); hand the same buffer to
 * vx_irt_lik_grad, which otherwise makes it itself. */
/* hT / epsT: optional dimension-major copies of h and eps (person-contiguous rows) for the weight-gradient kernel
 * of vx_mvn_enc_backward; written only by the packed fast path (H == 64, D % 4 == 0, J % 4 == 0). */
/* `packws` holds this step's packed copy of the head weights (a re-ordering of fc22 | fc21 rows that the
 * fast kernels use, see vipsy_amd/csrc/k_pack.hip), their fp16-pair operand images and the powers of two of the
 * f16x2 operands; forward fills it, the matching backward call of the SAME step reads it. */
int64_t vx_mvn_pack_floats(const vx_irt_cfg* cfg);
/* float offset inside packws of the three words that collect the step's largest |gx|, |gd|, |eps| (vx_irt_lik_grad's opmax;
 * cleared by vx_mvn_enc_forward), or -1 when this (cfg, nb) does not run the f16x2 kernels that use them */
int64_t vx_mvn_pack_opmax_offset(const vx_irt_cfg* cfg, int64_t nb);

/* ---- model likelihood + gradients for D >= 2 (irt_2pl..4pl + _get_p_data mask + Bernoulli
 * log-lik; vi.py:32-66, 596-625).  Consumes x, produces
 *   gx[nb][D]      d ELBO / d x  = scale * (R A^T - x)          (likelihood + N(0,I) prior)
 *   ll[nb]         per-person  log p(y_i | x_i) - 0.5 |x_i|^2
 *   gitem          gradient of the LOSS (= -ELBO), float [D*J + 3*J] laid out
 *                  [a: D*J | b: J | c_un: J | d_un: J]  (c/d w.r.t. the unconstrained leaves,
 *                  zero for models without them); summed over this rank's batch only.
 * `workspace`: vx_irt_lik_workspace_floats(cfg, nb) floats of scratch (partial slabs). */
int64_t vx_irt_lik_workspace_floats(const vx_irt_cfg* cfg, int64_t nb);
/* size of the optional x operand image (0: this shape does not use one) */
int64_t vx_irt_lik_ximg_bytes(const vx_irt_cfg* cfg, int64_t nb);
int vx_irt_lik_grad(const vx_irt_cfg* cfg, const uint8_t* y, const int64_t* rows, int64_t nb,
                    const float* x /*[nb][D]*/, const float* a /*[D][J]*/, const float* b /*[J]*/,
                    const float* c_un /*[J] or NULL*/, const float* d_un /*[J] or NULL*/,
                    float* gx /*[nb][D] or NULL*/, float* gxT /*[D][nb] or NULL*/, float* ll /*[nb]*/,
                    float* gitem /*[D*J + 3*J]*/, float* workspace,
                    const uint8_t* yT /*[J + 1][yT_stride] or NULL*/, int64_t yT_stride,
                    const uint8_t* ximg /*as written by vx_mvn_enc_forward, or NULL*/,
                    const float* epsT /*[D][nb] or NULL*/, const float* ldT /*[D][nb] or NULL*/,
                    float* gdT /*[D][nb] or NULL: fused output gxT * epsT * ldT + scale*/,
                    uint32_t* opmax /*[3] or NULL, with gdT*/, void* hip_stream);
/* gx and gxT are the same gradient in person-major / dimension-major order; at least one must be given.
 * gdT (optional, with gxT, epsT, ldT; nb * D % 4 == 0): the DIAG-row operand of the dimension-major guide-backward kernels,
 * written in the same pass over gxT -- point it at workspace + vx_mvn_enc_bwd_gd_offset(cfg, nb) of the backward call and
 * hand that call gd_ready = 1.
 * opmax (optional, with gdT): three words that receive, by integer atomicMax on the float bits, the largest |gx|, |gd| and
 * |eps| of the batch -- what the f16x2 head weight-gradient kernel takes its power of two from.  Point it at packws +
 * vx_mvn_pack_opmax_offset(cfg) (cleared by the forward call of the step) and set bit 2 of the backward call's gd_ready: the
 * head weight gradient then starts beside the hidden gradient instead of behind it.
 * yT (optional, full batches only: rows == NULL): the responses item-major -- row j = item j over the batch rows, row J
 * and every column past nb filled with 254 ("outside the problem"), yT_stride % 64 == 0 and >= nb rounded up to 64.
 * With it, 96 <= D <= 111 runs on the MFMA kernels -- 1PL / 2PL: k_irt_lik_h.hip (fp32 results from two fp16 terms per
 * operand, three products; a batch with a latent of |x| >= 511.75 is taken by k_irt_lik_b.hip instead, decided on the
 * device), 3PL / 4PL: k_irt_lik_b.hip (three bf16 terms, six products); the same buffer serves vx_mvn_enc_backward / vx_norm_enc_backward (which read rows 0..J-1). */

/* ---- amortized MVN guide, backward: encoder weight gradients of the LOSS from gx.
 * genc = d LOSS / d encoder parameters, flat in the nn.Linear order of vi.py:442-444:
 *   [W1: H*J | b1: H | W21: D*H | b21: D | W22: T*H | b22: T],  T = D(D+1)/2. */
int64_t vx_mvn_enc_param_floats(const vx_irt_cfg* cfg);          /* length of genc */
int64_t vx_mvn_enc_bwd_workspace_floats(const vx_irt_cfg* cfg, int64_t nb);
/* 1: this (cfg, nb) runs on the dimension-major kernels when hT, epsT and gxT are supplied (gx may then be NULL and
 * the likelihood call needs to produce gxT only); 0: the person-major kernels, which need gx. */
int vx_mvn_enc_bwd_layout(const vx_irt_cfg* cfg, int64_t nb);
int vx_mvn_enc_backward(const vx_irt_cfg* cfg, const uint8_t* y, const int64_t* rows, int64_t nb,
                        const float* W21, const float* W22,
                        const float* h, const float* eps, const float* ldT, const float* gx,
                        const float* hT /*or NULL*/, const float* epsT /*or NULL*/, const float* gxT /*or NULL*/,
                        const uint8_t* yT /*[>= J][yT_stride] item-major copy of y (pad bytes 0 or 254), or NULL*/, int64_t yT_stride,
                        float* genc, float* workspace, const float* packws,
                        int32_t gd_ready /*bit 0: vx_irt_lik_grad already wrote gdT into the workspace; bit 1:
                                           vx_mvn_enc_forward already wrote hs there; bit 2: the operand maxima are in packws
                                           (vx_irt_lik_grad's opmax)*/, void* hip_stream);
/* The same call, and the loss of the step with it: loss[0] = loss_alpha * (sum ll + sum ent) over the nb persons of the
 * batch -- what vx_sum2 computes, bit for bit (vi.py:516, the value svi.step returns) -- from the call's LAST launch, and the
 * device step counter of a captured step (cfg->step_dev) advanced there as vx_sum2 would.  A B = 100 step (test.py:338) is
 * a dozen launches of which this saves one; batches above 4 096 persons run vx_sum2's two launches behind the gradients.
 * sum_workspace: vx_sum_workspace_floats() floats. */
int vx_mvn_enc_backward_loss(const vx_irt_cfg* cfg, const uint8_t* y, const int64_t* rows, int64_t nb,
                             const float* W21, const float* W22,
                             const float* h, const float* eps, const float* ldT, const float* gx,
                             const float* hT /*or NULL*/, const float* epsT /*or NULL*/, const float* gxT /*or NULL*/,
                             const uint8_t* yT /*or NULL*/, int64_t yT_stride, float* genc, float* workspace,
                             const float* packws, int32_t gd_ready, const float* ll /*[nb]*/, const float* ent /*[nb]*/,
                             float loss_alpha, float* loss /*[1]*/, float* sum_workspace, void* hip_stream);
/* float offset of gdT[D][nb] inside the backward workspace, or -1 when this (cfg, nb) has no such operand */
int64_t vx_mvn_enc_bwd_gd_offset(const vx_irt_cfg* cfg, int64_t nb);
/* float offset of hs[2][64][nb] (fp16) inside the backward workspace, or -1 when this (cfg, nb) does not use it */
int64_t vx_mvn_enc_bwd_hs_offset(const vx_irt_cfg* cfg, int64_t nb);
/* With hT, epsT and gxT (the dimension-major copies made by the forward / likelihood calls) the head weight
 * gradients run on the DMA-staged kernel of k_mvn_bwd_t.hip; without them on the person-major one.
 * yT (full batch only, rows == NULL, yT_stride % 16 == 0): the responses item-major, which lets the fc1 weight
 * gradient read 16 persons of an item with one 16-byte LDS load; the responses never change, so the host
 * transposes them once. */

/* ---- D = 1 models (irt_1pl..4pl, Normal guide; vi.py:588-595, 677-684, 701-705), fused:
 * x = loc + exp(raw) eps, likelihood, prior, entropy, gradients w.r.t. loc/raw and the items.
 *   loc/raw: [nb] (already gathered for BBVI minibatches, or encoder outputs)
 *   outputs: gloc[nb], graw[nb] = d LOSS / d loc, d raw;  elbo[nb] = per-person
 *            log p(y|x) + log p(x) - log q(x) (unscaled);
 *   gitem:   d LOSS / d [a: J | b: J | c_un: J | d_un: J] for this rank's batch (a = 0 for 1PL).
 * J <= 1024.  workspace: vx_irt1d_workspace_floats(cfg, nb) floats -- size it with THAT call, not by hand: behind the
 *   n_slabs x (4 J + 1) slab words the step kernels (vx_irt1d_grad and vx_irt1d_sparse_grad alike, with or without the
 *   *_adam tail) leave Adam's count in one trailing word, slabs[n_slabs * (4 J + 1)], for k_reduce_adam; the size query
 *   includes it (+ 4 floats).  A buffer of n_slabs * (4 J + 1) floats is 4 bytes short.
 * loss (or NULL): receives d LOSS itself, -scale * sum(elbo), summed in a fixed order.
 * step_dev (or NULL): the step counter in DEVICE memory -- the kernel reads the Philox step from it instead of cfg->step,
 * and the call ADVANCES it by one when the gradients are done (so vx_adam_step's t_dev may point at the same word: Adam's
 * count is the step + 1).  Lets a whole step be captured once in a HIP graph and replayed. */
int64_t vx_irt1d_workspace_floats(const vx_irt_cfg* cfg, int64_t nb);
int vx_irt1d_grad(const vx_irt_cfg* cfg, const uint8_t* y, const int64_t* rows, int64_t nb, int64_t gid0,
                  const float* loc, const float* raw, const float* eps_in,
                  const float* a, const float* b, const float* c_un, const float* d_un,
                  float* gloc, float* graw, float* elbo, float* gitem, float* loss, uint32_t* step_dev,
                  float* workspace, void* hip_stream);

/* Score-function (REINFORCE) estimator with a control-variate baseline for the Normal guide of the D = 1 models
 * (BASELINE.json north_star).  The reference takes pathwise gradients for these guides (vi.py:684,705:
 * Normal has rsample), so this mode has no reference output to match: an opt-in of this build (IrtEngine(estimator=
 * "score")), checked against oracle/vi_oracle.py::irt_particle(estimator="score").  Called AFTER vx_irt1d_grad of the same
 * batch (whose item gradients and loss stand): overwrites gloc / graw [nb] with
 *     d LOSS / d loc_i = -(log_r_i - baseline_i) eps_i exp(-raw_i),   d LOSS / d raw_i = -(log_r_i - baseline_i) (eps_i^2 - 1),
 *     log_r_i = scale * elbo[i].
 * eps[nb]: the draws the step used (vx_philox_normals with the step's seed / step / stream, or the caller's own);
 * baseline: NULL, an explicit control variate in batch order (base_beta < 0), or a decaying average updated after use
 * (base_beta >= 0: baseline <- beta baseline + (1 - beta) log_r), indexed by rows[i] when base_by_row != 0 and rows != NULL;
 * log_r: optional output [nb]. */
int vx_irt1d_score_grad(int64_t nb, float scale, const float* elbo, const float* eps, const float* raw, const int64_t* rows,
                        float* baseline, float base_beta, int32_t base_by_row, float* log_r, float* gloc, float* graw,
                        void* hip_stream);

/* Score-function estimator for the multivariate Normal guides, x_feature > 1 (same standing as vx_irt1d_score_grad: an opt-in
 * of this build, IrtEngine(estimator="score"); oracle/vi_oracle.py::irt_particle).  With u_i = L_i^-T eps_i the score of the
 * guide has the pathwise gradient's shape -- d log q / d loc = u, / d L_kc = u_k eps_c, / d M_kk = u_k eps_k L_kk - 1 -- so
 * this call, made AFTER the forward and the likelihood of the same batch (whose item gradients and loss stand), only writes
 * the operands the guide-backward kernels take in place of d ELBO / d x:
 *     w[i] = log_r_i - baseline_i,  log_r_i = cfg->scale (ll[i] + ent[i])        (baseline arguments as vx_irt1d_score_grad)
 *     gx[nb][D], gxT[D][nb] = w_i u_i;   gdT[D][nb] = w_i (u_ik eps_ik L_kk - 1)   (the DIAG-row operand; any may be NULL)
 * kind 0: L from the encoder heads (h [nb][H], W22 [T][H], b22 [T]); hand gxT / gdT to vx_mvn_enc_backward (gd_ready bit 0),
 * which must be on the dimension-major kernels (vx_mvn_enc_bwd_layout == 1).  kind 1 / 2: L from M [n_local][D][D] (rows
 * index it) / the shared M [D][D]; run vx_mvn_bbvi_backward on gx with cfg->scale = 0, then vx_mvn_score_diag adds the
 * diagonal term (+ w_i on the person's own M_kk, + sum_i w_i on the shared one). */
int vx_mvn_score_operands(const vx_irt_cfg* cfg, int64_t nb, const int64_t* rows, int32_t kind, const float* h,
                          const float* W22, const float* b22, const float* M, const float* eps, const float* ll,
                          const float* ent, float* baseline, float base_beta, int32_t base_by_row, float* log_r, float* w,
                          float* gx, float* gxT, float* gdT, void* hip_stream);
int vx_mvn_score_diag(const vx_irt_cfg* cfg, int64_t nb, const int64_t* rows, const float* w, int32_t shared, float* gM,
                      void* hip_stream);
/* The kind-0 operands (L from the encoder heads) on the fp16 MFMA, for the shapes the f16x2 guide kernels take (hidden_dim 64,
 * x_feature % 4 == 0, 8 <= x_feature <= 124): u = L^-T eps as a back-substitution over COLUMN tiles of the head GEMM the
 * forward runs, 32 persons a wave (k_mvn_score_b.hip).  Same outputs as vx_mvn_score_operands(kind 0) with gx / w = NULL, to
 * the f16x2 products' 2^-22.  Made AFTER vx_mvn_enc_forward of the same batch and parameters:
 *   packws: the forward's pack workspace (the powers of two of its operands are read from its scale block);
 *   h[nb][64], eps[nb][D]: the forward's outputs (L_kk = exp(M_kk) is recomputed with the rows below it);
 *   workspace: vx_mvn_score_heads_workspace_floats(cfg) floats (the column-ordered weight image, rebuilt every call).
 * VX_EINVAL for any other shape (the caller then uses vx_mvn_score_operands). */
int64_t vx_mvn_score_heads_workspace_floats(const vx_irt_cfg* cfg);
int vx_mvn_score_heads(const vx_irt_cfg* cfg, int64_t nb, const int64_t* rows, const float* h, const float* W22, const float* b22,
                       const float* packws, const float* eps, const float* ll, const float* ent,
                       float* baseline, float base_beta, int32_t base_by_row, float* log_r, float* gxT, float* gdT,
                       float* workspace, void* hip_stream);

/* ---- black-box MVN guide with per-person or shared Cholesky rows (VIRT.guide, x_feature > 1, vi.py:706-723).
 *   loc: [n_local][D];  M: [n_local][D][D] unconstrained (shared == 0) or [D][D] (shared == 1, share_cov=True)
 *   forward : x[nb][D] = loc[row] + L eps, eps[nb][D], ent[nb] = 0.5|eps|^2 + sum_k M_kk
 *   backward: gloc[n_local][D], gM (same shape as M) = d LOSS / d loc, M for the batch rows; the caller zeroes
 *             both buffers first (dense per-person gradients, zero off the batch; shared M accumulates). */
int vx_mvn_bbvi_forward(const vx_irt_cfg* cfg, int64_t nb, const int64_t* rows, int64_t gid0, const float* loc,
                        const float* M, int32_t shared, const float* eps_in, float* x, float* eps, float* ent,
                        void* hip_stream);
int64_t vx_mvn_bbvi_bwd_workspace_floats(const vx_irt_cfg* cfg, int64_t nb, int32_t shared);
int vx_mvn_bbvi_backward(const vx_irt_cfg* cfg, int64_t nb, const int64_t* rows, const float* M, int32_t shared,
                         const float* gx, const float* eps, float* gloc, float* gM,
                         float* workspace /*shared == 1: per-block slabs, summed in fixed order into gM (overwritten)*/,
                         void* hip_stream);

/* ---- amortized Normal guide for ONE latent dimension (NormEncoder, vi.py:417-435; VaeIRT with
 * x_feature == 1, vi.py:677-684, and VaeCHoDina, vi.py:968-981).  cfg->J, cfg->H are used.
 *   forward : h[nb][H] = softplus(fc1 yin), loc[nb] = fc21 h, raw[nb] = fc22 h  (scale = exp(raw))
 *   backward: genc = d LOSS / d [W1: H*J | b1: H | W21: H | b21: 1 | W22: H | b22: 1] from
 *             gloc / graw = d LOSS / d loc, raw (as produced by vx_irt1d_grad / vx_hodina_grad). */
int64_t vx_norm_enc_param_floats(const vx_irt_cfg* cfg);
/* packws (optional): vx_norm_enc_pack_floats(cfg) floats of scratch, 16-byte aligned.  With it a large batch runs fc1 from
 * fp16-pair images of W1 made once per call (two small launches) instead of re-splitting W1 in every workgroup: 0.37 ->
 * 0.1x ms at 1M x 500.  NULL, or a batch below 4 096 persons: the kernels that need no scratch. */
int64_t vx_norm_enc_pack_floats(const vx_irt_cfg* cfg);           /* 0: this shape has no such kernel */
int vx_norm_enc_forward(const vx_irt_cfg* cfg, const uint8_t* y, const int64_t* rows, int64_t nb,
                        const float* W1, const float* b1, const float* W21, const float* b21,
                        const float* W22, const float* b22, float* h, float* loc, float* raw,
                        float* packws /*or NULL*/, void* hip_stream);
int64_t vx_norm_enc_bwd_workspace_floats(const vx_irt_cfg* cfg, int64_t nb);
int vx_norm_enc_backward(const vx_irt_cfg* cfg, const uint8_t* y, const int64_t* rows, int64_t nb,
                         const float* W21, const float* W22, const float* h, const float* gloc,
                         const float* graw, const uint8_t* yT /*[J][yT_stride] or NULL*/, int64_t yT_stride,
                         float* genc, float* workspace, void* hip_stream);

/* ---- D = 1 models on OBSERVED cells only (full batch; for heavily masked data, BASELINE config 4).  The responses
 * never change, so the host compacts them once (vipsy_amd/engine.py::_sparse_lists) into groups of 64 person SLOTS:
 *   pidx [n_groups][64] int32: the person (row of loc / raw / gloc / graw / elbo) in the slot, -1 = empty.  Any order
 *        works; lists of similar length in one group waste no lanes (the host sorts windows of persons by length);
 *   pent [n_groups][Lq][64][4] uint16: the slot's list in quads, entry = item | y << 15, 0xFFFF past its end
 *        (an empty slot's list is all 0xFFFF);
 *   glen [n_groups] int32: quads in the longest list of the group (<= Lq).
 * Same outputs as vx_irt1d_grad, one pass; item gradients are summed with integer atomics (bit-reproducible).
 * workspace: vx_irt1d_sparse_workspace_floats(cfg, n_groups) floats.  J <= 1024. */
int64_t vx_irt1d_sparse_workspace_floats(const vx_irt_cfg* cfg, int64_t n_groups);
int vx_irt1d_sparse_grad(const vx_irt_cfg* cfg, const uint16_t* pent, const int32_t* glen, int32_t Lq,
                         const int32_t* pidx, int64_t n_groups, int64_t gid0, const float* loc, const float* raw,
                         const float* eps_in, const float* a, const float* b, const float* c_un, const float* d_un,
                         float* gloc, float* graw, float* elbo, float* gitem, float* loss /*or NULL*/,
                         uint32_t* step_dev /*or NULL*/, float* workspace, void* hip_stream);

/* ---- HO-DINA with exact enumeration of the 2^K attribute patterns (VCHoDina / VaeCHoDina model,
 * vi.py:897-923, under TraceEnum_ELBO; guide theta ~ Normal(loc, exp(raw)), vi.py:925-934 / 968-981).
 *   q: [K][J] float 0/1 Q-matrix (vi.py:741);  lam0 [K], lam1_un [K] (positive -> exp), g_un / s_un [J]
 *   (interval(0,1) -> sigmoid).  K <= 10, J <= 1024.
 * Outputs: gloc / graw [nb] = d LOSS / d loc, raw;  elbo[nb] per-person ELBO terms (unscaled);
 *   gitem = d LOSS / d [g_un: J | s_un: J | lam0: K | lam1_un: K] for this rank's batch. */
typedef struct vx_hodina_cfg {
    int32_t K, J, H, _pad;
    float scale, _pad2;
    uint64_t seed;
    uint32_t step, stream;
    const uint32_t* step_dev; /* or NULL: the step counter in DEVICE memory (a captured step), read instead of `step`; vx_sum
                                 advances it behind the gradients, vx_adam_step reads it as t_dev */
} vx_hodina_cfg;
int64_t vx_hodina_workspace_floats(const vx_hodina_cfg* cfg, int64_t nb);
int vx_hodina_grad(const vx_hodina_cfg* cfg, const uint8_t* y, const int64_t* rows, int64_t nb, int64_t gid0,
                   const float* loc, const float* raw, const float* eps_in, const float* q,
                   const float* lam0, const float* lam1_un, const float* g_un, const float* s_un,
                   float* gloc, float* graw, float* elbo, float* gitem, float* workspace, void* hip_stream);

/* ---- VCCDM: pattern-enumerated DINA / DINO with the uniform pattern prior and an empty guide (vi.py:819-865;
 * dina vi.py:69-83, dino vi.py:86-101 -- the latter reproduced with its in-place sequencing: items that need a
 * single attribute always get eta = 0).  cfg: K, J, scale (seed / step / stream unused).
 *   elbo[nb] = log sum_c (1/C) prod_j Bern(y_ij | p_cj);  gitem = d LOSS / d [g_un: J | s_un: J]. */
int64_t vx_ccdm_workspace_floats(const vx_hodina_cfg* cfg, int64_t nb);
int vx_ccdm_grad(const vx_hodina_cfg* cfg, int32_t dino, const uint8_t* y, const int64_t* rows, int64_t nb,
                 const float* q /*[K][J]*/, const float* g_un, const float* s_un, float* elbo, float* gitem,
                 float* workspace, void* hip_stream);

/* ---- VCDM / VaeCDM: Bernoulli-guide DINA / DINO with the score-function (REINFORCE) estimator (vi.py:726-816),
 * as pyro's Trace_ELBO treats a non-reparameterised guide site; cfg: K, J, scale, seed, step, stream.
 *   u[nb][K]        guide logits in batch order (rows of the `attr_p` leaf gathered by the host, or the encoder output);
 *                   clamp_t = 1 when they come from a unit_interval leaf (SigmoidTransform clamps the probability)
 *   attr_in[nb][K]  the 0/1 draws to replay, or NULL: drawn in the kernel (Philox block k >> 2, word k & 3 of the person)
 *   prior_p         probability of attr = 1 under the model prior BEFORE clamping -- 1.5 reproduces vi.py:753
 *   baseline        control variate subtracted from log_r in the score term, or NULL (pyro's Trace_ELBO: none); indexed
 *                   by the local person row (base_by_row = 1) or the batch position; base_beta >= 0 turns it into a per-person
 *                   decaying average: baseline <- beta baseline + (1 - beta) log_r after use (any baseline that does not
 *                   depend on the person's own draw keeps the estimator unbiased)
 * Outputs: log_r[nb] = scale (log prior + log lik - log q) per person; gu[nb][K] = d LOSS / d u (one particle, unaveraged);
 *   attr_out (optional) the draws; gitem = d LOSS / d [g_un: J | s_un: J].  Responses must be complete (0 / 1): the
 *   reference passes them unmasked (vi.py:756).  Item gradients come from integer (eta, y) counts: order-independent. */
int64_t vx_cdm_sf_workspace_floats(const vx_hodina_cfg* cfg, int64_t nb);
int vx_cdm_sf_grad(const vx_hodina_cfg* cfg, int32_t dino, int32_t clamp_t, float prior_p, const uint8_t* y,
                   const int64_t* rows, int64_t nb, int64_t gid0, const float* q /*[K][J]*/, const float* g_un,
                   const float* s_un, const float* u, const uint8_t* attr_in, float* baseline, float base_beta,
                   int32_t base_by_row, float* gu, float* log_r, uint8_t* attr_out, float* gitem, float* workspace,
                   void* hip_stream);
/* leave-one-out control variate over the S >= 2 particles of a step: out[i] = mean_{s' != s} lr_all[s'][i] */
int vx_loo_baseline(const float* lr_all /*[S][nb]*/, int32_t S, int64_t nb, int32_t s, float* out, void* hip_stream);
/* BinEncoder of VaeCDM (vi.py:458-470): h[nb][H] = softplus(W1 yin + b1), u[nb][K] = W2 h + b2 (logits; H <= 128).
 * genc = d LOSS / d [W1: H*J | b1: H | W2: K*H | b2: K] from gu = d LOSS / d u. */
int64_t vx_bin_enc_param_floats(const vx_hodina_cfg* cfg);
int vx_bin_enc_forward(const vx_hodina_cfg* cfg, const uint8_t* y, const int64_t* rows, int64_t nb, const float* W1,
                       const float* b1, const float* W2, const float* b2, float* h, float* u, void* hip_stream);
int64_t vx_bin_enc_bwd_workspace_floats(const vx_hodina_cfg* cfg, int64_t nb);
int vx_bin_enc_backward(const vx_hodina_cfg* cfg, const uint8_t* y, const int64_t* rows, int64_t nb, const float* W2,
                        const float* h, const float* gu, float* genc, float* workspace, void* hip_stream);

/* ---- synthetic response matrices with the distributions of the reference's Random* generators (vi.py:120-412), written
 * straight into the uint8 storage contract; benchmark / test INPUT (never in a timed region).  Every draw is a Philox word
 * keyed by the GLOBAL person id (gid0 + row): a data set does not depend on the sharding.  cfg->seed keys the draws.
 *   vx_synth_irt: y ~ Bern(c + (d - c) sigmoid(Dc (x.a + b))); x = x_in or N(0, 1) drawn here (x_out optional);
 *                 `missing` = MCAR rate of 255 cells.  a [D][J], b / c / d [J] are the CONSTRAINED item parameters.
 *   vx_synth_cdm: DINA / DINO (the reference's dino(), vi.py:86-101) / HO-DINA responses; attributes ~ Bern(attr_p) or
 *                 Bern(sigmoid(theta lam1 + lam0)), theta ~ N(0, 1) (hodina = 1); attr_out [nb][K], theta_out [nb] optional. */
int vx_synth_irt(const vx_irt_cfg* cfg, int64_t nb, int64_t gid0, const float* x_in, const float* a, const float* b,
                 const float* c, const float* d, float missing, uint8_t* y, float* x_out, void* hip_stream);
int vx_synth_cdm(const vx_hodina_cfg* cfg, int32_t dino, int32_t hodina, float attr_p, int64_t nb, int64_t gid0,
                 const float* q, const float* g, const float* s, const float* lam0, const float* lam1, float missing,
                 uint8_t* y, uint8_t* attr_out, float* theta_out, void* hip_stream);

/* ---- VaeCCDM (vi.py:866-891): enumerated DINA / DINO whose pattern prior is Categorical(attr_p[i]), attr_p =
 * softmax(fc2(relu(fc1(data_))), dim = 0) -- the SoftmaxEncoder (vi.py:473-485) normalises over the BATCH -- and whose missing
 * responses stay in the observation as -1 (no mask, vi.py:882-891).  C = 2^K patterns; cfg: K, J, H, scale.  One step =
 *   vx_sm_enc_forward                      h[nb][H] = relu(..), z[nb][C] = fc2 scores
 *   vx_col_reduce(0, z) -> m[C]            column maximum over the batch          [+ all-reduce MAX over the ranks]
 *   vx_col_reduce(1, z, shift = m) -> Z[C] column sum of exp(z - m)               [+ all-reduce SUM]; off = m + log Z (host)
 *   vx_vaeccdm_grad                        elbo[nb], gitem, gla[nb][C] = d ELBO / d log attr_p (scaled)
 *   vx_col_reduce(2, gla) -> T[C]          column sums                            [+ all-reduce SUM]
 *   vx_sm_enc_backward                     genc = d LOSS / d [W1: H*J | b1: H | W2: C*H | b2: C]  (overwrites gla) */
int64_t vx_sm_enc_param_floats(const vx_hodina_cfg* cfg);
int vx_sm_enc_forward(const vx_hodina_cfg* cfg, const uint8_t* y, const int64_t* rows, int64_t nb, const float* W1,
                      const float* b1, const float* W2, const float* b2, float* h, float* z, void* hip_stream);
int64_t vx_col_reduce_workspace_floats(int64_t nb, int32_t C);
int vx_col_reduce(int32_t mode /*0 max, 1 sum exp(v - shift), 2 sum*/, const float* v /*[nb][C]*/, int64_t nb, int32_t C,
                  const float* shift /*[C], mode 1*/, float* out /*[C]*/, float* workspace, void* hip_stream);
int64_t vx_vaeccdm_workspace_floats(const vx_hodina_cfg* cfg, int64_t nb);
int vx_vaeccdm_grad(const vx_hodina_cfg* cfg, int32_t dino, const uint8_t* y, const int64_t* rows, int64_t nb,
                    const float* q, const float* g_un, const float* s_un, const float* z /*[nb][C]*/,
                    const float* off /*[C]*/, float* elbo, float* gla /*[nb][C]*/, float* gitem, float* workspace,
                    void* hip_stream);
int64_t vx_sm_enc_bwd_workspace_floats(const vx_hodina_cfg* cfg, int64_t nb);
int vx_sm_enc_backward(const vx_hodina_cfg* cfg, const uint8_t* y, const int64_t* rows, int64_t nb, const float* W2,
                       const float* h, const float* z, const float* off, const float* T, float* gla, float* genc,
                       float* workspace, void* hip_stream);

/* ---- slab reduction: out[i] = alpha * sum_s slabs[s][i]  (fixed order -> deterministic) */
int vx_reduce_slabs(const float* slabs, int64_t n_slabs, int64_t len, float alpha, float* out,
                    void* hip_stream);
/* out[0] = alpha * sum(v) (fixed-order two-stage tree; used for the loss);
 * workspace: vx_sum_workspace_floats() floats */
int64_t vx_sum_workspace_floats(void);
int vx_sum(const float* v, int64_t n, float alpha, float* out, float* workspace, uint32_t* step_dev /*or NULL: advanced by
           one in the last launch, as vx_sum2 does*/, void* hip_stream);
/* out[0] = alpha * (sum v1 + sum v2), both of length n, in one pass (the loss of the MVN guides: log-lik + entropy).
 * step_dev (or NULL): the device step counter of a captured step (vx_irt_cfg.step_dev): advanced by one here, the last
 * launch of loss-and-gradients, so that vx_adam_step's t_dev may point at the same word (Adam's count is the step + 1). */
int vx_sum2(const float* v1, const float* v2, int64_t n, float alpha, float* out, float* workspace, uint32_t* step_dev,
            void* hip_stream);

/* ---- optimiser: torch.optim.Adam on a flat float32 buffer split into segments with their own
 * learning rate (pyro.optim.Adam with callable optim_args; vi.py:514, test.py:345-350), optional
 * 0/1 `free` mask multiplied into the gradient first (vi.py:511-512).
 * t: the 1-based step count of the bias corrections; t_dev (or NULL): the same count in DEVICE memory, read by the
 * kernel instead of t (captured steps, see vx_irt1d_grad).
 * loss_src / loss_ring (or NULL): the launch also copies *loss_src (the step's loss slot, all-reduced by then) into
 * loss_ring[t % VX_LOSS_RING] -- the value `svi.step` returns (vi.py:516) then stays readable for VX_LOSS_RING - 1 further
 * steps without a copy launch of its own, in eager and in replayed steps alike. */
#define VX_LOSS_RING 64
typedef struct vx_adam_seg { int64_t begin, end; float lr; float _pad; } vx_adam_seg;
int vx_adam_step(float* p, const float* g, float* m, float* v, const float* free_mask /*or NULL*/,
                 int64_t n, const vx_adam_seg* segs /*host*/, int32_t n_segs, int32_t t, const uint32_t* t_dev,
                 float beta1, float beta2, float eps, const float* loss_src, float* loss_ring /*[VX_LOSS_RING]*/,
                 void* hip_stream);
/* the same for TWO buffers in one launch: A with its free mask (the replicated leaves), B without (the per-person rows
 * of a BBVI guide); same t, betas and eps for both */
int vx_adam_step2(float* pA, const float* gA, float* mA, float* vA, const float* freeA /*or NULL*/, int64_t nA,
                  const vx_adam_seg* segsA, int32_t n_segsA, float* pB, const float* gB, float* mB, float* vB, int64_t nB,
                  const vx_adam_seg* segsB, int32_t n_segsB, int32_t t, const uint32_t* t_dev, float beta1, float beta2,
                  float eps, const float* loss_src, float* loss_ring /*[VX_LOSS_RING]*/, void* hip_stream);

/* ---- a D = 1 step on ONE rank with its optimiser: vx_irt1d_grad / vx_irt1d_sparse_grad and vx_adam_step2 in one call,
 * the slab sum and Adam in ONE launch (three graph nodes a step become two; BASELINE config 2: 38.7 -> ~33 us).  What
 * SVI.step does between loss_and_grads and optim(params) (vi.py:505-514) needs no other rank here -- with a process group
 * the all-reduce sits between the two and the separate calls are used.
 *   A = the item leaves: pA / mA / vA / freeA of exactly 4 J floats, gradient = gitem (still written);
 *   B = the per-person rows [loc | raw] (may be empty: nB = 0): gradients gB as written by the step kernel;
 *   t: Adam's 1-based count (ignored with step_dev: the count is then the device counter + 1, which the call advances);
 *   loss_ring (or NULL): loss[0] is also filed in loss_ring[t % VX_LOSS_RING].
 * Same sums in the same order, same Adam arithmetic: the results are those of the separate calls, bit for bit.
 * workspace as for the separate calls (vx_irt1d_workspace_floats / vx_irt1d_sparse_workspace_floats). */
typedef struct vx_adam_tail {
    float* pA; float* mA; float* vA; const float* freeA /*or NULL*/; int64_t nA; const vx_adam_seg* segsA /*host*/;
    float* pB; const float* gB; float* mB; float* vB; int64_t nB; const vx_adam_seg* segsB /*host*/;
    int32_t n_segsA, n_segsB, t;
    float beta1, beta2, eps;
    float* loss_ring /*[VX_LOSS_RING] or NULL*/;
} vx_adam_tail;
int vx_irt1d_grad_adam(const vx_irt_cfg* cfg, const uint8_t* y, const int64_t* rows, int64_t nb, int64_t gid0,
                       const float* loc, const float* raw, const float* eps_in, const float* a, const float* b,
                       const float* c_un, const float* d_un, float* gloc, float* graw, float* elbo, float* gitem,
                       float* loss, uint32_t* step_dev /*or NULL*/, float* workspace, const vx_adam_tail* opt,
                       void* hip_stream);
int vx_irt1d_sparse_grad_adam(const vx_irt_cfg* cfg, const uint16_t* pent, const int32_t* glen, int32_t Lq,
                              const int32_t* pidx, int64_t n_groups, int64_t gid0, const float* loc, const float* raw,
                              const float* eps_in, const float* a, const float* b, const float* c_un, const float* d_un,
                              float* gloc, float* graw, float* elbo, float* gitem, float* loss,
                              uint32_t* step_dev /*or NULL*/, float* workspace, const vx_adam_tail* opt, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* VIPSY_AMD_H */
