"""The training-step engine: what `svi.step(data)` does in the reference (vi.py:503-516), driven
through the C ABI (include/vipsy_amd.h).  One process per GPU; persons are sharded by contiguous
ranges, item / encoder parameters are replicated, and the only exchange per step is one all-reduce
(RCCL via torch.distributed) of the flat gradient buffer [item grads | encoder grads | loss].

torch here is plumbing: it owns device buffers and the process group.  All arithmetic of the step
is in the HIP kernels behind vipsy_amd/_hip.py.
"""
import ctypes
import os
import math

import numpy as np
import torch

from . import _hip

MODEL_CODE = {"irt_1pl": 1, "irt_2pl": 2, "irt_3pl": 3, "irt_4pl": 4}       # vi.py:538-543
LOSS_RING = 64                                                               # VX_LOSS_RING (include/vipsy_amd.h)
ENC_KEYS = ("fc1.weight", "fc1.bias", "fc21.weight", "fc21.bias", "fc22.weight", "fc22.bias")


def _logit(p):
    return math.log(p) - math.log1p(-p)


class HipBackend(object):
    """Thin call layer over the C ABI.  tests/ substitute an oracle-backed object with the same
    methods to exercise the sharding / all-reduce host logic on CPU (gloo)."""

    name = "hip"

    def __init__(self):
        _hip.require_gpu()
        self.L = _hip.lib()

    # -- helpers -----------------------------------------------------------------------------
    @staticmethod
    def cfg(model, D, J, H, Dc, scale, seed, step, stream, step_dev=None, rows_ring=None):
        """step_dev: the device step counter of a captured step (vx_irt_cfg.step_dev) or None; rows_ring: the pinned
        [slots][stride] int64 host tensor a captured subsampled step takes its row indices from (vx_irt_cfg.rows_ring)."""
        c = _hip.IrtCfg(MODEL_CODE[model], D, J, H, Dc, scale, seed, step, stream,
                        None if step_dev is None else step_dev.data_ptr())
        if rows_ring is not None:
            c.rows_ring, c.rows_ring_stride, c.rows_ring_slots = rows_ring.data_ptr(), rows_ring.shape[1], rows_ring.shape[0]
        return c

    def mvn_enc_forward(self, cfg, y, rows, nb, gid0, enc, eps_in, out):
        rc = self.L.vx_mvn_enc_forward(ctypes.byref(cfg), _hip.ptr(y), _hip.ptr(rows), nb, gid0,
                                       _hip.ptr(enc["fc1.weight"]), _hip.ptr(enc["fc1.bias"]),
                                       _hip.ptr(enc["fc21.weight"]), _hip.ptr(enc["fc21.bias"]),
                                       _hip.ptr(enc["fc22.weight"]), _hip.ptr(enc["fc22.bias"]),
                                       _hip.ptr(eps_in), _hip.ptr(out["h"]), _hip.ptr(out["x"]),
                                       _hip.ptr(out["eps"]), _hip.ptr(out["ldT"]), _hip.ptr(out["ent"]),
                                       _hip.ptr(out.get("hT")), _hip.ptr(out.get("epsT")),
                                       _hip.ptr(out.get("packws")), _hip.ptr(out.get("ximg")), _hip.ptr(out.get("hs")),
                                       _hip.stream_ptr())
        _hip.check(rc, "vx_mvn_enc_forward")

    def mvn_pack_floats(self, cfg):
        n = self.L.vx_mvn_pack_floats(ctypes.byref(cfg))
        if n < 0:
            raise _hip.VxError("vx_mvn_pack_floats: unsupported configuration (code %d)" % n)
        return n

    def lik_ximg_bytes(self, cfg, nb):
        n = self.L.vx_irt_lik_ximg_bytes(ctypes.byref(cfg), nb)
        if n < 0:
            raise _hip.VxError("vx_irt_lik_ximg_bytes: unsupported configuration (code %d)" % n)
        return n

    def lik_workspace(self, cfg, nb):
        n = self.L.vx_irt_lik_workspace_floats(ctypes.byref(cfg), nb)
        if n < 0:
            raise _hip.VxError("vx_irt_lik_workspace_floats: unsupported configuration (code %d)" % n)
        return n

    def mvn_pack_opmax_offset(self, cfg, nb):
        return int(self.L.vx_mvn_pack_opmax_offset(ctypes.byref(cfg), nb))

    def lik_grad(self, cfg, y, rows, nb, x, a, b, c_un, d_un, gx, ll, gitem, ws, gxT=None, yT=None, ximg=None,
                 epsT=None, ldT=None, gdT=None, opmax=None):
        rc = self.L.vx_irt_lik_grad(ctypes.byref(cfg), _hip.ptr(y), _hip.ptr(rows), nb, _hip.ptr(x), _hip.ptr(a),
                                    _hip.ptr(b), _hip.ptr(c_un), _hip.ptr(d_un), _hip.ptr(gx), _hip.ptr(gxT),
                                    _hip.ptr(ll), _hip.ptr(gitem), _hip.ptr(ws), _hip.ptr(yT),
                                    int(yT.shape[1]) if yT is not None else 0, _hip.ptr(ximg), _hip.ptr(epsT),
                                    _hip.ptr(ldT), _hip.ptr(gdT), _hip.ptr(opmax), _hip.stream_ptr())
        _hip.check(rc, "vx_irt_lik_grad")

    def mvn_enc_bwd_workspace(self, cfg, nb):
        n = self.L.vx_mvn_enc_bwd_workspace_floats(ctypes.byref(cfg), nb)
        if n < 0:
            raise _hip.VxError("vx_mvn_enc_bwd_workspace_floats: unsupported configuration (code %d)" % n)
        return n

    def mvn_enc_bwd_layout(self, cfg, nb):
        return int(self.L.vx_mvn_enc_bwd_layout(ctypes.byref(cfg), nb))

    def mvn_enc_bwd_gd_offset(self, cfg, nb):
        return int(self.L.vx_mvn_enc_bwd_gd_offset(ctypes.byref(cfg), nb))

    def mvn_enc_bwd_hs_offset(self, cfg, nb):
        return int(self.L.vx_mvn_enc_bwd_hs_offset(ctypes.byref(cfg), nb))

    def mvn_enc_backward(self, cfg, y, rows, nb, enc, fw, gx, genc, ws, gxT=None, gd_ready=False, opmax_ready=False, loss=None):
        """loss = (ll, ent, alpha, slot, sum_ws): the step's loss from the call's last launch (vx_mvn_enc_backward_loss)."""
        args = [ctypes.byref(cfg), _hip.ptr(y), _hip.ptr(rows), nb,
                _hip.ptr(enc["fc21.weight"]), _hip.ptr(enc["fc22.weight"]),
                _hip.ptr(fw["h"]), _hip.ptr(fw["eps"]), _hip.ptr(fw["ldT"]), _hip.ptr(gx),
                _hip.ptr(fw.get("hT")), _hip.ptr(fw.get("epsT")), _hip.ptr(gxT),
                _hip.ptr(fw.get("yT")), int(fw["yT"].shape[1]) if fw.get("yT") is not None else 0,
                _hip.ptr(genc), _hip.ptr(ws), _hip.ptr(fw.get("packws")),
                int(bool(gd_ready)) | (2 if fw.get("hs") is not None else 0) | (4 if opmax_ready else 0)]
        if loss is None:
            rc = self.L.vx_mvn_enc_backward(*args, _hip.stream_ptr())
            _hip.check(rc, "vx_mvn_enc_backward")
            return
        ll, ent, alpha, slot, sum_ws = loss
        rc = self.L.vx_mvn_enc_backward_loss(*args, _hip.ptr(ll), _hip.ptr(ent), alpha, _hip.ptr(slot), _hip.ptr(sum_ws),
                                             _hip.stream_ptr())
        _hip.check(rc, "vx_mvn_enc_backward_loss")

    def irt1d_workspace(self, cfg, nb):
        n = self.L.vx_irt1d_workspace_floats(ctypes.byref(cfg), nb)
        if n < 0:
            raise _hip.VxError("vx_irt1d_workspace_floats: unsupported configuration (code %d)" % n)
        return n

    @staticmethod
    def _adam_tail(adam):
        """struct vx_adam_tail from the engine's description of its optimiser step (_EngineBase._fused_tail_args); the
        segment arrays ride along so that they outlive the call."""
        (pA, mA, vA, freeA, nA, segsA), B = adam["A"], adam.get("B")
        arrA = (_hip.AdamSeg * len(segsA))(*[_hip.AdamSeg(b, e, lr, 0.0) for (b, e, lr) in segsA])
        t = _hip.AdamTail()
        t.pA, t.mA, t.vA, t.freeA, t.nA, t.segsA, t.n_segsA = _hip.ptr(pA), _hip.ptr(mA), _hip.ptr(vA), _hip.ptr(freeA), nA, arrA, len(segsA)
        keep = [arrA]
        if B is not None:
            pB, gB, mB, vB, nB, segsB = B
            arrB = (_hip.AdamSeg * len(segsB))(*[_hip.AdamSeg(b, e, lr, 0.0) for (b, e, lr) in segsB])
            t.pB, t.gB, t.mB, t.vB, t.nB, t.segsB, t.n_segsB = _hip.ptr(pB), _hip.ptr(gB), _hip.ptr(mB), _hip.ptr(vB), nB, arrB, len(segsB)
            keep.append(arrB)
        t.t, (t.beta1, t.beta2), t.eps = adam["t"], adam["betas"], adam["eps"]
        t.loss_ring = _hip.ptr(adam.get("loss_ring"))
        return t, keep

    def irt1d_grad(self, cfg, y, rows, nb, gid0, loc, raw, eps_in, a, b, c_un, d_un, gloc, graw, elbo, gitem, ws,
                   loss=None, step_dev=None, adam=None):
        args = [ctypes.byref(cfg), _hip.ptr(y), _hip.ptr(rows), nb, gid0, _hip.ptr(loc),
                _hip.ptr(raw), _hip.ptr(eps_in), _hip.ptr(a), _hip.ptr(b), _hip.ptr(c_un),
                _hip.ptr(d_un), _hip.ptr(gloc), _hip.ptr(graw), _hip.ptr(elbo), _hip.ptr(gitem),
                _hip.ptr(loss), _hip.ptr(step_dev), _hip.ptr(ws)]
        if adam is not None:                                 # the optimiser in the step's last launch (vx_irt1d_grad_adam)
            tail, _keep = self._adam_tail(adam)
            rc = self.L.vx_irt1d_grad_adam(*args, ctypes.byref(tail), _hip.stream_ptr())
            _hip.check(rc, "vx_irt1d_grad_adam")
            return
        rc = self.L.vx_irt1d_grad(*args, _hip.stream_ptr())
        _hip.check(rc, "vx_irt1d_grad")

    def irt1d_sparse_workspace(self, cfg, n_groups):
        n = self.L.vx_irt1d_sparse_workspace_floats(ctypes.byref(cfg), n_groups)
        if n < 0:
            raise _hip.VxError("vx_irt1d_sparse_workspace_floats: unsupported configuration (code %d)" % n)
        return n

    def irt1d_sparse_grad(self, cfg, lists, gid0, loc, raw, eps_in, a, b, c_un, d_un, gloc, graw, elbo, gitem, ws,
                          loss=None, step_dev=None, adam=None):
        args = [ctypes.byref(cfg), _hip.ptr(lists["pent"]), _hip.ptr(lists["glen"]),
                int(lists["Lq"]), _hip.ptr(lists["pidx"]), int(lists["n_groups"]), gid0,
                _hip.ptr(loc), _hip.ptr(raw), _hip.ptr(eps_in), _hip.ptr(a), _hip.ptr(b),
                _hip.ptr(c_un), _hip.ptr(d_un), _hip.ptr(gloc), _hip.ptr(graw), _hip.ptr(elbo),
                _hip.ptr(gitem), _hip.ptr(loss), _hip.ptr(step_dev), _hip.ptr(ws)]
        if adam is not None:
            tail, _keep = self._adam_tail(adam)
            rc = self.L.vx_irt1d_sparse_grad_adam(*args, ctypes.byref(tail), _hip.stream_ptr())
            _hip.check(rc, "vx_irt1d_sparse_grad_adam")
            return
        rc = self.L.vx_irt1d_sparse_grad(*args, _hip.stream_ptr())
        _hip.check(rc, "vx_irt1d_sparse_grad")

    def mvn_bbvi_forward(self, cfg, nb, rows, gid0, loc, M, shared, eps_in, x, eps, ent):
        rc = self.L.vx_mvn_bbvi_forward(ctypes.byref(cfg), nb, _hip.ptr(rows), gid0, _hip.ptr(loc), _hip.ptr(M),
                                        int(shared), _hip.ptr(eps_in), _hip.ptr(x), _hip.ptr(eps), _hip.ptr(ent),
                                        _hip.stream_ptr())
        _hip.check(rc, "vx_mvn_bbvi_forward")

    def mvn_bbvi_bwd_workspace(self, cfg, nb, shared):
        return max(1, int(self.L.vx_mvn_bbvi_bwd_workspace_floats(ctypes.byref(cfg), nb, int(shared))))

    def mvn_bbvi_backward(self, cfg, nb, rows, M, shared, gx, eps, gloc, gM, ws=None):
        rc = self.L.vx_mvn_bbvi_backward(ctypes.byref(cfg), nb, _hip.ptr(rows), _hip.ptr(M), int(shared), _hip.ptr(gx),
                                         _hip.ptr(eps), _hip.ptr(gloc), _hip.ptr(gM), _hip.ptr(ws), _hip.stream_ptr())
        _hip.check(rc, "vx_mvn_bbvi_backward")

    def norm_enc_forward(self, cfg, y, rows, nb, enc, out):
        rc = self.L.vx_norm_enc_forward(ctypes.byref(cfg), _hip.ptr(y), _hip.ptr(rows), nb,
                                        _hip.ptr(enc["fc1.weight"]), _hip.ptr(enc["fc1.bias"]),
                                        _hip.ptr(enc["fc21.weight"]), _hip.ptr(enc["fc21.bias"]),
                                        _hip.ptr(enc["fc22.weight"]), _hip.ptr(enc["fc22.bias"]),
                                        _hip.ptr(out["h"]), _hip.ptr(out["loc"]), _hip.ptr(out["raw"]),
                                        _hip.ptr(out.get("packws")), _hip.stream_ptr())
        _hip.check(rc, "vx_norm_enc_forward")

    def norm_enc_pack_floats(self, cfg):
        n = self.L.vx_norm_enc_pack_floats(ctypes.byref(cfg))
        if n < 0:
            raise _hip.VxError("vx_norm_enc_pack_floats: unsupported configuration (code %d)" % n)
        return n

    def norm_enc_bwd_workspace(self, cfg, nb):
        n = self.L.vx_norm_enc_bwd_workspace_floats(ctypes.byref(cfg), nb)
        if n < 0:
            raise _hip.VxError("vx_norm_enc_bwd_workspace_floats: unsupported configuration (code %d)" % n)
        return n

    def norm_enc_backward(self, cfg, y, rows, nb, enc, h, gloc, graw, genc, ws, yT=None):
        rc = self.L.vx_norm_enc_backward(ctypes.byref(cfg), _hip.ptr(y), _hip.ptr(rows), nb,
                                         _hip.ptr(enc["fc21.weight"]), _hip.ptr(enc["fc22.weight"]), _hip.ptr(h),
                                         _hip.ptr(gloc), _hip.ptr(graw), _hip.ptr(yT),
                                         int(yT.shape[1]) if yT is not None else 0, _hip.ptr(genc), _hip.ptr(ws),
                                         _hip.stream_ptr())
        _hip.check(rc, "vx_norm_enc_backward")

    @staticmethod
    def hodina_cfg(K, J, H, scale, seed, step, stream, step_dev=None):
        return _hip.HoDinaCfg(K, J, H, 0, scale, 0.0, seed, step, stream, None if step_dev is None else step_dev.data_ptr())

    def hodina_workspace(self, cfg, nb):
        n = self.L.vx_hodina_workspace_floats(ctypes.byref(cfg), nb)
        if n < 0:
            raise _hip.VxError("vx_hodina_workspace_floats: unsupported configuration (code %d)" % n)
        return n

    def hodina_grad(self, cfg, y, rows, nb, gid0, loc, raw, eps_in, q, lam0, lam1, g, s_, gloc, graw, elbo, gitem, ws):
        rc = self.L.vx_hodina_grad(ctypes.byref(cfg), _hip.ptr(y), _hip.ptr(rows), nb, gid0, _hip.ptr(loc),
                                   _hip.ptr(raw), _hip.ptr(eps_in), _hip.ptr(q), _hip.ptr(lam0), _hip.ptr(lam1),
                                   _hip.ptr(g), _hip.ptr(s_), _hip.ptr(gloc), _hip.ptr(graw), _hip.ptr(elbo),
                                   _hip.ptr(gitem), _hip.ptr(ws), _hip.stream_ptr())
        _hip.check(rc, "vx_hodina_grad")

    def ccdm_workspace(self, cfg, nb):
        n = self.L.vx_ccdm_workspace_floats(ctypes.byref(cfg), nb)
        if n < 0:
            raise _hip.VxError("vx_ccdm_workspace_floats: unsupported configuration (code %d)" % n)
        return n

    def ccdm_grad(self, cfg, dino, y, rows, nb, q, g, s_, elbo, gitem, ws):
        rc = self.L.vx_ccdm_grad(ctypes.byref(cfg), int(dino), _hip.ptr(y), _hip.ptr(rows), nb, _hip.ptr(q), _hip.ptr(g),
                                 _hip.ptr(s_), _hip.ptr(elbo), _hip.ptr(gitem), _hip.ptr(ws), _hip.stream_ptr())
        _hip.check(rc, "vx_ccdm_grad")

    def cdm_sf_workspace(self, cfg, nb):
        n = self.L.vx_cdm_sf_workspace_floats(ctypes.byref(cfg), nb)
        if n < 0:
            raise _hip.VxError("vx_cdm_sf_workspace_floats: unsupported configuration (code %d)" % n)
        return n

    def cdm_sf_grad(self, cfg, dino, clamp_t, prior_p, y, rows, nb, gid0, q, g, s_, u, attr_in, baseline, base_beta,
                    base_by_row, gu, log_r, attr_out, gitem, ws):
        rc = self.L.vx_cdm_sf_grad(ctypes.byref(cfg), int(dino), int(clamp_t), float(prior_p), _hip.ptr(y), _hip.ptr(rows), nb,
                                   gid0, _hip.ptr(q), _hip.ptr(g), _hip.ptr(s_), _hip.ptr(u), _hip.ptr(attr_in),
                                   _hip.ptr(baseline), float(base_beta), int(base_by_row), _hip.ptr(gu), _hip.ptr(log_r),
                                   _hip.ptr(attr_out), _hip.ptr(gitem), _hip.ptr(ws), _hip.stream_ptr())
        _hip.check(rc, "vx_cdm_sf_grad")

    def irt1d_score_grad(self, nb, scale, elbo, eps, raw, rows, baseline, base_beta, base_by_row, log_r, gloc, graw):
        rc = self.L.vx_irt1d_score_grad(nb, float(scale), _hip.ptr(elbo), _hip.ptr(eps), _hip.ptr(raw), _hip.ptr(rows),
                                        _hip.ptr(baseline), float(base_beta), int(base_by_row), _hip.ptr(log_r),
                                        _hip.ptr(gloc), _hip.ptr(graw), _hip.stream_ptr())
        _hip.check(rc, "vx_irt1d_score_grad")

    def mvn_score_operands(self, cfg, nb, rows, kind, h, W22, b22, M, eps, ll, ent, baseline, base_beta, base_by_row, log_r, w,
                           gx, gxT, gdT):
        rc = self.L.vx_mvn_score_operands(ctypes.byref(cfg), nb, _hip.ptr(rows), int(kind), _hip.ptr(h), _hip.ptr(W22),
                                          _hip.ptr(b22), _hip.ptr(M), _hip.ptr(eps), _hip.ptr(ll), _hip.ptr(ent),
                                          _hip.ptr(baseline), float(base_beta), int(base_by_row), _hip.ptr(log_r), _hip.ptr(w),
                                          _hip.ptr(gx), _hip.ptr(gxT), _hip.ptr(gdT), _hip.stream_ptr())
        _hip.check(rc, "vx_mvn_score_operands")

    def mvn_score_heads_workspace(self, cfg):
        """floats of the column image of the MFMA score kernel, or -1 when the shape is not its own"""
        return int(self.L.vx_mvn_score_heads_workspace_floats(ctypes.byref(cfg)))

    def mvn_score_heads(self, cfg, nb, rows, h, W22, b22, packws, eps, ll, ent, baseline, base_beta, base_by_row, log_r,
                        gxT, gdT, ws):
        rc = self.L.vx_mvn_score_heads(ctypes.byref(cfg), nb, _hip.ptr(rows), _hip.ptr(h), _hip.ptr(W22), _hip.ptr(b22),
                                       _hip.ptr(packws), _hip.ptr(eps), _hip.ptr(ll), _hip.ptr(ent),
                                       _hip.ptr(baseline), float(base_beta), int(base_by_row), _hip.ptr(log_r), _hip.ptr(gxT),
                                       _hip.ptr(gdT), _hip.ptr(ws), _hip.stream_ptr())
        _hip.check(rc, "vx_mvn_score_heads")

    def mvn_score_diag(self, cfg, nb, rows, w, shared, gM):
        _hip.check(self.L.vx_mvn_score_diag(ctypes.byref(cfg), nb, _hip.ptr(rows), _hip.ptr(w), int(bool(shared)), _hip.ptr(gM),
                                            _hip.stream_ptr()), "vx_mvn_score_diag")

    def loo_baseline(self, lr_all, S, nb, s, out):
        _hip.check(self.L.vx_loo_baseline(_hip.ptr(lr_all), int(S), nb, int(s), _hip.ptr(out), _hip.stream_ptr()),
                   "vx_loo_baseline")

    def bin_enc_forward(self, cfg, y, rows, nb, enc, h, u):
        rc = self.L.vx_bin_enc_forward(ctypes.byref(cfg), _hip.ptr(y), _hip.ptr(rows), nb, _hip.ptr(enc["fc1.weight"]),
                                       _hip.ptr(enc["fc1.bias"]), _hip.ptr(enc["fc2.weight"]), _hip.ptr(enc["fc2.bias"]),
                                       _hip.ptr(h), _hip.ptr(u), _hip.stream_ptr())
        _hip.check(rc, "vx_bin_enc_forward")

    def bin_enc_bwd_workspace(self, cfg, nb):
        n = self.L.vx_bin_enc_bwd_workspace_floats(ctypes.byref(cfg), nb)
        if n < 0:
            raise _hip.VxError("vx_bin_enc_bwd_workspace_floats: unsupported configuration (code %d)" % n)
        return n

    def bin_enc_backward(self, cfg, y, rows, nb, enc, h, gu, genc, ws):
        rc = self.L.vx_bin_enc_backward(ctypes.byref(cfg), _hip.ptr(y), _hip.ptr(rows), nb, _hip.ptr(enc["fc2.weight"]),
                                        _hip.ptr(h), _hip.ptr(gu), _hip.ptr(genc), _hip.ptr(ws), _hip.stream_ptr())
        _hip.check(rc, "vx_bin_enc_backward")

    def sm_enc_forward(self, cfg, y, rows, nb, enc, h, z):
        rc = self.L.vx_sm_enc_forward(ctypes.byref(cfg), _hip.ptr(y), _hip.ptr(rows), nb, _hip.ptr(enc["fc1.weight"]),
                                      _hip.ptr(enc["fc1.bias"]), _hip.ptr(enc["fc2.weight"]), _hip.ptr(enc["fc2.bias"]),
                                      _hip.ptr(h), _hip.ptr(z), _hip.stream_ptr())
        _hip.check(rc, "vx_sm_enc_forward")

    def col_reduce(self, mode, v, nb, C, shift, out, ws):
        _hip.check(self.L.vx_col_reduce(int(mode), _hip.ptr(v), nb, int(C), _hip.ptr(shift), _hip.ptr(out), _hip.ptr(ws),
                                        _hip.stream_ptr()), "vx_col_reduce")

    def col_reduce_workspace(self, nb, C):
        return int(self.L.vx_col_reduce_workspace_floats(nb, int(C)))

    def vaeccdm_workspace(self, cfg, nb):
        n = self.L.vx_vaeccdm_workspace_floats(ctypes.byref(cfg), nb)
        if n < 0:
            raise _hip.VxError("vx_vaeccdm_workspace_floats: unsupported configuration (code %d)" % n)
        return n

    def vaeccdm_grad(self, cfg, dino, y, rows, nb, q, g, s_, z, off, elbo, gla, gitem, ws):
        rc = self.L.vx_vaeccdm_grad(ctypes.byref(cfg), int(dino), _hip.ptr(y), _hip.ptr(rows), nb, _hip.ptr(q), _hip.ptr(g),
                                    _hip.ptr(s_), _hip.ptr(z), _hip.ptr(off), _hip.ptr(elbo), _hip.ptr(gla), _hip.ptr(gitem),
                                    _hip.ptr(ws), _hip.stream_ptr())
        _hip.check(rc, "vx_vaeccdm_grad")

    def sm_enc_bwd_workspace(self, cfg, nb):
        n = self.L.vx_sm_enc_bwd_workspace_floats(ctypes.byref(cfg), nb)
        if n < 0:
            raise _hip.VxError("vx_sm_enc_bwd_workspace_floats: unsupported configuration (code %d)" % n)
        return n

    def sm_enc_backward(self, cfg, y, rows, nb, enc, h, z, off, T, gla, genc, ws):
        rc = self.L.vx_sm_enc_backward(ctypes.byref(cfg), _hip.ptr(y), _hip.ptr(rows), nb, _hip.ptr(enc["fc2.weight"]),
                                       _hip.ptr(h), _hip.ptr(z), _hip.ptr(off), _hip.ptr(T), _hip.ptr(gla), _hip.ptr(genc),
                                       _hip.ptr(ws), _hip.stream_ptr())
        _hip.check(rc, "vx_sm_enc_backward")

    def sum_into(self, v, n, alpha, out, ws, step_dev=None):
        rc = self.L.vx_sum(_hip.ptr(v), n, alpha, _hip.ptr(out), _hip.ptr(ws), _hip.ptr(step_dev), _hip.stream_ptr())
        _hip.check(rc, "vx_sum")

    def adam2(self, bufA, free, nA, segsA, bufB, nB, segsB, t, betas=(0.9, 0.999), eps=1e-8, t_dev=None, loss=None):
        arrA = (_hip.AdamSeg * len(segsA))(*[_hip.AdamSeg(b, e, lr, 0.0) for (b, e, lr) in segsA])
        arrB = (_hip.AdamSeg * len(segsB))(*[_hip.AdamSeg(b, e, lr, 0.0) for (b, e, lr) in segsB])
        src, ring = loss if loss is not None else (None, None)
        rc = self.L.vx_adam_step2(*[_hip.ptr(x) for x in bufA], _hip.ptr(free), nA, arrA, len(segsA),
                                  *[_hip.ptr(x) for x in bufB], nB, arrB, len(segsB), t, _hip.ptr(t_dev), betas[0], betas[1],
                                  eps, _hip.ptr(src), _hip.ptr(ring), _hip.stream_ptr())
        _hip.check(rc, "vx_adam_step2")

    def sum2_into(self, v1, v2, n, alpha, out, ws, step_dev=None):
        rc = self.L.vx_sum2(_hip.ptr(v1), _hip.ptr(v2), n, alpha, _hip.ptr(out), _hip.ptr(ws), _hip.ptr(step_dev),
                            _hip.stream_ptr())
        _hip.check(rc, "vx_sum2")

    def adam(self, p, g, m, v, free, n, segs, t, betas=(0.9, 0.999), eps=1e-8, t_dev=None, loss=None):
        arr = (_hip.AdamSeg * len(segs))(*[_hip.AdamSeg(b, e, lr, 0.0) for (b, e, lr) in segs])
        src, ring = loss if loss is not None else (None, None)
        rc = self.L.vx_adam_step(_hip.ptr(p), _hip.ptr(g), _hip.ptr(m), _hip.ptr(v), _hip.ptr(free), n, arr,
                                 len(segs), t, _hip.ptr(t_dev), betas[0], betas[1], eps, _hip.ptr(src), _hip.ptr(ring),
                                 _hip.stream_ptr())
        _hip.check(rc, "vx_adam_step")

    def philox_normals(self, out, gids, gid0, n, D, seed, step, stream):
        rc = self.L.vx_philox_normals(_hip.ptr(out), _hip.ptr(gids), gid0, n, D, seed, step, stream,
                                      _hip.stream_ptr())
        _hip.check(rc, "vx_philox_normals")


_ADAM_NOOP = {"weight_decay": 0, "amsgrad": False, "maximize": False, "foreach": None, "capturable": False,
              "differentiable": False, "fused": None}


class LrSpec(object):
    """Per-tensor learning rates + MultiStepLR, as pyro.optim.Adam / PyroLRScheduler give them
    (vi.py:514, 639-640; test.py:321-327, 345-359)."""

    def __init__(self, lr, milestones=(), gamma=0.1, betas=(0.9, 0.999), eps=1e-8):
        self.lr, self.milestones, self.gamma, self.betas, self.eps = lr, tuple(milestones), gamma, betas, eps
        self.epoch = 0

    def args_of(self, name):
        """The torch.optim.Adam kwargs of one tensor: `lr` may be a float, a dict, or pyro's
        callable(module_name, param_name) -> dict; betas / eps ride in the same dict (torch defaults otherwise)."""
        if callable(self.lr):
            module = name.split("$$$")[0]
            stripped = name.split("$$$")[1] if "$$$" in name else name
            d = dict(self.lr(module, stripped))
        elif isinstance(self.lr, dict):
            d = dict(self.lr)
        else:
            d = {"lr": self.lr}
        for k, noop in _ADAM_NOOP.items():                 # torch.optim.Adam defaults spelled out change nothing
            if k in d and (d[k] is None if noop is None else d[k] == noop):
                del d[k]
        unknown = set(d) - {"lr", "betas", "eps"}
        if unknown:
            raise NotImplementedError("Adam option(s) %s are not on the HIP path" % sorted(unknown))
        return d

    def lr_of(self, name):
        base = self.args_of(name)["lr"]
        k = sum(1 for m in self.milestones if m <= self.epoch)
        return base * self.gamma ** k

    def hyper_of(self, name):
        d = self.args_of(name)
        b = d.get("betas", self.betas)
        return (float(b[0]), float(b[1])), float(d.get("eps", self.eps))

    def scheduler_step(self):
        self.epoch += 1


class _EngineBase(object):
    """State and optimiser plumbing shared by the IRT and HO-DINA engines.

    Replicated parameters live in ONE flat float32 buffer `P` (subclasses define the segments in
    `self.off` / `self.shape`); the gradient buffer `G` has one extra trailing slot for the loss so a
    single all-reduce carries everything.  Per-person variational rows (BBVI
    guides) live in `PP` = [loc: n | raw: n], sharded with the persons, never reduced across ranks."""

    pp_names = ("x_local", "x_scale")

    def _alloc(self, n_params, n_local, per_person):
        f32 = dict(dtype=torch.float32, device=self.dev)
        self.n_params = n_params
        self.P = torch.zeros(n_params, **f32)
        self.G = torch.zeros(n_params + 1, **f32)
        self.M = torch.zeros(n_params, **f32)
        self.V = torch.zeros(n_params, **f32)
        self.free = torch.ones(n_params, **f32)
        self.per_person = per_person
        if per_person:
            if not hasattr(self, "pp_shape"):
                self.pp_shape = {self.pp_names[0]: (n_local, 1), self.pp_names[1]: (n_local, 1)}
            self.pp_off, o = {}, 0
            for nme in self.pp_names:
                if nme in self.pp_shape:
                    self.pp_off[nme] = o
                    o += int(np.prod(self.pp_shape[nme]))
            self.pp_len = o
            self.PP = torch.zeros(self.pp_len, **f32)
            self.GP = torch.zeros(self.pp_len, **f32)
            self.MP = torch.zeros(self.pp_len, **f32)
            self.VP = torch.zeros(self.pp_len, **f32)
        self._ws = {}
        # the losses of the last LOSS_RING steps: what step() returns is slot t % LOSS_RING, written by the optimiser launch
        self.loss_ring = torch.zeros(LOSS_RING, **f32)
        self.sum_ws = torch.empty(1024, **f32)
        self.last = {}
        self.events = None               # bench.py: list collecting (phase, start_event, end_event)
        self.t = 0                       # optimiser step counter (Adam bias correction, Philox step)

    def view(self, name, buf=None):
        buf = self.P if buf is None else buf
        o = self.off[name]
        return buf[o:o + int(np.prod(self.shape[name]))]

    def unconstrained(self, name, buf=None):
        if self.per_person and name in self.pp_off:
            b = self.PP if buf is None else buf
            o = self.pp_off[name]
            return b[o:o + int(np.prod(self.pp_shape[name]))].reshape(self.pp_shape[name])
        return self.view(name, buf).reshape(self.shape[name])

    def _enc(self):
        return {k: self.view("encoder$$$" + k) for k in ENC_KEYS}

    def _item_major_y(self, rows):
        """Item-major copy of the responses ([J + 1][stride] bytes, stride = n_local rounded up to 64; row J and the
        columns past n_local hold 254 = "outside the problem") for the dimension-major fc1 gradient kernel and the bf16-MFMA
        likelihood kernel (include/vipsy_amd.h, vx_irt_lik_grad): made once (the responses never change); full batches only."""
        if rows is not None or self.n_local == 0:
            return None
        if getattr(self, "_yT", None) is None:
            y = self._padded_y(lik=True)                     # (the shard's own responses unless persons / items are padded)
            stride = (y.shape[0] + 63) // 64 * 64
            yT = torch.full((self.J + 1, stride), 254, dtype=torch.uint8, device=self.dev)
            yT[:self.J, :y.shape[0]] = y.t()
            self._yT = yT
        return self._yT

    def _pad_persons(self, be, cfg, rows):
        """Persons the full-batch kernels of the amortized multivariate guide are launched over: the shard's own count, or --
        when that is not a multiple of 8 -- the next one.  The dimension-major operands of the large-batch kernels are rows of
        `nb` floats (fp16 planes: `nb` halves) read in 16-byte pieces, so a count that is not a multiple of 8 sends the head
        weight gradient (and, below a multiple of 4, everything) to the fp32-MFMA generation: 13.9 ms a step at 1 000 004
        persons, 26.3 at 999 999, against 9.1 at 1 000 000.  The 1-7 PHANTOM persons appended instead have every response
        missing (byte 255: a finite guide sample, no likelihood term) and are taken out of the step behind the likelihood:
        their columns of gxT / gdT (everything the backward kernels see of a person) and their entries of ll / ent are zeroed,
        so they add nothing to any gradient or to the loss.  Pathwise estimator, HIP backend, full batch only."""
        n = self.n_local
        if (rows is not None or n % 8 == 0 or n < 8 or not isinstance(be, HipBackend)
                or getattr(self, "estimator", "pathwise") != "pathwise" or not self.pad_persons):
            return n
        n_pad = (n + 7) // 8 * 8
        return n_pad if (be.mvn_enc_bwd_layout(cfg, n_pad) == 1 and be.mvn_enc_bwd_gd_offset(cfg, n_pad) >= 0) else n

    pad_persons = os.environ.get("VX_PAD_PERSONS", "1") != "0"      # test seam: 0 = launch over the shard's own count

    def _padded_y(self, lik=False):
        """The responses with the phantom persons of _pad_persons appended (made once; the responses themselves when none
        are).  lik: the copy the likelihood reads (phantom ITEMS absent instead of zero, IrtEngine.__init__)."""
        y = self.y_lik if (lik and getattr(self, "y_lik", None) is not None) else self.y
        n_pad = getattr(self, "_n_pad", self.n_local)
        if n_pad == self.n_local:
            return y
        key = "_y_pad_lik" if y is not self.y else "_y_pad"
        if getattr(self, key, None) is None:
            setattr(self, key, torch.cat([y, torch.full((n_pad - self.n_local, self.J), 255, dtype=torch.uint8, device=self.dev)]))
        return getattr(self, key)

    def _sparse_lists(self, rows):
        """Observed-cell lists for the D = 1 kernel (include/vipsy_amd.h, vx_irt1d_sparse_grad): built once -- the
        responses never change -- and only when most cells are missing and the batch is the whole shard.  Persons are
        placed in slots sorted by list length inside windows of 4096, so the 64 lists a wave walks end together."""
        if rows is not None or self.n_local == 0 or self.J > 1024 or not self.observed_lists:
            return None
        if getattr(self, "_sp", None) is None:
            y, n, J = (self.y_lik if getattr(self, "y_lik", None) is not None else self.y), self.n_local, self.J
            frac = float((y == 255).sum().item()) / float(n * J)
            if frac < 0.5:
                self._sp = False
            else:
                ng = (n + 63) // 64
                cnt = torch.full((ng * 64,), -1, dtype=torch.int64, device=self.dev)             # -1: empty slot, sorts last
                cnt[:n] = (y < 254).sum(1)                                                     # (254: a phantom item, absent)
                slot = torch.arange(ng * 64, device=self.dev)
                perm = torch.argsort((slot // 4096) * (J + 2) + (J - cnt), stable=True)           # slot -> person
                cnt_s = cnt[perm].clamp_(min=0)
                pidx = torch.where(perm < n, perm, torch.full_like(perm, -1)).to(torch.int32).contiguous()
                Lq = max(1, (int(cnt_s.max().item()) + 3) // 4)
                W = min(J, 4 * Lq)
                pent = torch.full((ng * 64, 4 * Lq), -1, dtype=torch.int16, device=self.dev)      # 0xFFFF = past the end
                ar = torch.arange(W, device=self.dev)[None, :]
                for lo in range(0, ng * 64, 65536):                                               # bounded temporaries
                    hi = min(ng * 64, lo + 65536)
                    yc = y[perm[lo:hi].clamp(max=n - 1)]
                    order = torch.argsort((yc >= 254).to(torch.uint8), dim=1, stable=True)[:, :W]  # observed items first
                    code = order.to(torch.int32) | ((torch.gather(yc, 1, order) == 1).to(torch.int32) << 15)
                    code = torch.where(ar < cnt_s[lo:hi, None], code, torch.full_like(code, 0xFFFF))
                    pent[lo:hi, :W] = code.to(torch.int16)                                        # wraps: bit pattern kept
                pent = pent.reshape(ng, 64, Lq, 4).permute(0, 2, 1, 3).contiguous()
                glen = ((cnt_s.reshape(ng, 64).max(1).values + 3) // 4).to(torch.int32).contiguous()
                self._sp = {"pent": pent, "glen": glen, "Lq": Lq, "pidx": pidx, "n_groups": ng, "missing": frac}
        return self._sp or None

    def _buf(self, key, n):
        t = self._ws.get(key)
        if t is None or t.numel() < n:
            t = torch.empty(max(int(n), 1), dtype=torch.float32, device=self.dev)
            self._ws[key] = t
        return t

    def _phase(self, name):
        return _Phase(self.events, name)

    def _gather_pp(self, rows, nb):
        """loc/raw (and their gradient targets) of the batch rows of a per-person guide."""
        n = self.n_local
        if rows is None:
            return self.PP[:n], self.PP[n:], self.GP[:n], self.GP[n:]
        return (self.PP[:n][rows].contiguous(), self.PP[n:][rows].contiguous(),
                self._buf("gloc", nb), self._buf("graw", nb))

    def _scatter_pp(self, rows, nb, gloc, graw):
        """Dense per-person gradients: zero off the batch, exactly what autograd hands the per-tensor
        Adam of the reference."""
        if rows is not None:
            n = self.n_local
            self.GP.zero_()
            self.GP[:n].index_add_(0, rows, gloc[:nb])
            self.GP[n:].index_add_(0, rows, graw[:nb])

    def allreduce(self):
        """The one exchange of a step: SUM of [gradients | loss] over the ranks that share this problem.  Sharding is
        explicit -- `group` handed to the constructor -- never inferred from global torch.distributed state, so that
        independent problems (replications, vipsy_amd/harness.py) can run side by side under one process group."""
        if self.group is not None:
            if self.G.is_cuda and torch.distributed.get_backend(self.group) == "gloo":
                host = self.G.cpu()                          # rehearsal of N ranks on fewer GPUs: gloo reduces on the host
                torch.distributed.all_reduce(host, group=self.group)
                self.G.copy_(host)
            else:                                            # the product path: RCCL (backend "nccl") on the device buffer
                torch.distributed.all_reduce(self.G, group=self.group)

    def apply_optim(self, lrs):
        """Adam on the unconstrained leaves with the `free` mask (vi.py:508-514)."""
        self.t += 1
        by_hyper = {}
        for name in self.names():
            o = self.off[name]
            by_hyper.setdefault(lrs.hyper_of(name), []).append(
                (o, o + int(np.prod(self.shape[name])), float(lrs.lr_of(name))))
        sd = getattr(self, "_step_dev", None)
        kw = {"t_dev": sd} if sd is not None else {}         # captured step: Adam's t = the (already advanced) device counter
        hip = isinstance(self.be, HipBackend)
        if hip:                                              # the first optimiser launch also files the step's loss in the ring
            kw["loss"] = (self.G[self.n_params:self.n_params + 1], self.loss_ring)
        pp_hyper = {}
        if self.per_person:
            for nme, o in self.pp_off.items():
                pp_hyper.setdefault(lrs.hyper_of(nme), []).append(
                    (o, o + int(np.prod(self.pp_shape[nme])), float(lrs.lr_of(nme))))
        if len(by_hyper) == 1 and len(pp_hyper) == 1 and list(by_hyper) == list(pp_hyper):
            (betas, eps), segs = next(iter(by_hyper.items()))          # the usual case: one launch for both buffers
            self.be.adam2((self.P, self.G, self.M, self.V), self.free, self.n_params, _merge_segments(segs),
                          (self.PP, self.GP, self.MP, self.VP), self.pp_len, _merge_segments(pp_hyper[(betas, eps)]),
                          self.t, betas, eps, **kw)
            kw.pop("loss", None)
        else:
            for (betas, eps), segs in by_hyper.items():     # one launch per distinct (betas, eps): normally one
                self.be.adam(self.P, self.G, self.M, self.V, self.free, self.n_params, _merge_segments(segs), self.t, betas, eps, **kw)
                kw.pop("loss", None)
            for (betas, eps), segs in pp_hyper.items():
                self.be.adam(self.PP, self.GP, self.MP, self.VP, None, self.pp_len, _merge_segments(segs), self.t, betas, eps, **kw)
                kw.pop("loss", None)
        if not hip or "loss" in kw:
            # the CPU rehearsal backends of tests/ -- or a HIP step that launched no optimiser kernel at all (no trainable
            # segment): nothing filed the loss, so it is copied into its slot here
            self.loss_ring[self.t % LOSS_RING] = self.G[self.n_params]

    # -- the optimiser inside the step's last launch (one rank, D = 1 per-person guides) --------------
    fuse_tail = True

    def _fused_tail_args(self, lrs, rows, eps, S):
        """What vx_irt1d_grad_adam needs to run Adam in the launch that sums the step's slabs -- or None when this step has
        to keep loss_and_grads | all-reduce | optimiser apart (a process group, an amortized or multivariate guide, a
        subsample whose gradients are scattered afterwards, the score-function estimator, more than one (betas, eps))."""
        if not (self.fuse_tail and self.group is None and S == 1 and rows is None and isinstance(self.be, HipBackend)
                and isinstance(self, IrtEngine) and self.D == 1 and not self.amortized and self.per_person
                and self.estimator == "pathwise" and self.events is None and self.n_params == 4 * self.J):
            return None
        by_hyper, pp_hyper = {}, {}
        for name in self.names():
            o = self.off[name]
            by_hyper.setdefault(lrs.hyper_of(name), []).append((o, o + int(np.prod(self.shape[name])), float(lrs.lr_of(name))))
        for nme, o in self.pp_off.items():
            pp_hyper.setdefault(lrs.hyper_of(nme), []).append((o, o + int(np.prod(self.pp_shape[nme])), float(lrs.lr_of(nme))))
        if not (len(by_hyper) == 1 and len(pp_hyper) == 1 and list(by_hyper) == list(pp_hyper)):
            return None
        (betas, eps_), segs = next(iter(by_hyper.items()))
        return {"A": (self.P, self.M, self.V, self.free, self.n_params, _merge_segments(segs)),
                "B": (self.PP, self.GP, self.MP, self.VP, self.pp_len, _merge_segments(pp_hyper[(betas, eps_)])),
                "t": self.t + 1, "betas": betas, "eps": eps_, "loss_ring": self.loss_ring}

    def _take_fused_tail(self):
        """{'adam': ...} for the step kernel's call when step() armed the fused tail (once: the call consumes it)."""
        ft, self._fused_tail = getattr(self, "_fused_tail", None), None
        if ft is None:
            return {}
        self._fused_done = True
        return {"adam": ft}

    def _grads_and_optim(self, lrs, rows, b_global, eps=None):
        """One particle's loss_and_grads, the all-reduce and the optimiser -- or, on one rank with a D = 1 per-person
        guide, all of it inside loss_and_grads (the optimiser rides in the step's last launch)."""
        self._fused_tail, self._fused_done = self._fused_tail_args(lrs, rows, eps, 1), False
        try:
            self.loss_and_grads(rows, b_global, eps, 0)
        finally:
            self._fused_tail = None
        if self._fused_done:
            self.t += 1                                      # (what apply_optim does first; the launch filed the loss in the ring)
            return
        with self._phase("allreduce"):
            self.allreduce()
        with self._phase("optimizer"):
            self.apply_optim(lrs)

    def step_loss(self):
        """The loss of the step just taken as a 0-d device tensor (no host sync): its slot of the ring, not overwritten for
        the next LOSS_RING - 1 steps -- a list of returned losses holds distinct values."""
        return self.loss_ring[self.t % LOSS_RING]

    # -- the whole step as one HIP graph ---------------------------------------------------------
    # A D = 1 full-batch step is a handful of 10-100 us kernels, a subsampled amortized step (the reference's own usage:
    # subsample_size = 100, test.py:338) ~25 launches around kernels of 5-80 us, and a 125 k-person shard of the headline ~25
    # launches around 1.3 ms of large kernels: launched one by one the host (Python + ctypes, ~10 us a launch) and the gaps
    # between short kernels are a fixed cost per step.  Nothing in such a step changes from call to call except the step count
    # (Philox counter, Adam bias corrections) -- read from a device word that the step's last reduction advances -- and, for a
    # subsample, the row indices -- copied into a fixed device buffer before the replay.  So the step is captured once per
    # FORM (full batch; or a subsample of nb rows out of b_global) and replayed.  A new learning rate (scheduler milestone)
    # or a re-allocated workspace captures again.
    #
    # Sharded persons (a process group): by default the step is TWO replays around the eager all-reduce -- loss_and_grads |
    # all_reduce(G) | optimiser -- three host calls instead of ~30, and no collective inside a capture.  VX_GRAPH_COLLECTIVE=1
    # captures the RCCL all-reduce into ONE graph (RCCL collectives are capturable); that form has not run on any hardware this
    # build could reach, so it is opt-in, and a capture that fails with a collective inside it RAISES: the communicator may be
    # unusable after a half-recorded collective, so the process must not carry on (no eager retry).
    use_graph = True
    score_mfma = os.environ.get("VX_SCORE_MFMA", "1") != "0"      # test seam: 0 = the scalar score-operand kernel for every shape
    graph_max_persons = int(os.environ.get("VX_GRAPH_MAX_PERSONS", str(1 << 40)))

    def _graph_mode(self, rows, b_global, eps, S):
        """The captured form this call can replay -- ('full' | 'rows', nb, b_global) and, with S > 1 particles, S as a fourth
        entry -- or None for an eager step."""
        if not (self.use_graph and eps is None and self.events is None and isinstance(self.be, HipBackend)
                and getattr(self, "estimator", "pathwise") == "pathwise"):
            return None
        if S > 1:
            # Trace_ELBO(num_particles = S) (test.py:430: 20, test.py:607: 10): the S passes, their accumulation and the one
            # optimiser step as ONE graph.  Every particle draws its own subsample: S host index
            # tensors of one length, staged with one copy
            if rows is None:
                r0 = None
            elif (isinstance(rows, (list, tuple)) and len(rows) == S and all(torch.is_tensor(r) and not r.is_cuda for r in rows)
                  and len({int(r.numel()) for r in rows}) == 1):
                r0 = rows[0]
            else:
                return None
            base = self._graph_mode(r0, b_global, None, 1)
            return None if base is None else base + (S,)
        D, amort = getattr(self, "D", 0), getattr(self, "amortized", True)
        full = rows is None and (b_global is None or int(b_global) == self.N)
        if D == 1 and not amort and isinstance(self, IrtEngine):
            if full:
                return ("full", self.n_local, self.N)
            if rows is not None and not isinstance(rows, (list, tuple)) and self.n_local > 0:
                nb = int(rows.numel())                       # a subsample: the rows' gradients are scattered into zeroed dense ones
                return ("rows", nb, int(b_global) if b_global is not None else nb)
            return None
        if D > 1 and amort and isinstance(self, IrtEngine):
            if full:
                # a large shard runs short kernels BESIDE long ones on a second stream (the last chip round of the forward and
                # the hidden gradient, the fc1 gradient); the capture keeps those branches as graph edges.  Until round 5 the
                # replayed 1M step was the slower one (10.7 against 10.1 ms) and shards above 300 k ran eagerly; with the
                # round-6 kernels it is the faster one at every size measured (tools/step_events_cost.py: 8.97 against
                # 9.00 ms at 1M, 4.53 against 4.65 at 500 k, 1.46 against 1.63 at 125 k), so every shard replays.
                # VX_GRAPH_MAX_PERSONS remains as the switch back
                return ("full", self.n_local, self.N) if self.n_local <= self.graph_max_persons else None
            if rows is not None and not isinstance(rows, (list, tuple)) and self.n_local > 0:
                nb = int(rows.numel())
                return ("rows", nb, int(b_global) if b_global is not None else nb)
        if D == 1 and amort and isinstance(self, IrtEngine):
            # the amortized 1-D guide (VaeIRT x_feature == 1; the reference's Irt2PLMissing.test_ai draws 100 rows a step): the
            # step kernel's slab sum advances the counter, nothing behind it reads it but Adam
            if full:
                return ("full", self.n_local, self.N)
            if rows is not None and not isinstance(rows, (list, tuple)) and self.n_local > 0:
                nb = int(rows.numel())
                return ("rows", nb, int(b_global) if b_global is not None else nb)
        if D > 1 and not amort and isinstance(self, IrtEngine):
            # VIRT with x_feature > 1 (vi.py:706-723; the CFA demo of test.py:420-430 draws 100 rows a particle): the backward
            # kernel scatters the batch's rows into the zeroed per-person gradients itself
            if full:
                return ("full", self.n_local, self.N)
            if rows is not None and not isinstance(rows, (list, tuple)) and self.n_local > 0:
                nb = int(rows.numel())
                return ("rows", nb, int(b_global) if b_global is not None else nb)
        if isinstance(self, CdmSfEngine) and getattr(self, "baseline", "none") != "loo":
            # VCDM / VaeCDM (vi.py:733-816; test.py:522-549: 100 or 1000 rows a step): the Bernoulli draws' Philox step and Adam's
            # t from the device counter that the loss sum advances (k_cdm_sf reads vx_hodina_cfg.step_dev); the per-person
            # guide's gather / scatter of the batch rows are torch index kernels inside the capture
            if full:
                return ("full", self.n_local, self.N)
            if rows is not None and not isinstance(rows, (list, tuple)) and self.n_local > 0:
                nb = int(rows.numel())
                return ("rows", nb, int(b_global) if b_global is not None else nb)
        if isinstance(self, VaeCcdmEngine) and self.group is None:
            # VaeCCDM (vi.py:868-891; test.py:565,629: 100 or 50 rows a step) on one rank: no random numbers; the softmax over
            # the batch is three column reductions on the device (sharded, they are small all-reduces between kernels: eager)
            if full:
                return ("full", self.n_local, self.N)
            if rows is not None and not isinstance(rows, (list, tuple)) and self.n_local > 0:
                nb = int(rows.numel())
                return ("rows", nb, int(b_global) if b_global is not None else nb)
        if isinstance(self, CcdmEngine):
            # VCCDM (vi.py:819-865; test.py:560,585,624: 100-1500 rows a step): no guide, no random numbers -- the pattern
            # kernel, the loss sum (which advances the counter Adam reads) and the optimiser
            if full:
                return ("full", self.n_local, self.N)
            if rows is not None and not isinstance(rows, (list, tuple)) and self.n_local > 0:
                nb = int(rows.numel())
                return ("rows", nb, int(b_global) if b_global is not None else nb)
        if isinstance(self, HoDinaEngine):
            if full:
                return ("full", self.n_local, self.N)        # the enumerated HO-DINA step (either guide): three or six launches
            if amort and rows is not None and not isinstance(rows, (list, tuple)) and self.n_local > 0:
                nb = int(rows.numel())                       # VaeCHoDina on a subsample (test.py:598-607: 20 rows, 10 particles)
                return ("rows", nb, int(b_global) if b_global is not None else nb)
        return None

    def _graphable(self):
        """Whether the full-batch step of this engine replays a graph (tests, bench.py)."""
        return self._graph_mode(None, None, None, 1) is not None

    def _one_graph(self):
        """One graph for the whole step (no group, or the opt-in captured RCCL all-reduce) or two around an eager all-reduce."""
        if self.group is None:
            return True
        return (torch.distributed.get_backend(self.group) == "nccl" and os.environ.get("VX_GRAPH_COLLECTIVE", "0") == "1")

    def _graph_key(self, lrs):
        """Everything a captured step bakes into its kernel arguments: the (begin, end, lr) segments and (betas, eps) of
        every leaf as apply_optim computes them now (a new fit(), a scheduler milestone or a callable optim_args whose
        answer changed all change it), and the addresses of the buffers the kernels were handed (a workspace that
        _buf() has re-allocated since the capture would leave the graph writing into the old one)."""
        hyp = tuple((name, float(lrs.lr_of(name))) + lrs.hyper_of(name) for name in self.all_names())
        ptrs = tuple(sorted((k, t.data_ptr()) for k, t in self._ws.items()))
        return hyp, ptrs, self._one_graph()

    # the pinned host ring a captured subsampled step fetches its draw from: 16 slots, an event behind every 4th replay (a
    # record between two replays costs the GPU ~4 us of gap; tools/ring_sweep.sh: 8 / 1: 173.7 us a B = 100 step, 16 / 4: 168.9)
    rows_ring_slots = 16
    rows_ring_event_every = 4

    def _stage_rows(self, rows, buf, st=None, t=None):
        """The step's row indices into the fixed device buffer the captured kernels read.  Host indices (what the fit loop
        draws) go through a small ring of pinned buffers: one asynchronous copy, no staging allocation, no sync unless the
        GPU is a whole ring behind.  A form whose capture reads the ring itself (st["ring"]: vx_irt_cfg.rows_ring) only has
        the draw written into slot t % slots -- the replay fetches it."""
        if isinstance(rows, (list, tuple)):                  # S particles, S draws: [S][nb] in one buffer, one copy
            rows = torch.cat([r.reshape(-1) for r in rows])
        nb = buf.numel()
        if st is not None and st.get("ring") is not None:
            if rows.is_cuda:
                rows = rows.cpu()
            # the replay that last read this slot (step t - slots, if this form ran it) has finished once the OLDEST event
            # recorded at or behind that step has: events follow every `every`-th replay, so the host still runs
            # slots - every steps ahead of the GPU
            slot = (self.t if t is None else t) % self.rows_ring_slots
            lo, evs = st["ring_read"][slot], st["ring_ev"]
            if lo is not None:                               # the step that last read this slot
                for k in [k for k in evs if k < lo]:
                    del evs[k]
                if evs:
                    evs[min(evs)].synchronize()
                else:
                    torch.cuda.current_stream().synchronize()
                st["ring_read"][slot] = None
            st["ring"][slot, :nb].copy_(rows.reshape(-1))
            return
        if rows.is_cuda:
            buf.copy_(rows)
            return
        ring = getattr(self, "_pin_ring", None)
        if ring is None or ring[0][0].numel() < nb:
            ring = self._pin_ring = [[torch.empty(max(nb, 1024), dtype=torch.int64).pin_memory(), None] for _ in range(8)]
            self._pin_i = 0
        slot = ring[self._pin_i]
        self._pin_i = (self._pin_i + 1) % len(ring)
        if slot[1] is not None:
            slot[1].synchronize()                            # the copy that last read this slot has run
        else:
            slot[1] = torch.cuda.Event()
        slot[0][:nb].copy_(rows.reshape(-1))
        buf.copy_(slot[0][:nb], non_blocking=True)
        slot[1].record()

    def _step_graph(self, lrs, mode, rows):
        st = self._graphs[mode]
        self._graph = st                                     # (the form last replayed: what tests / bench.py look at)
        key = self._graph_key(lrs)
        if st["graph"] is None or st["key"] != key:
            if getattr(self, "_ctr", None) is None:
                self._ctr, self._ctr_t = torch.zeros(1, dtype=torch.int32, device=self.dev), None
            rows_buf = None
            ring = None
            S = mode[3] if len(mode) > 3 else 1
            if mode[0] == "rows":
                rows_buf = st.get("rows")
                if rows_buf is None:
                    rows_buf = torch.zeros(mode[1] * S, dtype=torch.int64, device=self.dev)
                self._stage_rows(rows, rows_buf)             # valid indices while the capture records
                ring = st.get("ring")
                if ring is None and S == 1 and not rows.is_cuda:
                    ring = torch.zeros(self.rows_ring_slots, max(int(mode[1]), 1), dtype=torch.int64).pin_memory()
            one = self._one_graph()
            torch.cuda.synchronize()
            gA, gB = torch.cuda.CUDAGraph(), None
            t0 = self.t
            self._step_dev = self._ctr
            self._capture_ring, self._capture_ring_used = ring, False
            try:
                # (thread-local capture mode: a process group's watchdog thread may query its events while this thread records)
                with torch.cuda.graph(gA, capture_error_mode="thread_local"):
                    # reads the counter as the Philox step, then advances it
                    if S > 1:
                        self._particles(S, lambda sidx: None if rows_buf is None else rows_buf[sidx * mode[1]:(sidx + 1) * mode[1]],
                                        mode[2] if mode[0] == "rows" else None, lambda sidx: None)
                        if one:
                            self.allreduce()
                            self.apply_optim(lrs)
                    elif one:
                        self._grads_and_optim(lrs, rows_buf, mode[2] if mode[0] == "rows" else None)   # (Adam's t: the counter)
                    else:
                        self.loss_and_grads(rows_buf, mode[2] if mode[0] == "rows" else None, None, 0)
                if not one:
                    gB = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gB, pool=gA.pool(), capture_error_mode="thread_local"):
                        self.apply_optim(lrs)
            except Exception as e:
                if one and self.group is not None:
                    raise RuntimeError("capturing the step with its RCCL all-reduce failed; the communicator may be unusable "
                                       "after a half-recorded collective, so this process must not carry on -- run again "
                                       "without VX_GRAPH_COLLECTIVE=1 (two replays around an eager all-reduce)") from e
                if self.group is not None and os.environ.get("VX_GRAPH_STRICT", "0") != "1":
                    # no collective was being recorded: the communicator is untouched and the same kernels can be launched one
                    # by one.  Sharded steps have never been captured next to a live RCCL communicator on the builder's boxes
                    # (one GPU each), so a failure here degrades -- loudly -- to the eager step instead of ending the job.
                    raise _GraphCaptureFailed(repr(e)) from e
                raise
            finally:
                self._step_dev = None
                self._capture_ring = None
                self.t = t0                                  # capture records, it does not run
            if not self._capture_ring_used:
                ring = None                                  # (this engine's forward does not read a ring: the copy stays)
            if st.get("ring_ev"):
                torch.cuda.current_stream().synchronize()
            st.update(graph=gA, tail=gB, key=key, rows=rows_buf, ring=ring, ring_ev={} if ring is not None else None,
                      ring_read=[None] * self.rows_ring_slots)
        if mode[0] == "rows":
            self._stage_rows(rows, st["rows"], st)
        if self._ctr_t != self.t:                            # (re)seed the device counter
            self._ctr.fill_(self.t)
        st["graph"].replay()
        if st["tail"] is not None:
            self.allreduce()
            st["tail"].replay()
        if st.get("ring") is not None:
            st["ring_read"][self.t % self.rows_ring_slots] = self.t
            if (self.t + 1) % self.rows_ring_event_every == 0:
                ev = st["ring_ev"][self.t] = torch.cuda.Event()
                ev.record()
        self.t += 1
        self._ctr_t = self.t
        return self.step_loss()

    # -- several steps of a fit loop from ONE replay ----------------------------------------------
    # Between two replays the GPU idles 6-9 us (the end of a graph releases to the system scope, the next one starts behind
    # it): a fifth of BASELINE config 2's 42 us step, 5 % of the reference's own B = 100 step.  Nothing a step needs from the
    # host is made by the step before it -- the Philox step and Adam's t come from the device counter, a subsample's rows from
    # the pinned ring (slot = counter % slots), the loss goes into the ring -- so graph_steps consecutive steps are captured
    # as one graph and a fit loop replays that (steps(): same kernels, same order, same bits as step() called in a loop).
    graph_steps = 4

    def _steps_form(self, lrs, rows_seq, b_global, scheduler):
        """The single-step form whose K-step graph can take the next K = graph_steps steps of a fit loop, or None."""
        K = self.graph_steps
        if K < 2 or len(rows_seq) < K or not hasattr(self, "_graphs") or not self._one_graph():
            return None
        mode = self._graph_mode(rows_seq[0], b_global, None, 1)
        if mode is None or any(self._graph_mode(r, b_global, None, 1) != mode for r in rows_seq[1:K]):
            return None
        st = self._graphs.get(mode)
        if st is None or st.get("graph") is None or st.get("key") != self._graph_key(lrs):
            return None                                      # (the form's first steps run one by one: eager, then its own capture)
        if mode[0] == "rows" and (st.get("ring") is None or any(r.is_cuda for r in rows_seq[:K])):
            return None
        if scheduler and any(lrs.epoch < m <= lrs.epoch + K - 1 for m in lrs.milestones):
            return None                                      # a milestone inside the K steps: their learning rates differ
        return mode

    def capture_steps(self, lrs, rows=None, b_global=None):
        """Captures the graph_steps-step graph of this form now (nothing runs), so that the first steps() call of a timed loop
        does not pay for it; True when the form has one afterwards (its single-step graph must exist: two steps taken)."""
        mode = self._steps_form(lrs, [rows] * self.graph_steps, b_global, False)
        if mode is None:
            return False
        return self._steps_capture(lrs, mode) is not None

    def _steps_capture(self, lrs, mode):
        st, K = self._graphs[mode], self.graph_steps
        mk = st.get("multi")
        key = (self._graph_key(lrs), K)
        if mk is None or mk["key"] != key:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            t0 = self.t
            self._step_dev = self._ctr
            self._capture_ring, self._capture_ring_used = st.get("ring"), False
            try:
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    for _ in range(K):                        # (with a group: the captured collective inside)
                        self._grads_and_optim(lrs, st.get("rows"), mode[2] if mode[0] == "rows" else None)
            except Exception as e:
                # as in _step_graph: with a collective inside the capture the communicator may be unusable -- no way on; a
                # capture WITHOUT one (a second private pool that does not fit, a driver refusing the K-fold graph) leaves
                # nothing half-done: the form keeps its single-step graph and the fit loop takes the steps one by one
                if self.group is not None and self._one_graph():
                    raise RuntimeError("capturing %d steps with their RCCL all-reduce failed; the communicator may be unusable "
                                       "after a half-recorded collective -- run again without VX_GRAPH_COLLECTIVE=1" % K) from e
                if os.environ.get("VX_GRAPH_STRICT", "0") == "1":
                    raise
                import warnings
                warnings.warn("capturing %d steps as one graph failed (%r): this engine replays its steps one at a time from "
                              "now on (same results)" % (K, e))
                self.graph_steps = 1
                return None
            finally:
                self._step_dev = None
                self._capture_ring = None
                self.t = t0
            mk = st["multi"] = {"graph": g, "key": key}
        return mk

    def _steps_graph(self, lrs, mode, rows_seq, scheduler):
        st, K = self._graphs[mode], self.graph_steps
        self._graphs[mode] = self._graphs.pop(mode)          # most recently used last
        self._graph = st
        mk = self._steps_capture(lrs, mode)
        if mk is None:                                       # the K-step capture failed and was withdrawn: single steps
            return None
        t0 = self.t
        if mode[0] == "rows":
            for j in range(K):
                self._stage_rows(rows_seq[j], st["rows"], st, t=t0 + j)
        if self._ctr_t != self.t:
            self._ctr.fill_(self.t)
        mk["graph"].replay()
        if st.get("ring") is not None:
            for j in range(K):
                st["ring_read"][(t0 + j) % self.rows_ring_slots] = t0 + j
            ev = st["ring_ev"][t0 + K - 1] = torch.cuda.Event()      # (behind all K steps)
            ev.record()
        self.t += K
        self._ctr_t = self.t
        if scheduler:
            for _ in range(K):
                lrs.scheduler_step()
        return [self.loss_ring[(t0 + j + 1) % LOSS_RING] for j in range(K)]

    def steps(self, lrs, rows_seq, b_global=None, scheduler=False):
        """len(rows_seq) consecutive steps of a fit loop -- what step(lrs, rows=r, b_global=b_global) for r in rows_seq does,
        each followed by lrs.scheduler_step() when `scheduler` (vi.py:639-640) -- with the same results bit for bit, replayed
        graph_steps at a time from one graph where the form allows.  rows_seq: one entry per step (None = the full batch, or
        host / device int64 LOCAL row indices).  Returns the losses, one 0-d device tensor per step (slots of the loss ring:
        the last LOSS_RING - 1 of them stay valid)."""
        out, i, n = [], 0, len(rows_seq)
        while i < n:
            mode = self._steps_form(lrs, rows_seq[i:], b_global, scheduler)
            if mode is not None:
                K = self.graph_steps
                got = self._steps_graph(lrs, mode, rows_seq[i:i + K], scheduler)
                if got is not None:
                    out += got
                    i += K
                    continue
            out.append(self.step(lrs, rows=rows_seq[i], b_global=b_global))
            if scheduler:
                lrs.scheduler_step()
            i += 1
        return out

    graph_max_row_forms = 8

    def _evict_graph_forms(self):
        """Every distinct (nb, b_global) subsample form holds a captured graph, its private pool and a rows buffer.  A fit loop
        has one form per rank; a caller whose local row count varies from step to step would otherwise keep one per count for
        ever: beyond graph_max_row_forms the least recently used 'rows' forms are dropped, graph and buffer."""
        forms = [m for m in self._graphs if m[0] == "rows"]
        for m in forms[:max(0, len(forms) - self.graph_max_row_forms)]:
            st = self._graphs.pop(m)
            if self._graph is st:
                self._graph = None
            # a replay of this form may still be queued: it reads the pinned rows ring through a raw device pointer (no event
            # of torch's host allocator guards that use) and runs out of the graph's private pool -- wait for the newest event
            # recorded behind its replays, or for the stream when there is none, before any of it can be handed out again
            evs = st.get("ring_ev")
            if evs:
                evs[max(evs)].synchronize()
            elif st.get("graph") is not None and self.dev.type == "cuda":
                torch.cuda.current_stream().synchronize()
            st.clear()                                       # releases the CUDAGraph objects (and their pool) and the rows buffer

    def _particles(self, S, rows_of, b_global, eps_of):
        """S passes of loss_and_grads (particle sidx = Philox stream sidx) and their mean in G / GP: surrogate / num_particles
        (SURVEY.md App. B.2).  While a capture records, every pass's last launch advances the device step counter: it is set
        back behind all but the last, so that the S particles share the step and the optimiser sees step + 1."""
        accG = torch.zeros_like(self.G)
        accP = torch.zeros_like(self.GP) if self.per_person else None
        sd = getattr(self, "_step_dev", None)
        for sidx in range(S):
            self.loss_and_grads(rows_of(sidx), b_global, eps_of(sidx), sidx)
            if sd is not None and sidx < S - 1:
                sd.sub_(1)
            accG.add_(self.G, alpha=1.0 / S)
            if accP is not None:
                accP.add_(self.GP, alpha=1.0 / S)
        self.G.copy_(accG)
        if accP is not None:
            self.GP.copy_(accP)

    def _rows_on_device(self, rows):
        """Host row indices (the fit loop's draw) for an eager step."""
        if rows is None or isinstance(rows, (list, tuple)):
            return [self._rows_on_device(r) for r in rows] if isinstance(rows, (list, tuple)) else None
        return rows if rows.device == self.dev else rows.to(self.dev)

    def step(self, lrs, rows=None, b_global=None, eps=None, num_particles=1):
        """loss_and_grads + optimiser, the body of SVI.step (vi.py:505-516).  `rows` (and `eps`) may be
        lists with one entry per particle: every particle draws its own subsample.
        rows: int64 LOCAL row indices, on the device or on the host (a captured step copies host indices itself).
        Returns the loss as a 0-d device tensor (no host sync, no launch of its own: the optimiser launch files it in a ring of
        LOSS_RING slots, see step_loss())."""
        S = int(num_particles)
        mode = self._graph_mode(rows, b_global, eps, S)
        if mode is not None:
            if not hasattr(self, "_graphs"):
                self._graphs = {}
            if mode in self._graphs:
                self._graphs[mode] = self._graphs.pop(mode)  # most recently used last (dicts keep insertion order)
                try:
                    return self._step_graph(lrs, mode, rows)
                except _GraphCaptureFailed as e:
                    import warnings
                    warnings.warn("HIP graph capture of the sharded step failed (%s): this rank steps kernel by kernel from "
                                  "here on (VX_GRAPH_STRICT=1 makes this an error)" % e)
                    self.use_graph = False                   # (same kernels, same results: launched one by one)
                    self.graph_fallback = str(e)
                    del self._graphs[mode]
                    mode = None
            if mode is not None:
                self._graphs[mode] = {"graph": None}         # the first step of a form runs eagerly: workspaces and lists get built
                self._graph = self._graphs[mode]
                self._evict_graph_forms()
        rows = self._rows_on_device(rows)
        if S == 1:
            self._grads_and_optim(lrs, rows[0] if isinstance(rows, (list, tuple)) else rows, b_global,
                                  eps[0] if isinstance(eps, (list, tuple)) else eps)
            return self.step_loss()
        else:
            self._particles(S, lambda sidx: rows[sidx] if isinstance(rows, (list, tuple)) else rows, b_global,
                            lambda sidx: eps[sidx] if isinstance(eps, (list, tuple)) else eps)
        with self._phase("allreduce"):
            self.allreduce()
        with self._phase("optimizer"):
            self.apply_optim(lrs)
        return self.step_loss()


class _GraphCaptureFailed(RuntimeError):
    """Capture of a sharded step (no collective inside) failed; _EngineBase.step() falls back to the eager step."""


class IrtEngine(_EngineBase):
    """IRT ELBO-gradient step (VIRT / VaeIRT of the reference, vi.py:536-723) on one rank.

    Flat parameter buffer: [a: D*J | b: J | c_un: J | d_un: J | encoder (nn.Linear order, amortized only)]."""

    def __init__(self, y_u8, model="irt_2pl", D=1, Dc=1.0, n_global=None, gid0=0, amortized=False, H=64,
                 share_cov=False, a_free=None, a0=None, b0=None, encoder_init=None, seed=1234, group=None,
                 backend=None, observed_lists=True, estimator="pathwise", baseline="none", baseline_beta=0.9):
        """group: the torch.distributed process group whose ranks SHARE this problem (each holds a contiguous shard of
        persons: y_u8 = rows gid0 .. gid0 + n_local of the n_global); None = this process owns the whole problem.
        observed_lists: D = 1, full batch, >= 50 % missing -> step on compacted lists of observed cells.
        estimator: 'pathwise' = what pyro's Trace_ELBO does for the reference's Normal guides (vi.py:684,693,705,715,723);
        'score' (BASELINE.json north_star; every guide of this engine) = the score-function (REINFORCE) gradient of the
        guide, (log_r_i - baseline_i) d log q / d phi, with baseline 'none', 'avg' (per-person decaying average of log_r, rate
        baseline_beta) or 'loo' (leave-one-out mean over the particles of a step, num_particles >= 2; the particles then share
        the step's subsample).  The item gradients are the pathwise ones in every mode."""
        if estimator not in ("pathwise", "score"):
            raise ValueError("estimator must be 'pathwise' or 'score'")
        if baseline not in ("none", "avg", "loo"):
            raise ValueError("baseline must be 'none', 'avg' or 'loo'")
        self.estimator, self.baseline, self.baseline_beta = estimator, baseline, float(baseline_beta)
        self.be = backend if backend is not None else HipBackend()
        self.observed_lists = bool(observed_lists) and estimator == "pathwise"
        self.y = y_u8.contiguous()
        assert self.y.dtype == torch.uint8 and self.y.dim() == 2
        self.dev = self.y.device
        self.n_local, self.J_items = self.y.shape
        self.y_lik = None
        self.H_model = int(H) if amortized else 0                  # the encoder's hidden units; self.H: what the kernels see
        if amortized and 0 < int(H) < 64 and isinstance(self.be, HipBackend) and self.pad_hidden:
            # PHANTOM HIDDEN UNITS up to 64, the width the MFMA kernels are built for (hidden_dim 32: 23.6 ms a step where 64
            # takes 2.4, tools/hidden_cliffs.py).  A phantom unit has a zero row of fc1 (weight and bias) and zero columns in
            # both heads.  Its activation is softplus(0) = log 2, not zero (vi.py:430,447), so the heads' columns WOULD take
            # gradients: they are zeroed behind the backward call (_zero_phantom_head_grads) and the columns stay zero; d ELBO /
            # d h_u is a sum over head rows times those zero columns, so fc1's row takes no gradient at all.
            H = 64
        if (amortized and int(H) == 64 and self.J_items % 4 != 0 and self.J_items > 0
                and isinstance(self.be, HipBackend) and self.pad_items):
            # PHANTOM ITEMS up to a multiple of 4: the MFMA kernels of the amortized guides read response rows in
            # 16-byte pieces and refuse other item counts (J = 499: 18.0 ms a step where J = 500 takes 2.1, tools/shape_cliffs.py).
            # A phantom item is ABSENT for the likelihood (byte 254, "outside the problem": no log-probability, no gradient)
            # and ZERO for the encoder (its input and the fc1 gradient's activation), so there are two copies of the responses:
            # self.y (phantom byte 0) for the guide's forward / backward calls, self.y_lik (254) for the likelihood call and
            # the item-major copy.  Its parameters -- a column of a, entries of b / c / d, a column of fc1.weight -- take
            # zero gradients for ever (Adam leaves them where they start) and are not visible through unconstrained() / param().
            # (The observed-cell lists of the 1-D step count a phantom among a person's J - n_observed unobserved items, each
            # of which carries the reference's constant log Bern(0 | clamp(0)) = -1.19e-7: 1e-9 of a person's ELBO.)
            pad = (-self.J_items) % 4
            self.y_lik = torch.cat([self.y, torch.full((self.n_local, pad), 254, dtype=torch.uint8, device=self.dev)], 1).contiguous()
            self.y = torch.cat([self.y, torch.zeros((self.n_local, pad), dtype=torch.uint8, device=self.dev)], 1).contiguous()
        self.J = int(self.y.shape[1])                              # items the kernels see (J_items + phantoms)
        self.N = int(n_global) if n_global is not None else self.n_local
        self.gid0 = int(gid0)
        self.model, self.D, self.Dc = model, int(D), float(Dc)
        self.D_model = self.D                                      # the model's latent dimensions; self.D: what the kernels see
        if (amortized and self.D > 1 and self.D % 4 != 0 and self.D + 3 <= 124 and int(H) == 64 and isinstance(self.be, HipBackend)
                and self.pad_dims):
            # PHANTOM DIMENSIONS up to a multiple of 4 (the MFMA kernels' other shape condition: D = 99: 31 ms a step where
            # D = 100 takes 2.1).  A phantom dimension k has a_k = 0 for ever (free mask 0: it never reaches a logit) and head
            # rows (fc21 row k, the fc22 rows (k, .), their biases) that start at ZERO and are kept there by zeroing their
            # gradients behind the backward call: mu_k = 0, L_kk = e^0 = 1, L_kl = 0, so x_k = eps_k and the phantom's
            # terms of the ELBO cancel (-x_k^2 / 2 from the prior, +eps_k^2 / 2 + M_kk from the entropy), and the shared
            # hidden layer gets nothing from it (its rows of the head weights are what the hidden gradient multiplies by).
            # The draws of the real dimensions are the same (Philox blocks of four dimensions), tril order puts the real
            # fc22 rows first.  Not visible through unconstrained() / param().
            self.D = (self.D + 3) // 4 * 4
        self.amortized, self.H, self.share_cov = bool(amortized), int(H) if amortized else 0, bool(share_cov)
        self.seed, self.group = int(seed), group
        J, Dd = self.J, self.D
        self.off = {"a": 0, "b": Dd * J, "c": Dd * J + J, "d": Dd * J + 2 * J}
        self.shape = {"a": (Dd, J), "b": (1, J), "c": (1, J), "d": (1, J)}
        self.n_item = Dd * J + 3 * J
        o = (self.n_item + 63) // 64 * 64                         # encoder tensors start on 256-byte boundaries
        if self.amortized:
            T = Dd * (Dd + 1) // 2 if Dd > 1 else 1
            self.enc_shapes = {"fc1.weight": (self.H, J), "fc1.bias": (self.H,), "fc21.weight": (Dd, self.H),
                               "fc21.bias": (Dd,), "fc22.weight": (T, self.H), "fc22.bias": (T,)}
            self.enc_off0 = o
            for k in ENC_KEYS:
                self.off["encoder$$$" + k] = o
                self.shape["encoder$$$" + k] = self.enc_shapes[k]
                o += int(np.prod(self.enc_shapes[k]))
            self.n_enc = o - self.enc_off0
        else:
            o = self.n_item
            if Dd > 1:                                    # VIRT.guide for x_feature > 1 (vi.py:706-723)
                if self.share_cov:                        # one (D, D) Cholesky factor, replicated like an item parameter
                    o = (self.n_item + 63) // 64 * 64
                    self.off["x_scale"] = o
                    self.shape["x_scale"] = (Dd, Dd)
                    o += Dd * Dd
                    self.pp_shape = {"x_local": (self.n_local, Dd)}
                else:
                    self.pp_shape = {"x_local": (self.n_local, Dd), "x_scale": (self.n_local, Dd, Dd)}
        self._alloc(o, self.n_local, per_person=not self.amortized)
        # reference initial values (vi.py:567-587), over the problem's own items; phantom items (above) start at zero
        Ji, Dm = self.J_items, self.D_model

        def pad_j(t):                                              # [..., J_items] -> [..., J]
            return t if J == Ji else torch.nn.functional.pad(t, (0, J - Ji))

        def pad_rows(t, n):                                        # [r, ...] -> [n, ...], zero rows behind
            return t if t.shape[0] == n else torch.cat([t, torch.zeros((n - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype)])
        a_init = torch.ones(Dm, Ji) if a0 is None else torch.as_tensor(a0, dtype=torch.float32).reshape(Dm, Ji).clone()
        if a_free is None and Dm > 1:
            af = torch.ones(Dm, Ji)
            for i in range(Dm):
                af[i, Ji - i:] = 0                                 # vi.py:570-572
            a_init = a_init * af                                   # also a user-supplied a0 is zeroed there (vi.py:571)
            a_free = af
        if Dd != Dm and a_free is None:
            a_free = torch.ones(Dm, Ji)
        if model != "irt_1pl":
            self.view("a").copy_(pad_rows(pad_j(a_init), Dd).reshape(-1))
            if a_free is not None:
                self.free[self.off["a"]:self.off["a"] + Dd * J] = pad_rows(pad_j(
                    torch.as_tensor(a_free, dtype=torch.float32).reshape(Dm, Ji)), Dd).reshape(-1)
        if b0 is not None:
            self.view("b").copy_(pad_j(torch.as_tensor(b0, dtype=torch.float32).reshape(1, Ji)).reshape(-1))
        if model in ("irt_3pl", "irt_4pl"):
            self.view("c").fill_(float(np.float32(_logit(np.float32(0.1)))))
        if model == "irt_4pl":
            self.view("d").fill_(float(np.float32(_logit(np.float32(1.0) - np.float32(0.1)))))
        if self.amortized:
            Hm = self.H_model
            if encoder_init is None:
                encoder_init = default_encoder_init(Ji, Dm, Hm, seed)
            for k in ENC_KEYS:
                w = torch.as_tensor(encoder_init[k], dtype=torch.float32)
                if k == "fc1.weight":
                    w = pad_rows(pad_j(w.reshape(Hm, Ji)), self.H)       # phantom items: zero columns; phantom units: zero rows
                elif k == "fc1.bias":
                    w = pad_rows(w.reshape(-1), self.H)
                else:                                              # the heads: zero columns for the phantom units, zero rows
                    if k.endswith("weight"):                       # for the phantom dimensions
                        w = torch.nn.functional.pad(w.reshape(-1, Hm), (0, self.H - Hm))
                    w = pad_rows(w.reshape((-1, self.H) if k.endswith("weight") else (-1,)), self.enc_shapes[k][0])
                self.view("encoder$$$" + k).copy_(w.reshape(-1))
        self.base = (torch.zeros(max(self.n_local, 1), dtype=torch.float32, device=self.dev)
                     if (estimator == "score" and baseline == "avg") else None)

    pad_items = os.environ.get("VX_PAD_ITEMS", "1") != "0"          # test seam: 0 = the kernels see the problem's own item count

    pad_batch = os.environ.get("VX_PAD_BATCH", "1") != "0"          # test seam: 0 = a subsample is launched over its own size

    def _pad_batch(self, rows, b_global, eps):
        """A SUBSAMPLE whose size is no multiple of 4 (B = 50: 315 us a step where B = 100 takes 144 -- the person-major
        generation of the backward kernels) is drawn 1-3 rows longer: the extra rows point at ONE phantom person behind the
        shard's own (index n_local of an extended copy of the responses, every response missing), and loss_and_grads takes
        them out behind the likelihood by their index (columns of gxT / gdT, entries of ll / ent times 0), as _pad_persons
        does for a full batch.  The plate scale stays the caller's: b_global is made explicit.  Amortized multivariate guide,
        pathwise estimator, HIP backend; rows: one tensor or one per particle."""
        if (rows is None or eps is not None or not self.pad_batch or not (self.amortized and self.D > 1 and self.H == 64)
                or not isinstance(self.be, HipBackend) or self.estimator != "pathwise"):
            return rows, b_global
        many = isinstance(rows, (list, tuple))
        rs = list(rows) if many else [rows]
        if not rs or not all(torch.is_tensor(r) for r in rs) or len({int(r.numel()) for r in rs}) != 1:
            return rows, b_global
        n = int(rs[0].numel())
        pad = (-n) % 4
        if pad == 0 or n == 0:
            return rows, b_global
        cfg = self.be.cfg(self.model, self.D, self.J, self.H, self.Dc, 1.0, self.seed, 0, 0)
        if self.be.mvn_enc_bwd_layout(cfg, n + pad) != 1 or self.be.mvn_enc_bwd_gd_offset(cfg, n + pad) < 0:
            return rows, b_global
        if getattr(self, "_y_ext", None) is None:
            miss = torch.full((1, self.J), 255, dtype=torch.uint8, device=self.dev)
            self._y_ext = torch.cat([self.y, miss]).contiguous()
            self._y_ext_lik = self._y_ext if self.y_lik is None else torch.cat([self.y_lik, miss]).contiguous()
        rs = [torch.cat([r.reshape(-1), torch.full((pad,), self.n_local, dtype=r.dtype, device=r.device)]) for r in rs]
        return (rs if many else rs[0]), (n if b_global is None else b_global)

    def step(self, lrs, rows=None, b_global=None, eps=None, num_particles=1):
        rows, b_global = self._pad_batch(rows, b_global, eps)
        return super().step(lrs, rows=rows, b_global=b_global, eps=eps, num_particles=num_particles)

    def steps(self, lrs, rows_seq, b_global=None, scheduler=False):
        if rows_seq and all(torch.is_tensor(r) for r in rows_seq) and len({int(r.numel()) for r in rows_seq}) == 1:
            padded, b_global = self._pad_batch(list(rows_seq), b_global, None)   # (a list: the same rule as for particles)
            rows_seq = padded
        return super().steps(lrs, rows_seq, b_global=b_global, scheduler=scheduler)

    pad_hidden = os.environ.get("VX_PAD_HIDDEN", "1") != "0"        # test seam: 0 = the kernels see the encoder's own width
    pad_dims = os.environ.get("VX_PAD_DIMS", "1") != "0"            # test seam: 0 = the kernels see the model's own dimensions

    def unconstrained(self, name, buf=None):
        u = super().unconstrained(name, buf)
        if self.J != self.J_items and name in ("a", "b", "c", "d", "encoder$$$fc1.weight"):
            u = u[..., :self.J_items]                              # (phantom items are nobody's business)
        if self.D != self.D_model:                                 # (nor are phantom dimensions)
            if name == "a" or name.startswith("encoder$$$fc21"):
                u = u[:self.D_model]
            elif name.startswith("encoder$$$fc22"):
                u = u[:self.D_model * (self.D_model + 1) // 2]
        if self.amortized and self.H != self.H_model:              # (nor phantom hidden units)
            if name.startswith("encoder$$$fc1."):
                u = u[:self.H_model]
            elif name in ("encoder$$$fc21.weight", "encoder$$$fc22.weight"):
                u = u[:, :self.H_model]
        return u

    def _zero_phantom_head_grads(self):
        """Phantom dimensions: the gradients of their head rows go back to zero; phantom hidden units: the gradients of their
        head columns (IrtEngine.__init__)."""
        Dm, Dp, H, Hm = self.D_model, self.D, self.H, self.H_model
        if Dp != Dm:
            Tm = Dm * (Dm + 1) // 2
            for k, lo in (("fc21.weight", Dm * H), ("fc21.bias", Dm), ("fc22.weight", Tm * H), ("fc22.bias", Tm)):
                self.view("encoder$$$" + k, self.G)[lo:].zero_()
        if self.amortized and H != Hm:
            for k in ("fc21.weight", "fc22.weight"):
                self.view("encoder$$$" + k, self.G).view(-1, H)[:, Hm:].zero_()

    # -- parameter access (constrained values as pyro.param(name) returns them) -----------------
    def names(self):
        out = ["b"]
        if self.model != "irt_1pl":
            out.append("a")
        if self.model in ("irt_3pl", "irt_4pl"):
            out.append("c")
        if self.model == "irt_4pl":
            out.append("d")
        if self.amortized:
            out += ["encoder$$$" + k for k in ENC_KEYS]
        elif self.D > 1 and self.share_cov:
            out.append("x_scale")
        return out

    def all_names(self):
        return self.names() + (list(self.pp_off) if self.per_person else [])

    def param(self, name):
        u = self.unconstrained(name)
        if name in ("c", "d"):
            return torch.sigmoid(u)
        if name == "x_scale":
            if self.D == 1:
                return torch.exp(u)
            diag = torch.diagonal(u, dim1=-2, dim2=-1)                # lower_cholesky transform (vi.py:712,720)
            return torch.tril(u, -1) + torch.diag_embed(torch.exp(diag))
        return u.clone()

    # -- one ELBO-gradient step ------------------------------------------------------------------
    def _score_baseline(self, baseline_buf, guide_grads):
        """(control variate, decay rate or -1, indexed-by-row flag) of the score-function estimator for this pass."""
        if baseline_buf is not None:
            return baseline_buf, -1.0, 0
        if self.baseline == "avg":
            return self.base, (self.baseline_beta if guide_grads else -1.0), 1
        return None, -1.0, 0

    def loss_and_grads(self, rows=None, b_global=None, eps=None, stream_id=0, baseline_buf=None, guide_grads=True):
        """Fills self.G (flat grads + loss slot) and per-person grads for ONE particle.
        rows: int64 device tensor of LOCAL row indices (None = all local rows, i.e. full batch).
        baseline_buf / guide_grads: the 'loo' mode of the score-function estimator (step())."""
        be = self.be
        nb = self.n_local if rows is None else int(rows.numel())
        Bg = int(b_global) if b_global is not None else (self.N if rows is None else nb)
        scale = float(self.N) / float(Bg)
        sd = getattr(self, "_step_dev", None)               # captured step: the step count lives in device memory
        sdc = {"step_dev": sd} if sd is not None else {}
        cfg = be.cfg(self.model, self.D, self.J, self.H, self.Dc, scale, self.seed, self.t, stream_id, **sdc)
        c_un = self.view("c") if self.model in ("irt_3pl", "irt_4pl") else None
        d_un = self.view("d") if self.model == "irt_4pl" else None
        a = self.view("a") if self.model != "irt_1pl" else None
        gitem = self.G[:self.n_item]
        lossslot = self.G[self.n_params:self.n_params + 1]
        if self.D > 1 and not self.amortized:
            D, n = self.D, self.n_local
            fw = {"x": self._buf("x", nb * D), "eps": self._buf("eps", nb * D), "ent": self._buf("ent", nb)}
            gx, ll = self._buf("gx", nb * D), self._buf("ll", nb)
            lik_ws = self._buf("lik_ws", be.lik_workspace(cfg, nb))
            loc = self.PP[:n * D]
            if self.share_cov:
                Mq, gM = self.view("x_scale"), self.view("x_scale", self.G)
            else:
                Mq, gM = self.PP[n * D:], self.GP[n * D:]
            with self._phase("guide_forward"):
                be.mvn_bbvi_forward(cfg, nb, rows, self.gid0, loc, Mq, self.share_cov, eps, fw["x"], fw["eps"], fw["ent"])
            with self._phase("likelihood"):
                be.lik_grad(cfg, self.y, rows, nb, fw["x"], a, self.view("b"), c_un, d_un, gx, ll, gitem, lik_ws)
            cfg_b = cfg
            if self.estimator == "score":
                # the guide's score-function gradient in place of the pathwise one: gx <- w_i L_i^-T eps_i, and the diagonal
                # term -w_i in place of the entropy's +scale (k_mvn_score.hip); item gradients and loss stand
                with self._phase("mvn_score"):
                    w_sf, log_r = self._buf("sf_w", nb), self._buf("sf_lr%d" % stream_id if not guide_grads else "sf_lr", nb)
                    base, beta, by_row = self._score_baseline(baseline_buf, guide_grads)
                    be.mvn_score_operands(cfg, nb, rows, 2 if self.share_cov else 1, None, None, None, Mq, fw["eps"], ll,
                                          fw["ent"], base, beta, by_row, log_r, w_sf, gx, None, None)
                    self.last_log_r = log_r
                cfg_b = be.cfg(self.model, self.D, self.J, self.H, self.Dc, 0.0, self.seed, self.t, stream_id)
            with self._phase("guide_backward"):
                self.GP.zero_()                                       # dense per-person grads: zero off the batch
                if self.share_cov:
                    gM.zero_()
                be.mvn_bbvi_backward(cfg_b, nb, rows, Mq, self.share_cov, gx, fw["eps"], self.GP[:n * D], gM,
                                     self._buf("bbvi_ws", be.mvn_bbvi_bwd_workspace(cfg, nb, self.share_cov)))
                if self.estimator == "score":
                    be.mvn_score_diag(cfg, nb, rows, w_sf, self.share_cov, gM)
            be.sum2_into(ll, fw["ent"], nb, -scale, lossslot, self.sum_ws, **sdc)   # loss = -scale * sum_i (ll_i + ent_i)
            self.last = {"fw": fw, "gx": gx, "ll": ll, "nb": nb}
        elif self.D > 1:
            D, H = self.D, self.H
            if eps is not None and D != self.D_model:      # a caller's draws are the model's: the phantom dimensions draw zero
                e_p = torch.zeros(nb, D, dtype=torch.float32, device=self.dev)
                e_p[:, :self.D_model] = eps.reshape(nb, self.D_model)
                eps = e_p
            n_valid, y_k, y_l = nb, self.y, (self.y_lik if self.y_lik is not None else self.y)
            keep = None
            if rows is not None and getattr(self, "_y_ext", None) is not None:
                # rows of a padded subsample may point at the phantom person behind the shard's own (_pad_batch)
                y_k, y_l = self._y_ext, self._y_ext_lik
                keep = (rows != self.n_local).to(torch.float32)
            if rows is None and eps is None:
                self._n_pad = self._pad_persons(be, cfg, rows)
                if self._n_pad != nb:                      # phantom persons up to a multiple of 8 (_pad_persons)
                    nb, y_k, y_l = self._n_pad, self._padded_y(), self._padded_y(lik=True)
            fw = {"h": self._buf("h", nb * H), "x": self._buf("x", nb * D), "eps": self._buf("eps", nb * D),
                  "ldT": self._buf("ldT", nb * D), "ent": self._buf("ent", nb)}
            fw["packws"] = self._buf("packws", be.mvn_pack_floats(cfg))
            # dimension-major copies (person-contiguous rows) for the DMA-staged weight-gradient kernel
            fw["hT"], fw["epsT"] = self._buf("hT", nb * H), self._buf("epsT", nb * D)
            gxT = self._buf("gxT", nb * D)
            yT = self._item_major_y(rows)
            if yT is not None:
                fw["yT"] = yT
                n_img = be.lik_ximg_bytes(cfg, nb)        # x once more as the likelihood kernel's operand image
                if n_img > 0:
                    fw["ximg"] = self._buf("ximg", (n_img + 3) // 4)
            gx, ll = self._buf("gx", nb * D), self._buf("ll", nb)
            if be.mvn_enc_bwd_layout(cfg, nb) == 1:
                gx = None                                  # the backward kernels read gxT only
            enc = self._enc()
            lik_ws = self._buf("lik_ws", be.lik_workspace(cfg, nb))
            encb_ws = self._buf("encb_ws", be.mvn_enc_bwd_workspace(cfg, nb))
            hs_off = be.mvn_enc_bwd_hs_offset(cfg, nb)     # the weight-gradient kernel's bf16 terms of hT, written by the forward
            if hs_off >= 0:
                fw["hs"] = encb_ws[hs_off:hs_off + nb * 64]
            with self._phase("guide_forward"):
                cfg_f = cfg
                ring = getattr(self, "_capture_ring", None)
                if ring is not None and rows is not None and sd is not None and isinstance(be, HipBackend):
                    # a captured subsampled step: the forward call's first launch fetches the step's draw from the pinned
                    # host ring (vx_irt_cfg.rows_ring) -- no copy in front of every replay
                    cfg_f = be.cfg(self.model, self.D, self.J, self.H, self.Dc, scale, self.seed, self.t, stream_id,
                                   step_dev=sd, rows_ring=ring)
                    self._capture_ring_used = True
                be.mvn_enc_forward(cfg_f, y_k, rows, nb, self.gid0, enc, eps, fw)
            gd_off = be.mvn_enc_bwd_gd_offset(cfg, nb)     # the backward's DIAG-row operand, made in the likelihood's last pass
            gdT = encb_ws[gd_off:gd_off + nb * D] if gd_off >= 0 else None
            # the step's largest |gx|, |gd|, |eps| (the head weight gradient's power of two), collected by the likelihood's last
            # pass into words of packws that the forward call cleared -- unless the score-function mode replaces gxT / gdT below
            opmax = None
            if gdT is not None and self.estimator == "pathwise" and isinstance(be, HipBackend):
                om = be.mvn_pack_opmax_offset(cfg, nb)
                if om >= 0:
                    opmax = fw["packws"][om:om + 3]
            with self._phase("likelihood"):
                be.lik_grad(cfg, y_l, rows, nb, fw["x"], a, self.view("b"), c_un, d_un, gx, ll, gitem, lik_ws,
                            gxT=gxT, yT=yT, ximg=fw.get("ximg"), epsT=fw["epsT"] if gdT is not None else None,
                            ldT=fw["ldT"] if gdT is not None else None, gdT=gdT, **({"opmax": opmax} if opmax is not None else {}))
                if nb != n_valid:
                    # the phantom persons leave the step here: nothing behind this line sees them but as zeros
                    gxT[:nb * D].view(D, nb)[:, n_valid:].zero_()
                    gdT[:nb * D].view(D, nb)[:, n_valid:].zero_()
                    ll[n_valid:nb].zero_()
                    fw["ent"][n_valid:nb].zero_()
                if keep is not None and gdT is not None and gx is None:
                    gxT[:nb * D].view(D, nb).mul_(keep)    # (a padded subsample's phantom rows, wherever they stand)
                    gdT[:nb * D].view(D, nb).mul_(keep)
                    ll[:nb].mul_(keep)
                    fw["ent"][:nb].mul_(keep)
            if self.estimator == "score":
                # gxT, gdT <- the score-function operands (k_mvn_score.hip); item gradients and loss stand
                if gx is not None or gdT is None:
                    raise NotImplementedError("the score-function estimator of the amortized multivariate guide runs on the "
                                              "dimension-major backward kernels: hidden_dim 64, x_feature % 4 == 0 (<= 124), "
                                              "item_size % 4 == 0, batch % 4 == 0")
                with self._phase("mvn_score"):
                    log_r = self._buf("sf_lr%d" % stream_id if not guide_grads else "sf_lr", nb)
                    base, beta, by_row = self._score_baseline(baseline_buf, guide_grads)
                    n_sh = be.mvn_score_heads_workspace(cfg) if (isinstance(be, HipBackend) and self.score_mfma) else -1
                    if n_sh > 0:
                        # u = L^-T eps on the fp16 MFMA, the head rows in column order (k_mvn_score_b.hip)
                        be.mvn_score_heads(cfg, nb, rows, fw["h"], enc["fc22.weight"], enc["fc22.bias"], fw["packws"], fw["eps"],
                                           ll, fw["ent"], base, beta, by_row, log_r, gxT, gdT,
                                           self._buf("score_img", n_sh))
                    else:
                        be.mvn_score_operands(cfg, nb, rows, 0, fw["h"], enc["fc22.weight"], enc["fc22.bias"], None, fw["eps"], ll,
                                              fw["ent"], base, beta, by_row, log_r, None, None, gxT, gdT)
                    self.last_log_r = log_r
            with self._phase("guide_backward"):
                # loss = -scale * sum_i (ll_i + ent_i), from the backward call's last launch; a captured step's counter
                # advances there, behind every kernel that read it
                fused_loss = isinstance(be, HipBackend)
                be.mvn_enc_backward(cfg, y_k, rows, nb, enc, fw, gx,
                                    self.G[self.enc_off0:self.enc_off0 + self.n_enc], encb_ws, gxT=gxT,
                                    gd_ready=gdT is not None, **({"opmax_ready": True} if opmax is not None else {}),
                                    **({"loss": (ll, fw["ent"], -scale, lossslot, self.sum_ws)} if fused_loss else {}))
            if not fused_loss:
                be.sum2_into(ll, fw["ent"], nb, -scale, lossslot, self.sum_ws, **sdc)
            self._zero_phantom_head_grads()
            self.last = {"fw": fw, "gx": gx, "gxT": gxT, "ll": ll, "nb": nb, "n_valid": n_valid}
        else:
            # D = 1: the flat item layout [a: J | b: J | c: J | d: J] IS the kernels' gradient layout -> written in place
            g1d, i1d_ws = gitem, self._buf("i1d_ws", be.irt1d_workspace(cfg, nb))
            elbo = self._buf("elbo", nb)
            nb_e, y_e = nb, self.y                                 # persons the ENCODER kernels are launched over
            if self.amortized:
                H = self.H
                enc = self._enc()
                if (rows is None and eps is None and nb % 8 != 0 and nb >= 8 and isinstance(be, HipBackend)
                        and self.estimator == "pathwise" and self.pad_persons):
                    # phantom persons up to a multiple of 8 for the encoder's kernels only (their dimension-major operands want
                    # 16-byte row starts: 2.0 against 0.87 ms a step at 999 999 x 500 persons); the step kernel below runs over
                    # the shard's own persons, so the phantoms never reach the likelihood, the loss or the item gradients, and
                    # their entries of gloc / graw -- all the encoder's backward sees of a person -- are zeroed
                    self._n_pad = (nb + 7) // 8 * 8
                    nb_e, y_e = self._n_pad, self._padded_y()
                fw = {"h": self._buf("h", nb_e * H), "loc": self._buf("loc", nb_e), "raw": self._buf("raw", nb_e)}
                n_pk = be.norm_enc_pack_floats(cfg) if hasattr(be, "norm_enc_pack_floats") else 0
                if n_pk > 0:
                    fw["packws"] = self._buf("nenc_packws", n_pk)      # the forward's fp16-pair images of W1 (large batches)
                gloc, graw = self._buf("gloc", nb_e), self._buf("graw", nb_e)
                nb_ws = self._buf("nencb_ws", be.norm_enc_bwd_workspace(cfg, nb_e))
                with self._phase("guide_forward"):
                    be.norm_enc_forward(cfg, y_e, rows, nb_e, enc, fw)
                loc, raw = fw["loc"], fw["raw"]
            else:
                loc, raw, gloc, graw = self._gather_pp(rows, nb)
            lists = self._sparse_lists(rows)
            sdk = sdc                                              # captured step: the step count lives in device memory
            with self._phase("irt1d"):
                if lists is not None:                      # mostly-missing responses: observed cells only
                    sp_ws = self._buf("i1d_sp_ws", be.irt1d_sparse_workspace(cfg, lists["n_groups"]))
                    be.irt1d_sparse_grad(cfg, lists, self.gid0, loc, raw, eps, a, self.view("b"), c_un, d_un,
                                         gloc, graw, elbo, g1d, sp_ws, loss=lossslot, **sdk, **self._take_fused_tail())
                else:
                    be.irt1d_grad(cfg, self.y_lik if self.y_lik is not None else self.y, rows, nb, self.gid0, loc, raw, eps, a,
                                  self.view("b"), c_un, d_un,
                                  gloc, graw, elbo, g1d, i1d_ws, loss=lossslot, **sdk, **self._take_fused_tail())
            if self.estimator == "score":
                # score-function gradient of the guide in place of the pathwise one (the step kernel's item gradients and
                # loss stand); the draws are the step's own Philox normals, keyed by the global person id
                with self._phase("irt1d_score"):
                    if eps is None:
                        eps_sf = self._buf("sf_eps", nb)
                        gids = None if rows is None else (rows + self.gid0)
                        be.philox_normals(eps_sf, gids, self.gid0, nb, 1, self.seed, self.t, stream_id)
                    else:
                        eps_sf = eps
                    log_r = self._buf("sf_lr%d" % stream_id if not guide_grads else "sf_lr", nb)
                    base, beta, by_row = self._score_baseline(baseline_buf, guide_grads)
                    be.irt1d_score_grad(nb, scale, elbo, eps_sf, raw, rows, base, beta, by_row, log_r, gloc, graw)
                    self.last_log_r = log_r
            if self.amortized:
                with self._phase("guide_backward"):
                    if nb_e != nb:
                        gloc[nb:nb_e].zero_()
                        graw[nb:nb_e].zero_()
                    be.norm_enc_backward(cfg, y_e, rows, nb_e, enc, fw["h"], gloc, graw,
                                         self.G[self.enc_off0:self.enc_off0 + self.n_enc], nb_ws,
                                         yT=self._item_major_y(rows))
                    self._zero_phantom_head_grads()
            else:
                self._scatter_pp(rows, nb, gloc, graw)
            self.last = {"elbo": elbo, "nb": nb}                # the loss itself came out of the kernel's reduction


def _irt_step(self, lrs, rows=None, b_global=None, eps=None, num_particles=1):
    """IrtEngine.step: the leave-one-out control variate of the score-function estimator needs two passes over the
    particles (log_r of every particle first; the same Philox draws both times: the particle index is the Philox stream);
    everything else is the common step."""
    S = int(num_particles)
    if self.estimator != "score" or self.baseline != "loo" or S < 2:
        return _EngineBase.step(self, lrs, rows=rows, b_global=b_global, eps=eps, num_particles=S)
    rows = self._rows_on_device(rows)
    r = rows[0] if isinstance(rows, (list, tuple)) else rows
    nb = self.n_local if r is None else int(r.numel())
    lr_all = self._buf("sf_lr_all", S * nb)
    for sidx in range(S):
        e = eps[sidx] if isinstance(eps, (list, tuple)) else eps
        self.loss_and_grads(r, b_global, e, sidx, guide_grads=False)
        lr_all[sidx * nb:(sidx + 1) * nb].copy_(self.last_log_r[:nb])
    accG = torch.zeros_like(self.G)
    accP = torch.zeros_like(self.GP) if self.per_person else None
    loo = self._buf("sf_loo", nb)
    for sidx in range(S):
        e = eps[sidx] if isinstance(eps, (list, tuple)) else eps
        self.be.loo_baseline(lr_all, S, nb, sidx, loo)
        self.loss_and_grads(r, b_global, e, sidx, baseline_buf=loo)
        accG.add_(self.G, alpha=1.0 / S)
        if accP is not None:
            accP.add_(self.GP, alpha=1.0 / S)
    self.G.copy_(accG)
    if accP is not None:
        self.GP.copy_(accP)
    self.allreduce()
    self.apply_optim(lrs)
    return self.step_loss()


IrtEngine.step = _irt_step


class HoDinaEngine(_EngineBase):
    """HO-DINA ELBO-gradient step with exact enumeration (VCHoDina / VaeCHoDina, vi.py:894-981).

    Flat parameter buffer: [g_un: J | s_un: J | lam0: K | lam1_un: K | encoder (amortized only)];
    per-person rows theta_local / log theta_scale for the BBVI guide (vi.py:928-929)."""

    pp_names = ("theta_local", "theta_scale")

    def __init__(self, y_u8, q, n_global=None, gid0=0, amortized=False, H=64, encoder_init=None, seed=1234,
                 group=None, backend=None):
        self.be = backend if backend is not None else HipBackend()
        self.y = y_u8.contiguous()
        assert self.y.dtype == torch.uint8 and self.y.dim() == 2
        self.dev = self.y.device
        self.n_local, self.J = self.y.shape
        self.N = int(n_global) if n_global is not None else self.n_local
        self.gid0 = int(gid0)
        q = torch.as_tensor(q, dtype=torch.float32)
        assert q.dim() == 2 and q.shape[1] == self.J
        if not bool(((q == 0) | (q == 1)).all()):
            raise ValueError("the Q-matrix must be binary (the subset test of vi.py:78-81 assumes it)")
        self.K = int(q.shape[0])
        self.q = q.to(self.dev).contiguous()
        self.amortized, self.H = bool(amortized), int(H) if amortized else 0
        # The amortized guide's ENCODER on the MFMA kernels whatever the item count and width (the reference's HO-DINA has 30
        # items: its full-batch step took 0.81 ms at 200 k persons where 32 items take 0.24, tools/hodina_cliffs.py): the
        # encoder and the HO-DINA kernel are separate calls with separate configurations, so only the ENCODER sees phantom
        # items (J_enc, a copy of the responses with zero columns behind) and phantom hidden units (zero fc1 rows, zero head
        # columns whose gradients are zeroed behind the backward call: softplus(0) = log 2) -- IrtEngine.__init__ has the
        # argument; the HO-DINA kernel keeps the problem's own items and responses.
        self.H_model, self.J_enc, self.y_enc = self.H, self.J, self.y
        if self.amortized and isinstance(self.be, HipBackend) and self.pad_encoder:
            if 0 < self.H < 64:
                self.H = 64
            if self.H == 64 and self.J % 4 != 0:
                self.J_enc = (self.J + 3) // 4 * 4
                self.y_enc = torch.cat([self.y, torch.zeros((self.n_local, self.J_enc - self.J), dtype=torch.uint8,
                                                            device=self.dev)], 1).contiguous()
        self.seed, self.group = int(seed), group
        J, K = self.J, self.K
        self.off = {"g": 0, "s": J, "lam0": 2 * J, "lam1": 2 * J + K}
        self.shape = {"g": (1, J), "s": (1, J), "lam0": (1, K), "lam1": (1, K)}
        self.n_item = 2 * J + 2 * K
        o = self.n_item
        if self.amortized:
            o = (self.n_item + 63) // 64 * 64
            self.enc_shapes = {"fc1.weight": (self.H, self.J_enc), "fc1.bias": (self.H,), "fc21.weight": (1, self.H),
                               "fc21.bias": (1,), "fc22.weight": (1, self.H), "fc22.bias": (1,)}
            self.enc_off0 = o
            for k in ENC_KEYS:
                self.off["encoder$$$" + k] = o
                self.shape["encoder$$$" + k] = self.enc_shapes[k]
                o += int(np.prod(self.enc_shapes[k]))
            self.n_enc = o - self.enc_off0
        self._alloc(o, self.n_local, per_person=not self.amortized)
        # vi.py:901-904: lam0 = 0, lam1 = 1 (positive -> log 1 = 0), g = s = 0.1 (interval(0,1) -> logit)
        self.view("g").fill_(float(np.float32(_logit(np.float32(0.1)))))
        self.view("s").fill_(float(np.float32(_logit(np.float32(0.1)))))
        if self.amortized:
            Hm = self.H_model
            if encoder_init is None:
                encoder_init = default_encoder_init(J, 1, Hm, seed)
            for k in ENC_KEYS:
                w = torch.as_tensor(encoder_init[k], dtype=torch.float32)
                if k == "fc1.weight":
                    w = torch.nn.functional.pad(w.reshape(Hm, J), (0, self.J_enc - J, 0, self.H - Hm))
                elif k == "fc1.bias":
                    w = torch.nn.functional.pad(w.reshape(-1), (0, self.H - Hm))
                elif k.endswith("weight"):
                    w = torch.nn.functional.pad(w.reshape(1, Hm), (0, self.H - Hm))
                self.view("encoder$$$" + k).copy_(w.reshape(-1))

    pad_encoder = os.environ.get("VX_PAD_ENCODER", "1") != "0"      # test seam: 0 = the encoder in the problem's own shape

    def unconstrained(self, name, buf=None):
        u = super().unconstrained(name, buf)
        if self.amortized and name.startswith("encoder$$$"):       # (phantom items / hidden units are nobody's business)
            if name == "encoder$$$fc1.weight":
                u = u[:self.H_model, :self.J]
            elif name == "encoder$$$fc1.bias":
                u = u[:self.H_model]
            elif name.endswith("weight"):
                u = u[:, :self.H_model]
        return u

    def _item_major_y_enc(self, rows):
        """The encoder's item-major responses (_item_major_y over the encoder's copy: phantom item rows hold byte 0)."""
        if self.J_enc == self.J:
            return self._item_major_y(rows)
        if rows is not None or self.n_local == 0:
            return None
        if getattr(self, "_yT_enc", None) is None:
            stride = (self.n_local + 63) // 64 * 64
            yT = torch.full((self.J_enc + 1, stride), 254, dtype=torch.uint8, device=self.dev)
            yT[:self.J_enc, :self.n_local] = self.y_enc.t()
            self._yT_enc = yT
        return self._yT_enc

    def names(self):
        out = ["g", "s", "lam0", "lam1"]
        if self.amortized:
            out += ["encoder$$$" + k for k in ENC_KEYS]
        return out

    def all_names(self):
        return self.names() + (list(self.pp_names) if self.per_person else [])

    def param(self, name):
        u = self.unconstrained(name)
        if name in ("g", "s"):
            return torch.sigmoid(u)
        if name in ("lam1", "theta_scale"):
            return torch.exp(u)
        return u.clone()

    def loss_and_grads(self, rows=None, b_global=None, eps=None, stream_id=0):
        be = self.be
        nb = self.n_local if rows is None else int(rows.numel())
        Bg = int(b_global) if b_global is not None else (self.N if rows is None else nb)
        scale = float(self.N) / float(Bg)
        sd = getattr(self, "_step_dev", None)               # captured step: the step count lives in device memory
        sdc = {"step_dev": sd} if sd is not None else {}
        cfg = be.hodina_cfg(self.K, self.J, self.H, scale, self.seed, self.t, stream_id, **sdc)
        elbo = self._buf("elbo", nb)
        ws = self._buf("hd_ws", be.hodina_workspace(cfg, nb))
        lossslot = self.G[self.n_params:self.n_params + 1]
        if self.amortized:
            icfg = be.cfg("irt_2pl", 1, self.J_enc, self.H, 1.0, scale, self.seed, self.t, stream_id)
            enc = self._enc()
            fw = {"h": self._buf("h", nb * self.H), "loc": self._buf("loc", nb), "raw": self._buf("raw", nb)}
            n_pk = be.norm_enc_pack_floats(icfg) if hasattr(be, "norm_enc_pack_floats") else 0
            if n_pk > 0:
                fw["packws"] = self._buf("nenc_packws", n_pk)
            gloc, graw = self._buf("gloc", nb), self._buf("graw", nb)
            nb_ws = self._buf("nencb_ws", be.norm_enc_bwd_workspace(icfg, nb))
            with self._phase("guide_forward"):
                be.norm_enc_forward(icfg, self.y_enc, rows, nb, enc, fw)
            loc, raw = fw["loc"], fw["raw"]
        else:
            loc, raw, gloc, graw = self._gather_pp(rows, nb)
        with self._phase("hodina"):
            be.hodina_grad(cfg, self.y, rows, nb, self.gid0, loc, raw, eps, self.q, self.view("lam0"),
                           self.view("lam1"), self.view("g"), self.view("s"), gloc, graw, elbo,
                           self.G[:self.n_item], ws)
        if self.amortized:
            with self._phase("guide_backward"):
                be.norm_enc_backward(icfg, self.y_enc, rows, nb, enc, fw["h"], gloc, graw,
                                     self.G[self.enc_off0:self.enc_off0 + self.n_enc], nb_ws,
                                     yT=self._item_major_y_enc(rows))
                if self.H != self.H_model:                         # phantom hidden units: their head columns take no gradient
                    for k in ("fc21.weight", "fc22.weight"):
                        self.view("encoder$$$" + k, self.G)[self.H_model:].zero_()
        else:
            self._scatter_pp(rows, nb, gloc, graw)
        be.sum_into(elbo, nb, -scale, lossslot, self.sum_ws, **sdc)   # (a captured step's counter advances here)
        self.last = {"elbo": elbo, "nb": nb}


class CcdmEngine(_EngineBase):
    """Pattern-enumerated DINA / DINO with the uniform pattern prior and an empty guide (VCCDM, vi.py:819-865).
    Flat parameter buffer: [g_un: J | s_un: J]; no per-person state, no random numbers."""

    pp_names = ()

    def __init__(self, y_u8, q, cdm="dina", n_global=None, gid0=0, seed=1234, group=None, backend=None):
        if cdm not in ("dina", "dino"):
            raise ValueError("model must be 'dina' or 'dino' (BaseCDM.CDM_FUN, vi.py:728-731)")
        self.be = backend if backend is not None else HipBackend()
        self.y = y_u8.contiguous()
        assert self.y.dtype == torch.uint8 and self.y.dim() == 2
        self.dev = self.y.device
        self.n_local, self.J = self.y.shape
        self.N = int(n_global) if n_global is not None else self.n_local
        self.gid0 = int(gid0)
        q = torch.as_tensor(q, dtype=torch.float32)
        assert q.dim() == 2 and q.shape[1] == self.J
        if not bool(((q == 0) | (q == 1)).all()):
            raise ValueError("the Q-matrix must be binary (the subset tests of vi.py:78-81, 96-100 assume it)")
        self.K = int(q.shape[0])
        self.q = q.to(self.dev).contiguous()
        self.cdm, self.amortized, self.H = cdm, False, 0
        self.seed, self.group = int(seed), group
        J = self.J
        self.off = {"g": 0, "s": J}
        self.shape = {"g": (1, J), "s": (1, J)}
        self.n_item = 2 * J
        self._alloc(self.n_item, self.n_local, per_person=False)
        self.view("g").fill_(float(np.float32(_logit(np.float32(0.1)))))      # vi.py:844-845: g = s = 0.1
        self.view("s").fill_(float(np.float32(_logit(np.float32(0.1)))))

    def names(self):
        return ["g", "s"]

    def all_names(self):
        return self.names()

    def param(self, name):
        return torch.sigmoid(self.unconstrained(name))

    def loss_and_grads(self, rows=None, b_global=None, eps=None, stream_id=0):
        be = self.be
        nb = self.n_local if rows is None else int(rows.numel())
        Bg = int(b_global) if b_global is not None else (self.N if rows is None else nb)
        scale = float(self.N) / float(Bg)
        sd = getattr(self, "_step_dev", None)               # captured step: the loss sum advances the device counter
        sdc = {"step_dev": sd} if sd is not None else {}
        cfg = be.hodina_cfg(self.K, self.J, 0, scale, self.seed, self.t, stream_id)
        elbo = self._buf("elbo", nb)
        ws = self._buf("cd_ws", be.ccdm_workspace(cfg, nb))
        with self._phase("ccdm"):
            be.ccdm_grad(cfg, self.cdm == "dino", self.y, rows, nb, self.q, self.view("g"), self.view("s"), elbo,
                         self.G[:self.n_item], ws)
        be.sum_into(elbo, nb, -scale, self.G[self.n_params:self.n_params + 1], self.sum_ws, **sdc)
        self.last = {"elbo": elbo, "nb": nb}


class VaeCcdmEngine(_EngineBase):
    """VaeCCDM (vi.py:866-891): pattern-enumerated DINA / DINO with the SoftmaxEncoder prior (vi.py:473-485).  The encoder's
    softmax runs over the BATCH (dim = 0), so a step needs three batch-wide reductions of C = 2^K floats (column max, column
    sum of exponentials, column sum of the prior gradients) -- all-reduced over the ranks when the persons are sharded.
    Missing responses stay in the observation as -1, as in the reference.  Flat buffer: [g_un: J | s_un: J | encoder]."""

    pp_names = ()

    def __init__(self, y_u8, q, cdm="dina", n_global=None, gid0=0, H=64, encoder_init=None, seed=1234, group=None,
                 backend=None):
        if cdm not in ("dina", "dino"):
            raise ValueError("model must be 'dina' or 'dino' (BaseCDM.CDM_FUN, vi.py:728-731)")
        self.be = backend if backend is not None else HipBackend()
        self.y = y_u8.contiguous()
        assert self.y.dtype == torch.uint8 and self.y.dim() == 2
        self.dev = self.y.device
        self.n_local, self.J = self.y.shape
        self.N = int(n_global) if n_global is not None else self.n_local
        self.gid0 = int(gid0)
        q = torch.as_tensor(q, dtype=torch.float32)
        assert q.dim() == 2 and q.shape[1] == self.J
        if not bool(((q == 0) | (q == 1)).all()):
            raise ValueError("the Q-matrix must be binary (the subset tests of vi.py:78-81, 96-100 assume it)")
        self.K = int(q.shape[0])
        self.C = 1 << self.K
        self.q = q.to(self.dev).contiguous()
        self.cdm, self.amortized, self.H = cdm, True, int(H)
        self.seed, self.group = int(seed), group
        J, C = self.J, self.C
        self.off = {"g": 0, "s": J}
        self.shape = {"g": (1, J), "s": (1, J)}
        self.n_item = 2 * J
        o = (self.n_item + 63) // 64 * 64
        self.enc_shapes = {"fc1.weight": (self.H, J), "fc1.bias": (self.H,), "fc2.weight": (C, self.H), "fc2.bias": (C,)}
        self.enc_off0 = o
        for k in BIN_ENC_KEYS:
            self.off["encoder$$$" + k] = o
            self.shape["encoder$$$" + k] = self.enc_shapes[k]
            o += int(np.prod(self.enc_shapes[k]))
        self.n_enc = o - self.enc_off0
        self._alloc(o, self.n_local, per_person=False)
        self.view("g").fill_(float(np.float32(_logit(np.float32(0.1)))))      # vi.py:875-876
        self.view("s").fill_(float(np.float32(_logit(np.float32(0.1)))))
        if encoder_init is None:
            encoder_init = default_bin_encoder_init(J, C, self.H, seed)       # same nn.Linear shapes with C outputs
        for k in BIN_ENC_KEYS:
            self.view("encoder$$$" + k).copy_(torch.as_tensor(encoder_init[k], dtype=torch.float32).reshape(-1))

    def names(self):
        return ["g", "s"] + ["encoder$$$" + k for k in BIN_ENC_KEYS]

    def all_names(self):
        return self.names()

    def param(self, name):
        u = self.unconstrained(name)
        return torch.sigmoid(u) if name in ("g", "s") else u.clone()

    def _allreduce_small(self, t, op):
        if self.group is not None:
            if t.is_cuda and torch.distributed.get_backend(self.group) == "gloo":
                host = t.cpu()
                torch.distributed.all_reduce(host, op=op, group=self.group)
                t.copy_(host)
            else:
                torch.distributed.all_reduce(t, op=op, group=self.group)

    def loss_and_grads(self, rows=None, b_global=None, eps=None, stream_id=0):
        be = self.be
        nb = self.n_local if rows is None else int(rows.numel())
        Bg = int(b_global) if b_global is not None else (self.N if rows is None else nb)
        scale = float(self.N) / float(Bg)
        C, H = self.C, self.H
        sd = getattr(self, "_step_dev", None)               # captured step: the loss sum advances the device counter
        sdc = {"step_dev": sd} if sd is not None else {}
        cfg = be.hodina_cfg(self.K, self.J, H, scale, self.seed, self.t, stream_id)
        enc = {k: self.view("encoder$$$" + k) for k in BIN_ENC_KEYS}
        h, z, gla = self._buf("vc_h", nb * H), self._buf("vc_z", nb * C), self._buf("vc_gla", nb * C)
        m, Z, T = self._buf("vc_m", C), self._buf("vc_Z", C), self._buf("vc_T", C)
        cws = self._buf("vc_cws", be.col_reduce_workspace(max(nb, 1), C))
        elbo = self._buf("elbo", nb)
        ws = self._buf("vc_ws", be.vaeccdm_workspace(cfg, nb))
        with self._phase("guide_forward"):
            be.sm_enc_forward(cfg, self.y, rows, nb, enc, h, z)
            be.col_reduce(0, z, nb, C, None, m, cws)                      # column maximum over the batch ...
            self._allreduce_small(m[:C], torch.distributed.ReduceOp.MAX)  # ... of every rank
            be.col_reduce(1, z, nb, C, m, Z, cws)
            self._allreduce_small(Z[:C], torch.distributed.ReduceOp.SUM)
            off = self._buf("vc_off", C)
            torch.add(m[:C], torch.log(Z[:C]), out=off[:C])               # attr_p = exp(z - off)
        with self._phase("vaeccdm"):
            be.vaeccdm_grad(cfg, self.cdm == "dino", self.y, rows, nb, self.q, self.view("g"), self.view("s"), z, off, elbo,
                            gla, self.G[:self.n_item], ws)
        with self._phase("guide_backward"):
            be.col_reduce(2, gla, nb, C, None, T, cws)
            self._allreduce_small(T[:C], torch.distributed.ReduceOp.SUM)
            bws = self._buf("vc_bws", be.sm_enc_bwd_workspace(cfg, nb))
            be.sm_enc_backward(cfg, self.y, rows, nb, enc, h, z, off, T, gla,
                               self.G[self.enc_off0:self.enc_off0 + self.n_enc], bws)
        be.sum_into(elbo, nb, -scale, self.G[self.n_params:self.n_params + 1], self.sum_ws, **sdc)
        self.last = {"elbo": elbo, "nb": nb}


BIN_ENC_KEYS = ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias")
CDM_REF_PRIOR = 1.5          # vi.py:753: Bernoulli(torch.ones(..) + 0.5) -- probability 1.5, clamped by log_prob


class CdmSfEngine(_EngineBase):
    """Bernoulli-guide DINA / DINO with the score-function (REINFORCE) estimator: VCDM / VaeCDM (vi.py:726-816).

    Flat parameter buffer: [g_un: J | s_un: J | BinEncoder (amortized only)]; per-person rows `attr_p` (n, K) logits of the
    guide probabilities for VCDM (vi.py:811).  baseline: 'none' = pyro's Trace_ELBO; 'avg' = per-person decaying average
    of log_r (rate `baseline_beta`); 'loo' = leave-one-out mean over the particles of a step (all particles then share the
    step's subsample).  attr_prior: probability of mastering an attribute under the model prior; None reproduces the
    reference's Bernoulli(1.5)."""

    pp_names = ("attr_p",)

    def __init__(self, y_u8, q, cdm="dina", n_global=None, gid0=0, amortized=False, H=64, encoder_init=None, seed=1234,
                 group=None, backend=None, attr_prior=None, baseline="none", baseline_beta=0.9):
        if cdm not in ("dina", "dino"):
            raise ValueError("model must be 'dina' or 'dino' (BaseCDM.CDM_FUN, vi.py:728-731)")
        if baseline not in ("none", "avg", "loo"):
            raise ValueError("baseline must be 'none', 'avg' or 'loo'")
        self.be = backend if backend is not None else HipBackend()
        self.y = y_u8.contiguous()
        assert self.y.dtype == torch.uint8 and self.y.dim() == 2
        if bool((self.y > 1).any()):
            raise ValueError("VCDM / VaeCDM need complete 0/1 responses: the reference hands them to the likelihood "
                             "unmasked (vi.py:756), a missing cell makes its loss NaN")
        self.dev = self.y.device
        self.n_local, self.J = self.y.shape
        self.N = int(n_global) if n_global is not None else self.n_local
        self.gid0 = int(gid0)
        q = torch.as_tensor(q, dtype=torch.float32)
        assert q.dim() == 2 and q.shape[1] == self.J
        if not bool(((q == 0) | (q == 1)).all()):
            raise ValueError("the Q-matrix must be binary (the subset tests of vi.py:78-81, 96-100 assume it)")
        self.K = int(q.shape[0])
        self.q = q.to(self.dev).contiguous()
        self.cdm, self.amortized, self.H = cdm, bool(amortized), int(H) if amortized else 0
        self.seed, self.group = int(seed), group
        self.attr_prior = CDM_REF_PRIOR if attr_prior is None else float(attr_prior)
        self.baseline, self.baseline_beta = baseline, float(baseline_beta)
        J, K = self.J, self.K
        self.off = {"g": 0, "s": J}
        self.shape = {"g": (1, J), "s": (1, J)}
        self.n_item = 2 * J
        o = self.n_item
        if self.amortized:
            o = (self.n_item + 63) // 64 * 64
            self.enc_shapes = {"fc1.weight": (self.H, J), "fc1.bias": (self.H,), "fc2.weight": (K, self.H), "fc2.bias": (K,)}
            self.enc_off0 = o
            for k in BIN_ENC_KEYS:
                self.off["encoder$$$" + k] = o
                self.shape["encoder$$$" + k] = self.enc_shapes[k]
                o += int(np.prod(self.enc_shapes[k]))
            self.n_enc = o - self.enc_off0
        self.pp_shape = {"attr_p": (self.n_local, K)}
        self._alloc(o, self.n_local, per_person=not self.amortized)
        self.view("g").fill_(float(np.float32(_logit(np.float32(0.1)))))      # vi.py:748-749: g = s = 0.1
        self.view("s").fill_(float(np.float32(_logit(np.float32(0.1)))))
        if self.amortized:
            if encoder_init is None:
                encoder_init = default_bin_encoder_init(J, K, self.H, seed)
            for k in BIN_ENC_KEYS:
                self.view("encoder$$$" + k).copy_(torch.as_tensor(encoder_init[k], dtype=torch.float32).reshape(-1))
        self.base = torch.zeros(max(self.n_local, 1), dtype=torch.float32, device=self.dev) if baseline == "avg" else None

    def names(self):
        return ["g", "s"] + (["encoder$$$" + k for k in BIN_ENC_KEYS] if self.amortized else [])

    def all_names(self):
        return self.names() + (list(self.pp_names) if self.per_person else [])

    def param(self, name):
        u = self.unconstrained(name)
        if name in ("g", "s"):
            return torch.sigmoid(u)
        if name == "attr_p":                                                   # unit_interval: clamped sigmoid
            return torch.clamp(torch.sigmoid(u), min=float(np.finfo(np.float32).tiny), max=1.0 - float(np.finfo(np.float32).eps))
        return u.clone()

    def loss_and_grads(self, rows=None, b_global=None, eps=None, stream_id=0, baseline_buf=None, grads=True):
        """One particle.  `eps`: uint8 (nb, K) attribute draws to replay, or None (drawn in the kernel from Philox, keyed
        by the global person id).  baseline_buf: explicit control variate in batch order (the 'loo' mode of step())."""
        be = self.be
        nb = self.n_local if rows is None else int(rows.numel())
        Bg = int(b_global) if b_global is not None else (self.N if rows is None else nb)
        scale = float(self.N) / float(Bg)
        K = self.K
        sd = getattr(self, "_step_dev", None)               # captured step: the step count lives in device memory
        sdc = {"step_dev": sd} if sd is not None else {}
        cfg = be.hodina_cfg(K, self.J, self.H, scale, self.seed, self.t, stream_id, **sdc)
        ws = self._buf("cs_ws", be.cdm_sf_workspace(cfg, nb))
        gu, log_r = self._buf("cs_gu", nb * K), self._buf("cs_lr%d" % stream_id if not grads else "cs_lr", nb)
        if self.amortized:
            enc = {k: self.view("encoder$$$" + k) for k in BIN_ENC_KEYS}
            h, u = self._buf("cs_h", nb * self.H), self._buf("cs_u", nb * K)
            with self._phase("guide_forward"):
                be.bin_enc_forward(cfg, self.y, rows, nb, enc, h, u)
        else:
            u = self.PP if rows is None else self.PP.reshape(self.n_local, K)[rows].contiguous().reshape(-1)
        if baseline_buf is not None:
            base, beta, by_row = baseline_buf, -1.0, 0
        elif self.baseline == "avg":
            base, beta, by_row = self.base, (self.baseline_beta if grads else -1.0), 1
        else:
            base, beta, by_row = None, -1.0, 0
        with self._phase("cdm_sf"):
            be.cdm_sf_grad(cfg, self.cdm == "dino", not self.amortized, self.attr_prior, self.y, rows, nb, self.gid0, self.q,
                           self.view("g"), self.view("s"), u, eps, base, beta, by_row, gu, log_r, None,
                           self.G[:self.n_item], ws)
        if grads:
            if self.amortized:
                with self._phase("guide_backward"):
                    bw = self._buf("cs_bws", be.bin_enc_bwd_workspace(cfg, nb))
                    be.bin_enc_backward(cfg, self.y, rows, nb, enc, h, gu,
                                        self.G[self.enc_off0:self.enc_off0 + self.n_enc], bw)
            elif rows is None:
                self.GP.copy_(gu[:nb * K])
            else:                                              # dense per-person gradient: zero off the batch (vi.py:811)
                self.GP.zero_()
                self.GP.reshape(self.n_local, K).index_add_(0, rows, gu[:nb * K].reshape(nb, K))
        be.sum_into(log_r, nb, -1.0, self.G[self.n_params:self.n_params + 1], self.sum_ws, **(sdc if grads else {}))
        self.last = {"log_r": log_r, "gu": gu, "nb": nb}

    def step(self, lrs, rows=None, b_global=None, eps=None, num_particles=1):
        S = int(num_particles)
        if self.baseline != "loo" or S < 2:
            return super().step(lrs, rows=rows, b_global=b_global, eps=eps, num_particles=S)
        # leave-one-out control variate: a first pass records log_r of every particle (same subsample, the same Philox
        # draws as the second pass: the particle index is the Philox stream), the second pass uses the means
        rows = self._rows_on_device(rows)
        r = rows[0] if isinstance(rows, (list, tuple)) else rows
        nb = self.n_local if r is None else int(r.numel())
        lr_all = self._buf("cs_lr_all", S * nb)
        for sidx in range(S):
            e = eps[sidx] if isinstance(eps, (list, tuple)) else eps
            self.loss_and_grads(r, b_global, e, sidx, grads=False)
            lr_all[sidx * nb:(sidx + 1) * nb].copy_(self.last["log_r"][:nb])
        accG = torch.zeros_like(self.G)
        accP = torch.zeros_like(self.GP) if self.per_person else None
        loo = self._buf("cs_loo", nb)
        for sidx in range(S):
            e = eps[sidx] if isinstance(eps, (list, tuple)) else eps
            self.be.loo_baseline(lr_all, S, nb, sidx, loo)
            self.loss_and_grads(r, b_global, e, sidx, baseline_buf=loo)
            accG.add_(self.G, alpha=1.0 / S)
            if accP is not None:
                accP.add_(self.GP, alpha=1.0 / S)
        self.G.copy_(accG)
        if accP is not None:
            self.GP.copy_(accP)
        self.allreduce()
        self.apply_optim(lrs)
        return self.step_loss()


def default_bin_encoder_init(J, K, H, seed):
    """nn.Linear default initialisation for BinEncoder (vi.py:458-470), drawn on the host."""
    g = torch.Generator().manual_seed(int(seed) + 7919)

    def lin(out_f, in_f):
        k = 1.0 / math.sqrt(in_f)
        return (torch.rand(out_f, in_f, generator=g) * 2 - 1) * k, (torch.rand(out_f, generator=g) * 2 - 1) * k
    w1, b1 = lin(H, J)
    w2, b2 = lin(K, H)
    return {"fc1.weight": w1, "fc1.bias": b1, "fc2.weight": w2, "fc2.bias": b2}


class _Phase(object):
    """Brackets a phase with HIP events on the current stream when bench.py asks for it."""

    def __init__(self, sink, name):
        self.sink, self.name = sink, name

    def __enter__(self):
        if self.sink is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.sink is not None:
            self.e1.record()
            self.sink.append((self.name, self.e0, self.e1))
        return False


def _merge_segments(segs):
    segs = sorted(segs)
    out = []
    for b, e, lr in segs:
        if out and out[-1][1] == b and out[-1][2] == lr:
            out[-1] = (out[-1][0], e, lr)
        else:
            out.append((b, e, lr))
    return out


def default_encoder_init(J, D, H, seed):
    """nn.Linear default initialisation (kaiming-uniform with a = sqrt(5) => U(-1/sqrt(fan_in), +))
    for the encoder of vi.py:417-455, drawn on the host."""
    g = torch.Generator().manual_seed(int(seed) + 7919)
    T = D * (D + 1) // 2 if D > 1 else 1

    def lin(out_f, in_f):
        k = 1.0 / math.sqrt(in_f)
        w = (torch.rand(out_f, in_f, generator=g) * 2 - 1) * k
        b = (torch.rand(out_f, generator=g) * 2 - 1) * k
        return w, b
    w1, b1 = lin(H, J)
    w21, b21 = lin(D, H)
    w22, b22 = lin(T, H)
    return {"fc1.weight": w1, "fc1.bias": b1, "fc21.weight": w21, "fc21.bias": b21, "fc22.weight": w22,
            "fc22.bias": b22}
