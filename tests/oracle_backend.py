"""A stand-in for vipsy_amd.engine.HipBackend that computes with the numpy oracle on CPU tensors.
TEST INFRASTRUCTURE: it lets the world_size-2 gloo tests exercise the engine's host logic (person
sharding, global-id RNG keys, plate scale from the GLOBAL batch, flat-buffer all-reduce, replicated
Adam) without a GPU.  It is never importable from the product package."""
import numpy as np
import torch

from oracle import vi_oracle as vo
from vipsy_amd import _hip
from vipsy_amd.engine import MODEL_CODE

CODE_MODEL = {v: k for k, v in MODEL_CODE.items()}


def _np(t, shape=None, dtype=np.float64):
    if t is None:
        return None
    a = t.detach().cpu().numpy().astype(dtype)
    return a.reshape(shape) if shape is not None else a


def _put(dst, arr):
    dst[:arr.size].copy_(torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32).reshape(-1)))


class OracleBackend(object):
    name = "oracle"

    @staticmethod
    def cfg(model, D, J, H, Dc, scale, seed, step, stream):
        return _hip.IrtCfg(MODEL_CODE[model], D, J, H, Dc, scale, seed, step, stream)

    @staticmethod
    def _rows(rows, nb):
        return np.arange(nb) if rows is None else rows.cpu().numpy()

    def irt1d_workspace(self, cfg, nb):
        return 1

    def irt1d_grad(self, cfg, y, rows, nb, gid0, loc, raw, eps_in, a, b, c_un, d_un, gloc, graw, elbo, gitem, ws,
                   loss=None):
        model, J = CODE_MODEL[cfg.model], cfg.J
        r = self._rows(rows, nb)
        yy = y.cpu().numpy()[r]
        eps = _np(eps_in, (nb, 1)) if eps_in is not None else vo.philox_normals(cfg.seed, cfg.step, cfg.stream, gid0 + r, 1).astype(np.float64)
        lc, rw = _np(loc)[:nb].reshape(nb, 1), _np(raw)[:nb].reshape(nb, 1)
        sig = np.exp(rw)
        x = lc + sig * eps
        c = vo.sigmoid(_np(c_un, (1, J))) if c_un is not None else None
        d = vo.sigmoid(_np(d_un, (1, J))) if d_un is not None else None
        ll, g = vo.irt_loglik(model, x, _np(a, (1, J)) if a is not None else None, _np(b, (1, J)), c, d, cfg.Dc, yy)
        gxt = cfg.scale * (g["x"] - x)
        _put(gloc, -gxt)
        _put(graw, -(gxt * sig * eps + cfg.scale))
        el = ll + (-0.5 * x ** 2 + 0.5 * eps ** 2 + rw)[:, 0]
        _put(elbo, el)
        if loss is not None:
            loss[0] = -cfg.scale * float(el.sum())
        out = np.zeros(4 * J)
        if "a" in g:
            out[0:J] = -cfg.scale * g["a"].reshape(-1)
        out[J:2 * J] = -cfg.scale * g["b"].reshape(-1)
        if "c" in g:
            out[2 * J:3 * J] = -cfg.scale * (g["c"] * c * (1 - c)).reshape(-1)
        if "d" in g:
            out[3 * J:4 * J] = -cfg.scale * (g["d"] * d * (1 - d)).reshape(-1)
        _put(gitem, out)

    def sum_into(self, v, n, alpha, out, ws, step_dev=None):
        out[0] = float(alpha) * float(v[:n].double().sum())

    def sum2_into(self, v1, v2, n, alpha, out, ws):
        out[0] = float(alpha) * (float(v1[:n].double().sum()) + float(v2[:n].double().sum()))

    def adam(self, p, g, m, v, free, n, segs, t, betas=(0.9, 0.999), eps=1e-8):
        b1, b2 = betas
        for (lo, hi, lr) in segs:
            gi = g[lo:hi] * (free[lo:hi] if free is not None else 1.0)
            m[lo:hi] = b1 * m[lo:hi] + (1 - b1) * gi
            v[lo:hi] = b2 * v[lo:hi] + (1 - b2) * gi * gi
            denom = v[lo:hi].sqrt() / np.sqrt(1 - b2 ** t) + eps
            p[lo:hi] -= (lr / (1 - b1 ** t)) * m[lo:hi] / denom

    def adam2(self, bufA, free, nA, segsA, bufB, nB, segsB, t, betas=(0.9, 0.999), eps=1e-8):
        self.adam(bufA[0], bufA[1], bufA[2], bufA[3], free, nA, segsA, t, betas, eps)
        self.adam(bufB[0], bufB[1], bufB[2], bufB[3], None, nB, segsB, t, betas, eps)

    # ---- amortized MVN guide + D >= 2 likelihood (restated through oracle.irt_particle pieces) ----------
    def mvn_enc_forward(self, cfg, y, rows, nb, gid0, enc, eps_in, out):
        D, J, H = cfg.D, cfg.J, cfg.H
        r = self._rows(rows, nb)
        yy = y.cpu().numpy()[r]
        T = D * (D + 1) // 2
        W = {"fc1.weight": _np(enc["fc1.weight"], (H, J)), "fc1.bias": _np(enc["fc1.bias"]),
             "fc21.weight": _np(enc["fc21.weight"], (D, H)), "fc21.bias": _np(enc["fc21.bias"]),
             "fc22.weight": _np(enc["fc22.weight"], (T, H)), "fc22.bias": _np(enc["fc22.bias"])}
        eps = _np(eps_in, (nb, D)) if eps_in is not None else vo.philox_normals(cfg.seed, cfg.step, cfg.stream, gid0 + r, D).astype(np.float64)
        loc, raw, cache = vo.enc_forward(W, vo.enc_input(yy, np.float64))
        rr, cc = vo.tril_rows_cols(D)
        M = np.zeros((nb, D, D))
        M[:, rr, cc] = raw
        diag = np.einsum("bii->bi", M)
        L = np.tril(M, -1) + np.einsum("bi,ij->bij", np.exp(diag), np.eye(D))
        x = loc + np.einsum("bij,bj->bi", L, eps)
        _put(out["h"], cache[2]); _put(out["x"], x); _put(out["eps"], eps)
        _put(out["ldT"], np.exp(diag).T.copy()); _put(out["ent"], 0.5 * (eps ** 2).sum(1) + diag.sum(1))
        _put(out["hT"], cache[2].T.copy()); _put(out["epsT"], eps.T.copy())

    def mvn_pack_floats(self, cfg):
        return 1

    def lik_ximg_bytes(self, cfg, nb):
        return 0

    def mvn_enc_bwd_layout(self, cfg, nb):
        return 0                                            # person-major gx is what this backend's backward reads

    def lik_workspace(self, cfg, nb):
        return 1

    def lik_grad(self, cfg, y, rows, nb, x, a, b, c_un, d_un, gx, ll, gitem, ws, gxT=None, yT=None, ximg=None, epsT=None,
                 ldT=None, gdT=None):
        model, D, J = CODE_MODEL[cfg.model], cfg.D, cfg.J
        r = self._rows(rows, nb)
        yy = y.cpu().numpy()[r]
        xx = _np(x)[:nb * D].reshape(nb, D)
        c = vo.sigmoid(_np(c_un, (1, J))) if c_un is not None else None
        d = vo.sigmoid(_np(d_un, (1, J))) if d_un is not None else None
        lls, g = vo.irt_loglik(model, xx, _np(a, (D, J)), _np(b, (1, J)), c, d, cfg.Dc, yy)
        _put(gx, cfg.scale * (g["x"] - xx))
        if gxT is not None:
            _put(gxT, (cfg.scale * (g["x"] - xx)).T.copy())
        _put(ll, lls - 0.5 * (xx ** 2).sum(1))
        out = np.zeros(D * J + 3 * J)
        out[:D * J] = -cfg.scale * g["a"].reshape(-1)
        out[D * J:D * J + J] = -cfg.scale * g["b"].reshape(-1)
        if "c" in g:
            out[D * J + J:D * J + 2 * J] = -cfg.scale * (g["c"] * c * (1 - c)).reshape(-1)
        if "d" in g:
            out[D * J + 2 * J:] = -cfg.scale * (g["d"] * d * (1 - d)).reshape(-1)
        _put(gitem, out)

    def mvn_enc_bwd_workspace(self, cfg, nb):
        return 1

    def mvn_enc_bwd_hs_offset(self, cfg, nb):
        return -1

    def mvn_enc_bwd_gd_offset(self, cfg, nb):
        return -1

    def mvn_enc_backward(self, cfg, y, rows, nb, enc, fw, gx, genc, ws, gxT=None, gd_ready=False):
        D, J, H = cfg.D, cfg.J, cfg.H
        T = D * (D + 1) // 2
        r = self._rows(rows, nb)
        yy = y.cpu().numpy()[r]
        W = {"fc1.weight": _np(enc["fc1.weight"], (H, J)), "fc1.bias": _np(enc["fc1.bias"]),
             "fc21.weight": _np(enc["fc21.weight"], (D, H)), "fc21.bias": _np(enc["fc21.bias"]),
             "fc22.weight": _np(enc["fc22.weight"], (T, H)), "fc22.bias": _np(enc["fc22.bias"])}
        yin = vo.enc_input(yy, np.float64)
        pre = yin @ W["fc1.weight"].T + W["fc1.bias"]
        h = vo.softplus(pre)
        g_x = _np(gx)[:nb * D].reshape(nb, D)
        eps = _np(fw["eps"])[:nb * D].reshape(nb, D)
        ld = _np(fw["ldT"])[:nb * D].reshape(D, nb).T
        gM = np.tril(np.einsum("bi,bj->bij", g_x, eps))
        dg = np.einsum("bii->bi", gM) * ld + cfg.scale
        gM = np.tril(gM, -1) + np.einsum("bi,ij->bij", dg, np.eye(D))
        rr, cc = vo.tril_rows_cols(D)
        ge = vo.enc_backward(W, (yin, pre, h), -g_x, -gM[:, rr, cc])
        _put(genc, np.concatenate([ge[k].reshape(-1) for k in vo.ENC_KEYS]))
