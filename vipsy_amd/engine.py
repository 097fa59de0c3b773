"""The training-step engine: what `svi.step(data)` does in the reference (vi.py:503-516), driven
through the C ABI (include/vipsy_amd.h).  One process per GPU; persons are sharded by contiguous
ranges, item / encoder parameters are replicated, and the only exchange per step is one all-reduce
(RCCL via torch.distributed) of the flat gradient buffer [item grads | encoder grads | loss].

torch here is plumbing: it owns device buffers and the process group.  All arithmetic of the step
is in the HIP kernels behind vipsy_amd/_hip.py.
"""
import ctypes
import math

import numpy as np
import torch

from . import _hip

MODEL_CODE = {"irt_1pl": 1, "irt_2pl": 2, "irt_3pl": 3, "irt_4pl": 4}       # vi.py:538-543
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
    def cfg(model, D, J, H, Dc, scale, seed, step, stream):
        return _hip.IrtCfg(MODEL_CODE[model], D, J, H, Dc, scale, seed, step, stream)

    def mvn_enc_forward(self, cfg, y, rows, nb, gid0, enc, eps_in, out):
        rc = self.L.vx_mvn_enc_forward(ctypes.byref(cfg), _hip.ptr(y), _hip.ptr(rows), nb, gid0,
                                       _hip.ptr(enc["fc1.weight"]), _hip.ptr(enc["fc1.bias"]),
                                       _hip.ptr(enc["fc21.weight"]), _hip.ptr(enc["fc21.bias"]),
                                       _hip.ptr(enc["fc22.weight"]), _hip.ptr(enc["fc22.bias"]),
                                       _hip.ptr(eps_in), _hip.ptr(out["h"]), _hip.ptr(out["x"]),
                                       _hip.ptr(out["eps"]), _hip.ptr(out["ldT"]), _hip.ptr(out["ent"]),
                                       _hip.stream_ptr())
        _hip.check(rc, "vx_mvn_enc_forward")

    def lik_workspace(self, cfg, nb):
        n = self.L.vx_irt_lik_workspace_floats(ctypes.byref(cfg), nb)
        if n < 0:
            raise _hip.VxError("vx_irt_lik_workspace_floats: unsupported configuration (code %d)" % n)
        return n

    def lik_grad(self, cfg, y, rows, nb, x, a, b, c_un, d_un, gx, ll, gitem, ws):
        rc = self.L.vx_irt_lik_grad(ctypes.byref(cfg), _hip.ptr(y), _hip.ptr(rows), nb, _hip.ptr(x), _hip.ptr(a),
                                    _hip.ptr(b), _hip.ptr(c_un), _hip.ptr(d_un), _hip.ptr(gx), _hip.ptr(ll),
                                    _hip.ptr(gitem), _hip.ptr(ws), _hip.stream_ptr())
        _hip.check(rc, "vx_irt_lik_grad")

    def mvn_enc_bwd_workspace(self, cfg, nb):
        n = self.L.vx_mvn_enc_bwd_workspace_floats(ctypes.byref(cfg), nb)
        if n < 0:
            raise _hip.VxError("vx_mvn_enc_bwd_workspace_floats: unsupported configuration (code %d)" % n)
        return n

    def mvn_enc_backward(self, cfg, y, rows, nb, enc, fw, gx, genc, ws):
        rc = self.L.vx_mvn_enc_backward(ctypes.byref(cfg), _hip.ptr(y), _hip.ptr(rows), nb,
                                        _hip.ptr(enc["fc21.weight"]), _hip.ptr(enc["fc22.weight"]),
                                        _hip.ptr(fw["h"]), _hip.ptr(fw["eps"]), _hip.ptr(fw["ldT"]), _hip.ptr(gx),
                                        _hip.ptr(genc), _hip.ptr(ws), _hip.stream_ptr())
        _hip.check(rc, "vx_mvn_enc_backward")

    def irt1d_workspace(self, cfg, nb):
        n = self.L.vx_irt1d_workspace_floats(ctypes.byref(cfg), nb)
        if n < 0:
            raise _hip.VxError("vx_irt1d_workspace_floats: unsupported configuration (code %d)" % n)
        return n

    def irt1d_grad(self, cfg, y, rows, nb, gid0, loc, raw, eps_in, a, b, c_un, d_un, gloc, graw, elbo, gitem, ws):
        rc = self.L.vx_irt1d_grad(ctypes.byref(cfg), _hip.ptr(y), _hip.ptr(rows), nb, gid0, _hip.ptr(loc),
                                  _hip.ptr(raw), _hip.ptr(eps_in), _hip.ptr(a), _hip.ptr(b), _hip.ptr(c_un),
                                  _hip.ptr(d_un), _hip.ptr(gloc), _hip.ptr(graw), _hip.ptr(elbo), _hip.ptr(gitem),
                                  _hip.ptr(ws), _hip.stream_ptr())
        _hip.check(rc, "vx_irt1d_grad")

    def sum_into(self, v, n, alpha, out, ws):
        rc = self.L.vx_sum(_hip.ptr(v), n, alpha, _hip.ptr(out), _hip.ptr(ws), _hip.stream_ptr())
        _hip.check(rc, "vx_sum")

    def adam(self, p, g, m, v, free, n, segs, t, betas=(0.9, 0.999), eps=1e-8):
        arr = (_hip.AdamSeg * len(segs))(*[_hip.AdamSeg(b, e, lr, 0.0) for (b, e, lr) in segs])
        rc = self.L.vx_adam_step(_hip.ptr(p), _hip.ptr(g), _hip.ptr(m), _hip.ptr(v), _hip.ptr(free), n, arr,
                                 len(segs), t, betas[0], betas[1], eps, _hip.stream_ptr())
        _hip.check(rc, "vx_adam_step")

    def philox_normals(self, out, gids, gid0, n, D, seed, step, stream):
        rc = self.L.vx_philox_normals(_hip.ptr(out), _hip.ptr(gids), gid0, n, D, seed, step, stream,
                                      _hip.stream_ptr())
        _hip.check(rc, "vx_philox_normals")


class LrSpec(object):
    """Per-tensor learning rates + MultiStepLR, as pyro.optim.Adam / PyroLRScheduler give them
    (vi.py:514, 639-640; test.py:321-327, 345-359)."""

    def __init__(self, lr, milestones=(), gamma=0.1, betas=(0.9, 0.999), eps=1e-8):
        self.lr, self.milestones, self.gamma, self.betas, self.eps = lr, tuple(milestones), gamma, betas, eps
        self.epoch = 0

    def lr_of(self, name):
        if callable(self.lr):
            module = name.split("$$$")[0]
            stripped = name.split("$$$")[1] if "$$$" in name else name
            base = self.lr(module, stripped)["lr"]
        elif isinstance(self.lr, dict):
            base = self.lr["lr"]
        else:
            base = self.lr
        k = sum(1 for m in self.milestones if m <= self.epoch)
        return base * self.gamma ** k

    def scheduler_step(self):
        self.epoch += 1


class IrtEngine(object):
    """IRT ELBO-gradient step (VIRT / VaeIRT of the reference, vi.py:536-723) on one rank.

    Parameters live in ONE flat float32 buffer, replicated on every rank:
        [a: D*J | b: J | c_un: J | d_un: J | encoder (nn.Linear order, amortized only)]
    followed in the gradient buffer by one extra slot holding the loss, so a single all-reduce
    carries everything.  Per-person variational rows (BBVI) are sharded and never leave the rank.
    """

    def __init__(self, y_u8, model="irt_2pl", D=1, Dc=1.0, n_global=None, gid0=0, amortized=False, H=64,
                 share_cov=False, a_free=None, a0=None, b0=None, encoder_init=None, seed=1234, group=None,
                 backend=None):
        self.be = backend if backend is not None else HipBackend()
        self.y = y_u8.contiguous()
        assert self.y.dtype == torch.uint8 and self.y.dim() == 2
        self.dev = self.y.device
        self.n_local, self.J = self.y.shape
        self.N = int(n_global) if n_global is not None else self.n_local
        self.gid0 = int(gid0)
        self.model, self.D, self.Dc = model, int(D), float(Dc)
        self.amortized, self.H, self.share_cov = bool(amortized), int(H) if amortized else 0, bool(share_cov)
        self.seed, self.group = int(seed), group
        self.t = 0                       # optimiser step counter (Adam bias correction, Philox step)
        J, Dd = self.J, self.D
        f32 = dict(dtype=torch.float32, device=self.dev)
        # ---- flat parameter buffer ------------------------------------------------------------
        self.off = {"a": 0, "b": Dd * J, "c": Dd * J + J, "d": Dd * J + 2 * J}
        self.n_item = Dd * J + 3 * J
        self.n_enc = 0
        if self.amortized:
            T = Dd * (Dd + 1) // 2 if Dd > 1 else 1
            nloc = Dd
            self.enc_shapes = {"fc1.weight": (self.H, J), "fc1.bias": (self.H,), "fc21.weight": (nloc, self.H),
                               "fc21.bias": (nloc,), "fc22.weight": (T, self.H), "fc22.bias": (T,)}
            o = self.n_item
            for k in ENC_KEYS:
                self.off["encoder$$$" + k] = o
                o += int(np.prod(self.enc_shapes[k]))
            self.n_enc = o - self.n_item
        self.n_params = self.n_item + self.n_enc
        self.P = torch.zeros(self.n_params, **f32)
        self.G = torch.zeros(self.n_params + 1, **f32)          # + loss slot
        self.M = torch.zeros(self.n_params, **f32)
        self.V = torch.zeros(self.n_params, **f32)
        self.free = torch.ones(self.n_params, **f32)
        # reference initial values (vi.py:567-587)
        a_init = torch.ones(Dd, J) if a0 is None else torch.as_tensor(a0, dtype=torch.float32).reshape(Dd, J).clone()
        if a_free is None and Dd > 1:
            af = torch.ones(Dd, J)
            for i in range(Dd):
                af[i, J - i:] = 0                                  # vi.py:570-572
            if a0 is None:
                a_init = a_init * af
            a_free = af
        if model != "irt_1pl":
            self.view("a").copy_(a_init.reshape(-1))
            if a_free is not None:
                self.free[self.off["a"]:self.off["a"] + Dd * J] = torch.as_tensor(a_free, dtype=torch.float32).reshape(-1)
        if b0 is not None:
            self.view("b").copy_(torch.as_tensor(b0, dtype=torch.float32).reshape(-1))
        if model in ("irt_3pl", "irt_4pl"):
            self.view("c").fill_(float(np.float32(_logit(np.float32(0.1)))))
        if model == "irt_4pl":
            self.view("d").fill_(float(np.float32(_logit(np.float32(1.0) - np.float32(0.1)))))
        if self.amortized:
            if encoder_init is None:
                encoder_init = default_encoder_init(J, Dd, self.H, seed)
            for k in ENC_KEYS:
                self.view("encoder$$$" + k).copy_(torch.as_tensor(encoder_init[k], dtype=torch.float32).reshape(-1))
        else:
            # per-person variational rows (vi.py:702-703, 707, 717-721), sharded with the persons
            n = self.n_local
            if Dd == 1:
                self.pp_len = 2 * n                                 # [x_local: n | log x_scale: n]
            else:
                raise NotImplementedError("BBVI with x_feature > 1 is not on the HIP path yet")
            self.PP = torch.zeros(self.pp_len, **f32)
            self.GP = torch.zeros(self.pp_len, **f32)
            self.MP = torch.zeros(self.pp_len, **f32)
            self.VP = torch.zeros(self.pp_len, **f32)
        self._ws = {}
        self.sum_ws = torch.empty(1024, **f32)
        self.last = {}
        self.events = None               # bench.py: list collecting (phase, start_event, end_event)

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
        return out

    def view(self, name, buf=None):
        buf = self.P if buf is None else buf
        o = self.off[name]
        if name == "a":
            return buf[o:o + self.D * self.J]
        if name in ("b", "c", "d"):
            return buf[o:o + self.J]
        k = name.split("$$$")[1]
        return buf[o:o + int(np.prod(self.enc_shapes[k]))]

    def unconstrained(self, name, buf=None):
        if name == "a":
            return self.view(name, buf).reshape(self.D, self.J)
        if name in ("b", "c", "d"):
            return self.view(name, buf).reshape(1, self.J)
        if name == "x_local":
            return (self.PP if buf is None else buf)[:self.n_local].reshape(self.n_local, 1)
        if name == "x_scale":
            return (self.PP if buf is None else buf)[self.n_local:].reshape(self.n_local, 1)
        return self.view(name, buf).reshape(self.enc_shapes[name.split("$$$")[1]])

    def param(self, name):
        u = self.unconstrained(name)
        if name in ("c", "d"):
            return torch.sigmoid(u)
        if name == "x_scale":
            return torch.exp(u)
        return u.clone()

    def _enc(self):
        return {k: self.view("encoder$$$" + k) for k in ENC_KEYS}

    def _buf(self, key, n):
        t = self._ws.get(key)
        if t is None or t.numel() < n:
            t = torch.empty(max(int(n), 1), dtype=torch.float32, device=self.dev)
            self._ws[key] = t
        return t

    def _phase(self, name):
        return _Phase(self.events, name)

    # -- one ELBO-gradient step ------------------------------------------------------------------
    def loss_and_grads(self, rows=None, b_global=None, eps=None, stream_id=0):
        """Fills self.G (flat grads + loss slot) and per-person grads for ONE particle.
        rows: int64 device tensor of LOCAL row indices (None = all local rows, i.e. full batch)."""
        be = self.be
        nb = self.n_local if rows is None else int(rows.numel())
        Bg = int(b_global) if b_global is not None else (self.N if rows is None else nb)
        scale = float(self.N) / float(Bg)
        cfg = be.cfg(self.model, self.D, self.J, self.H, self.Dc, scale, self.seed, self.t, stream_id)
        c_un = self.view("c") if self.model in ("irt_3pl", "irt_4pl") else None
        d_un = self.view("d") if self.model == "irt_4pl" else None
        a = self.view("a") if self.model != "irt_1pl" else None
        gitem = self.G[:self.n_item]
        lossslot = self.G[self.n_params:self.n_params + 1]
        if self.D > 1 and self.amortized:
            D, H = self.D, self.H
            fw = {"h": self._buf("h", nb * H), "x": self._buf("x", nb * D), "eps": self._buf("eps", nb * D),
                  "ldT": self._buf("ldT", nb * D), "ent": self._buf("ent", nb)}
            gx, ll = self._buf("gx", nb * D), self._buf("ll", nb)
            enc = self._enc()
            lik_ws = self._buf("lik_ws", be.lik_workspace(cfg, nb))
            encb_ws = self._buf("encb_ws", be.mvn_enc_bwd_workspace(cfg, nb))
            with self._phase("guide_forward"):
                be.mvn_enc_forward(cfg, self.y, rows, nb, self.gid0, enc, eps, fw)
            with self._phase("likelihood"):
                be.lik_grad(cfg, self.y, rows, nb, fw["x"], a, self.view("b"), c_un, d_un, gx, ll, gitem, lik_ws)
            with self._phase("guide_backward"):
                be.mvn_enc_backward(cfg, self.y, rows, nb, enc, fw, gx, self.G[self.n_item:self.n_params], encb_ws)
            # loss = -scale * sum_i (ll_i + ent_i)
            tmp = self._buf("loss2", 2)
            be.sum_into(ll, nb, -scale, tmp[0:1], self.sum_ws)
            be.sum_into(fw["ent"], nb, -scale, tmp[1:2], self.sum_ws)
            torch.add(tmp[0:1], tmp[1:2], out=lossslot)
            self.last = {"fw": fw, "gx": gx, "ll": ll, "nb": nb}
        elif self.D == 1 and not self.amortized:
            n = self.n_local
            if rows is None:
                loc, raw = self.PP[:n], self.PP[n:]
                gloc, graw = self.GP[:n], self.GP[n:]
            else:
                loc, raw = self.PP[:n][rows].contiguous(), self.PP[n:][rows].contiguous()
                gloc, graw = self._buf("gloc", nb), self._buf("graw", nb)
            elbo = self._buf("elbo", nb)
            g1d, i1d_ws = self._buf("g1d", 4 * self.J), self._buf("i1d_ws", be.irt1d_workspace(cfg, nb))
            with self._phase("irt1d"):
                be.irt1d_grad(cfg, self.y, rows, nb, self.gid0, loc, raw, eps, a, self.view("b"), c_un, d_un,
                              gloc, graw, elbo, g1d, i1d_ws)
            g1 = self._ws["g1d"]
            J = self.J
            gitem.zero_()
            gitem[self.off["a"]:self.off["a"] + J].copy_(g1[0:J])
            gitem[self.off["b"]:self.off["b"] + 3 * J].copy_(g1[J:4 * J])
            if rows is not None:                                  # dense grads over all local rows (App. A.2)
                self.GP.zero_()
                self.GP[:n].index_add_(0, rows, gloc[:nb])
                self.GP[n:].index_add_(0, rows, graw[:nb])
            be.sum_into(elbo, nb, -scale, lossslot, self.sum_ws)
            self.last = {"elbo": elbo, "nb": nb}
        else:
            raise NotImplementedError("guide/model combination not on the HIP path yet")

    def allreduce(self):
        if self.group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            torch.distributed.all_reduce(self.G, group=self.group)

    def apply_optim(self, lrs):
        """Adam on the unconstrained leaves with the `free` mask (vi.py:508-514)."""
        self.t += 1
        segs = []
        for name in self.names():
            v = self.view(name)
            o = self.off[name]
            segs.append((o, o + v.numel(), float(lrs.lr_of(name))))
        segs = _merge_segments(segs)
        self.be.adam(self.P, self.G, self.M, self.V, self.free, self.n_params, segs, self.t, lrs.betas, lrs.eps)
        if not self.amortized:
            n = self.n_local
            segs = _merge_segments([(0, n, float(lrs.lr_of("x_local"))), (n, 2 * n, float(lrs.lr_of("x_scale")))])
            self.be.adam(self.PP, self.GP, self.MP, self.VP, None, self.pp_len, segs, self.t, lrs.betas, lrs.eps)

    def step(self, lrs, rows=None, b_global=None, eps=None, num_particles=1):
        """loss_and_grads + optimiser, the body of SVI.step (vi.py:505-516).  Returns the loss as a
        0-d device tensor (no host sync)."""
        if num_particles == 1:
            self.loss_and_grads(rows, b_global, eps, 0)
        else:
            raise NotImplementedError("num_particles > 1 is handled by vipsy_amd.svi")
        with self._phase("allreduce"):
            self.allreduce()
        loss = self.G[self.n_params].clone()
        with self._phase("optimizer"):
            self.apply_optim(lrs)
        return loss


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
