"""Host-side mirror of the reference's model-class surface (/root/reference/vi.py) for the path that is
accelerated: same class names, constructor kwargs, `.fit(optim, loss, max_iter, random_instance)`
signature and result access (`param(name)` / `clear_param_store()` standing in for pyro.param /
pyro.clear_param_store, test.py:66,73-87), so the reference's demos read the same:

    from vipsy_amd.vi import VaeIRT, Adam, MultiStepLR, Trace_ELBO, param
    model = VaeIRT(data=y, model='irt_2pl', subsample_size=100, x_feature=100)     # test.py:343
    model.fit(optim=MultiStepLR({...}), max_iter=20000, random_instance=ri, loss=Trace_ELBO(num_particles=1))
    a_hat = param('a')

What runs underneath is NOT pyro: every iteration is one vipsy_amd.engine step, i.e. the HIP kernels behind
include/vipsy_amd.h (no CPU fallback).  Every model class of the reference is built: VIRT, VaeIRT, VCDM, VaeCDM, VCCDM,
VaeCCDM, VCHoDina, VaeCHoDina.
"""
import numpy as np
import torch

from .engine import CcdmEngine, CdmSfEngine, IrtEngine, HoDinaEngine, LrSpec, VaeCcdmEngine
from .random_data import (RandomPsyData, RandomIrt1PL, RandomIrt2PL, RandomIrt3PL, RandomIrt4PL, RandomMilIrt2PL,    # noqa: F401
                          RandomMilIrt3PL, RandomMilIrt4PL, RandomDina, RandomDino, RandomHoDina)  # vi.py:120-412 live in vi too

_STORE = {}          # name -> engine that owns the parameter (the process-global store of the reference)


def clear_param_store():
    _STORE.clear()


def param(name):
    """Constrained value of a parameter, like pyro.param(name) (vi.py:644-654; test.py:73-87)."""
    if name not in _STORE:
        raise KeyError(name)
    return _STORE[name].param(name)


# ---- optimiser / loss descriptors (pyro.optim.Adam, pyro.optim.MultiStepLR, pyro.infer.*_ELBO) ------------
class Adam(object):
    """pyro.optim.Adam(optim_args): optim_args is a dict or a callable(module_name, param_name) -> dict
    (test.py:321-327, 345-350).  One Adam state per tensor, betas/eps as torch defaults."""

    def __init__(self, optim_args):
        self.optim_args = optim_args

    def spec(self):
        return LrSpec(self.optim_args)                  # lr / betas / eps per tensor, dict or callable alike


class PyroLRScheduler(object):
    pass


class MultiStepLR(PyroLRScheduler):
    """pyro.optim.MultiStepLR({'optimizer': torch.optim.Adam, 'optim_args': dict|callable, 'milestones': [...],
    'gamma': g}) (test.py:352-359).  The scheduler advances once per iteration (vi.py:639-640)."""

    def __init__(self, args):
        if args.get("optimizer") not in (None, torch.optim.Adam):
            raise NotImplementedError("only torch.optim.Adam is on the HIP path")
        self.args = args

    def spec(self):
        return LrSpec(self.args["optim_args"], milestones=tuple(self.args.get("milestones", ())),
                      gamma=float(self.args.get("gamma", 0.1)))


class Trace_ELBO(object):
    def __init__(self, num_particles=1):
        self.num_particles = int(num_particles)


class TraceEnum_ELBO(Trace_ELBO):
    pass


# ---- data ----------------------------------------------------------------------------------------------
def to_u8(data, device):
    """Reference data contract (float tensor, NaN = missing; vi.py:621) -> uint8 0/1/255 on the device."""
    if isinstance(data, np.ndarray):
        data = torch.from_numpy(data)
    if data.dtype == torch.uint8:
        return data.to(device).contiguous()
    d = data.to(device)
    miss = torch.isnan(d)
    vals = torch.where(miss, torch.zeros_like(d), d)
    if not bool(((vals == 0) | (vals == 1)).all()):
        raise ValueError("responses must be 0/1 (NaN = missing)")
    out = vals.to(torch.uint8)
    out[miss] = 255
    return out.contiguous()


class BasePsy(object):
    """vi.py:521-533.  Multi-GPU is explicit: pass `group` (a torch.distributed process group, e.g.
    torch.distributed.group.WORLD) and `data` is THIS rank's shard of persons, placed in the global plate by
    `sample_size_global` / `gid0`.  Without `group` the object owns its whole problem even when a process group is up
    (independent replications on different ranks, vipsy_amd/harness.py)."""

    def __init__(self, data, subsample_size=None, sample_size_global=None, gid0=0, seed=1234, device=None, group=None,
                 **kwargs):
        if device is None:
            if torch.is_tensor(data) and (data.device.type != "cpu" or kwargs.get("backend") is not None):
                device = data.device
            else:
                device = torch.device("cuda", torch.cuda.current_device())
        dev = torch.device(device)
        self.data = to_u8(data, dev)
        self.sample_size_local = int(self.data.shape[0])
        self.sample_size = int(sample_size_global) if sample_size_global is not None else self.sample_size_local
        self.item_size = int(self.data.shape[1])
        self.subsample_size = int(subsample_size) if subsample_size is not None else self.sample_size
        self.gid0, self.seed, self.device = int(gid0), int(seed), dev
        self.kwargs = kwargs
        self.group = group
        self.world = torch.distributed.get_world_size(group) if group is not None else 1
        if self.world == 1 and self.sample_size != self.sample_size_local:
            raise ValueError("sample_size_global differs from the rows of `data` but no `group` shares the problem")
        self._gen = torch.Generator(device=dev)
        self._gen.manual_seed(self.seed * 7919 + self.gid0)
        self._np_gen = np.random.Generator(np.random.PCG64([self.seed, self.gid0, 7919]))
        # engine keyword arguments shared by every model class (`backend` is the test seam of tests/oracle_backend.py)
        self._eng_kw = {"n_global": self.sample_size, "gid0": self.gid0, "seed": self.seed, "group": group}
        if kwargs.get("backend") is not None:
            self._eng_kw["backend"] = kwargs["backend"]

    def _register(self):
        for n in self.engine.all_names():
            _STORE[n] = self.engine

    def _subsample(self):
        """plate("data", N, subsample_size=B): randperm(N)[:B].  Multi-rank: every
        rank draws B / world of its own shard (stratified, unbiased)."""
        B = self.subsample_size
        if B >= self.sample_size:
            return None, self.sample_size
        b_local = B // self.world if self.world > 1 else B
        b_local = max(1, min(b_local, self.sample_size_local))
        n = self.sample_size_local
        if 16 * b_local <= n:
            # a small subset of many rows: a full permutation on the device is a radix sort of n keys (0.25 ms at 1M, a
            # third of a B = 100 step).  numpy's Generator.choice draws b distinct rows in O(b); same distribution over
            # subsets, ordered like the first b of a random permutation (shuffle=True)
            # (left on the host: the engine copies the indices into the fixed buffer its captured step reads, or moves them itself)
            idx = torch.from_numpy(self._np_gen.choice(n, size=b_local, replace=False, shuffle=True).astype(np.int64))
        else:
            idx = torch.randperm(n, generator=self._gen, device=self.device)[:b_local].contiguous()
        return idx, b_local * self.world

    def _loop(self, optim, loss, max_iter, progress):
        lrs = optim.spec()
        S = getattr(loss, "num_particles", 1)
        it = range(max_iter)
        bar = None
        if progress:
            try:
                from tqdm import trange
                bar = trange(max_iter)
                it = bar
            except Exception:  # pragma: no cover
                bar = None
        last = None
        sched = isinstance(optim, PyroLRScheduler)
        K = getattr(self.engine, "graph_steps", 1) if S == 1 and hasattr(self.engine, "steps") else 1
        if K > 1:
            # one particle: the iterations go to the engine K at a time (IrtEngine.steps: several steps of the loop replayed from
            # one graph where the step's form allows; the draws are made in the same order, the results are the same bits)
            i = 0
            while i < max_iter:
                n = min(K, max_iter - i)
                draws = [self._subsample() for _ in range(n)]
                last = self.engine.steps(lrs, [d[0] for d in draws], b_global=draws[0][1], scheduler=sched)[-1]
                i += n
                if bar is not None:
                    bar.update(n)
                    if (i - n) // 50 != i // 50 or i >= max_iter:
                        bar.set_postfix(loss="{0:1.2f}".format(float(last.item())), **self._postfix())
            if bar is not None:
                bar.close()
            return None if last is None else float(last.item())
        for i in it:
            if S == 1:
                rows, bg = self._subsample()
            else:
                draws = [self._subsample() for _ in range(S)]
                rows, bg = [d[0] for d in draws], draws[0][1]
            last = self.engine.step(lrs, rows=rows, b_global=bg, num_particles=S)
            if sched:
                lrs.scheduler_step()                                   # vi.py:639-640
            if bar is not None and (i % 50 == 0 or i == max_iter - 1):
                bar.set_postfix(loss="{0:1.2f}".format(float(last.item())), **self._postfix())
        return None if last is None else float(last.item())

    def _postfix(self):
        return {}


class BaseIRT(BasePsy):
    """vi.py:536-656 (constructor kwargs identical: model, x_feature, share_cov, D, a_free, a0, b0)."""

    amortized = False

    def __init__(self, model="irt_2pl", x_feature=1, share_cov=False, D=1, hidden_dim=64, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._model, self.x_feature, self.share_cov, self.D = model, int(x_feature), share_cov, D
        self.engine = IrtEngine(self.data, model=model, D=self.x_feature, Dc=float(D), amortized=self.amortized,
                                H=hidden_dim, share_cov=share_cov, a_free=self.kwargs.get("a_free"),
                                a0=self.kwargs.get("a0"), b0=self.kwargs.get("b0"),
                                encoder_init=self.kwargs.get("encoder_init"),
                                observed_lists=self.kwargs.get("observed_lists", True),
                                estimator=self.kwargs.get("estimator", "pathwise"),      # 'score': north_star's REINFORCE mode
                                baseline=self.kwargs.get("baseline", "none"),
                                baseline_beta=self.kwargs.get("baseline_beta", 0.9), **self._eng_kw)
        self._register()
        self._ri = None
        self._S, self._loo_left = 1, 0

    def fit(self, optim=None, loss=None, max_iter=5000, random_instance=None, progress=True):
        """vi.py:627-656 (defaults Adam lr 5e-2, Trace_ELBO(1), 5000 iterations).  Returns the last loss."""
        optim = optim if optim is not None else Adam({"lr": 5e-2})
        loss = loss if loss is not None else Trace_ELBO(num_particles=1)
        self._ri = random_instance
        self._S, self._loo_left = max(1, getattr(loss, "num_particles", 1)), 0
        return self._loop(optim, loss, max_iter, progress)

    def _subsample(self):
        if self.engine.estimator == "score" and self.engine.baseline == "loo":   # the particles of a step share one subsample
            if self._loo_left == 0:
                self._loo_cache, self._loo_left = super()._subsample(), self._S
            self._loo_left -= 1
            return self._loo_cache
        return super()._subsample()

    def _postfix(self):
        ri, out = self._ri, {}
        if ri is None:
            return out
        dev = self.device
        out["threshold_error"] = "{0:.4f}".format(float((param("b") - ri.b.to(dev)).abs().mean()))      # vi.py:645
        if self._model != "irt_1pl":
            d = self.x_feature
            den = d * self.item_size - d * (d - 1) / 2
            out["slop_error"] = "{0:.4f}".format(float((param("a") - ri.a.to(dev)).abs().sum() / den))  # vi.py:648
        if self._model in ("irt_3pl", "irt_4pl"):
            out["guess_error"] = "{0:.4f}".format(float((param("c") - ri.c.to(dev)).abs().mean()))
        if self._model == "irt_4pl":
            out["slip_error"] = "{0:.4f}".format(float((param("d") - ri.d.to(dev)).abs().mean()))
        return out


class VIRT(BaseIRT):
    """Black-box VI with per-person variational rows (vi.py:696-723)."""
    amortized = False


class VaeIRT(BaseIRT):
    """Amortized VI with the NormEncoder / MvnEncoder guide (vi.py:659-693)."""
    amortized = True


class _HoDinaBase(BasePsy):
    amortized = False

    def __init__(self, q, hidden_dim=64, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.q = q
        self.attr_size = int(q.shape[0])
        self.engine = HoDinaEngine(self.data, q, amortized=self.amortized, H=hidden_dim,
                                   encoder_init=self.kwargs.get("encoder_init"), **self._eng_kw)
        self._register()
        self._ri = None

    def fit(self, optim=None, loss=None, max_iter=50000, random_instance=None, progress=True):
        """vi.py:936-958 (defaults Adam lr 5e-3, TraceEnum_ELBO(1), 50000 iterations).  The reference runs
        a second evaluate_loss per iteration only to display it (vi.py:942); the step's own loss is shown."""
        optim = optim if optim is not None else Adam({"lr": 5e-3})
        loss = loss if loss is not None else TraceEnum_ELBO(num_particles=1)
        self._ri = random_instance
        return self._loop(optim, loss, max_iter, progress)

    def _postfix(self):
        ri, out = self._ri, {}
        if ri is None:
            return out
        for n in ("g", "s", "lam0", "lam1"):
            out[n] = "{0:.4f}".format(float((param(n) - getattr(ri, n).to(self.device)).abs().mean()))   # vi.py:953-956
        return out


class VCHoDina(_HoDinaBase):
    """vi.py:894-958."""
    amortized = False


class VaeCHoDina(_HoDinaBase):
    """vi.py:961-981."""
    amortized = True


class _CdmSfBase(BasePsy):
    """BaseCDM (vi.py:726-782) with a Bernoulli guide: the score-function (REINFORCE) estimator of pyro's Trace_ELBO.
    Extra keyword arguments of this build: attr_prior (None = the reference's Bernoulli(1.5) prior, vi.py:753; a float
    p = Bernoulli(p)), baseline ('none' = pyro; 'avg' = per-person decaying average; 'loo' = leave-one-out over the
    particles), baseline_beta."""

    amortized = False

    def __init__(self, q=None, model="dina", hidden_dim=64, *args, **kwargs):
        if q is None or kwargs.get("data") is None:
            raise NotImplementedError("%s needs q and data (vi.py:733-743)" % type(self).__name__)
        super().__init__(*args, **kwargs)
        self.q, self._model = q, model
        self.attr_size = int(q.shape[0])
        self.engine = CdmSfEngine(self.data, q, cdm=model, amortized=self.amortized, H=hidden_dim,
                                  encoder_init=self.kwargs.get("encoder_init"), attr_prior=self.kwargs.get("attr_prior"),
                                  baseline=self.kwargs.get("baseline", "none"),
                                  baseline_beta=self.kwargs.get("baseline_beta", 0.9), **self._eng_kw)
        self._register()
        self._ri = None

    def fit(self, optim=None, loss=None, max_iter=5000, random_instance=None, progress=True):
        """vi.py:758-782 (defaults Adam lr 1e-3, Trace_ELBO(1), 5000 iterations).  The reference evaluates the loss a
        second time per iteration only to display it (vi.py:770); the step's own loss is shown."""
        optim = optim if optim is not None else Adam({"lr": 1e-3})
        loss = loss if loss is not None else Trace_ELBO(num_particles=1)
        self._ri = random_instance
        self._S, self._loo_left = max(1, getattr(loss, "num_particles", 1)), 0
        return self._loop(optim, loss, max_iter, progress)

    def _subsample(self):
        if self.engine.baseline == "loo":                 # the particles of a step share one subsample
            if self._loo_left == 0:
                self._loo_cache, self._loo_left = super()._subsample(), self._S
            self._loo_left -= 1
            return self._loo_cache
        return super()._subsample()

    def _postfix(self):
        ri, out = self._ri, {}
        if ri is None:
            return out
        for n in ("g", "s"):
            out[n] = "{0:.4f}".format(float((param(n) - getattr(ri, n).to(self.device)).abs().mean()))   # vi.py:775-779
        return out


class VCDM(_CdmSfBase):
    """Black-box VI for DINA / DINO with per-person Bernoulli guide rows `attr_p` (vi.py:807-816)."""
    amortized = False


class VaeCDM(_CdmSfBase):
    """Amortized VI for DINA / DINO with the BinEncoder guide (vi.py:785-804, 458-470).  Positional order as in the
    reference: (hidden_dim, q, model) (vi.py:785-793)."""
    amortized = True

    def __init__(self, hidden_dim=64, q=None, model="dina", *args, **kwargs):
        super().__init__(q, model, hidden_dim, *args, **kwargs)


class VCCDM(BasePsy):
    """Pattern-enumerated DINA / DINO with the uniform pattern prior (vi.py:819-865; BaseCDM ctor vi.py:733-743:
    q, model='dina'|'dino'; fit defaults vi.py:758, 863-864: Adam lr 1e-3, TraceEnum_ELBO(1), 5000 iterations)."""

    def __init__(self, q=None, model="dina", *args, **kwargs):
        if q is None or kwargs.get("data") is None:
            raise NotImplementedError("VCCDM needs q and data (vi.py:733-743)")
        super().__init__(*args, **kwargs)
        self.q, self._model = q, model
        self.attr_size = int(q.shape[0])
        self.engine = CcdmEngine(self.data, q, cdm=model, **self._eng_kw)
        self._register()
        self._ri = None

    def fit(self, optim=None, loss=None, max_iter=5000, random_instance=None, progress=True):
        optim = optim if optim is not None else Adam({"lr": 1e-3})
        loss = loss if loss is not None else TraceEnum_ELBO(num_particles=1)
        self._ri = random_instance
        return self._loop(optim, loss, max_iter, progress)

    def _postfix(self):
        ri, out = self._ri, {}
        if ri is None:
            return out
        for n in ("g", "s"):
            out[n] = "{0:.4f}".format(float((param(n) - getattr(ri, n).to(self.device)).abs().mean()))   # vi.py:775-779
        return out


class VaeCCDM(VCCDM):
    """Pattern-enumerated DINA / DINO with the SoftmaxEncoder prior inside the model (vi.py:866-891, 473-485); as in the
    reference the encoder's softmax normalises over the persons of the batch and missing responses enter the likelihood as -1."""

    def __init__(self, hidden_dim=64, q=None, model="dina", *args, **kwargs):
        if q is None or kwargs.get("data") is None:
            raise NotImplementedError("VaeCCDM needs q and data (vi.py:733-743)")
        BasePsy.__init__(self, *args, **kwargs)
        self.q, self._model = q, model
        self.attr_size = int(q.shape[0])
        self.engine = VaeCcdmEngine(self.data, q, cdm=model, H=hidden_dim, encoder_init=self.kwargs.get("encoder_init"),
                                    **self._eng_kw)
        self._register()
        self._ri = None


def rmse_(item_size, model_name, r, x_feature):
    """The reference's error metric (test.py:70-91) -- a mean ABSOLUTE error despite the name."""
    out = {}
    if model_name in ("irt_2pl", "irt_3pl", "irt_4pl"):
        a = param("a")
        out["a"] = float((a - r.a.to(a.device)).abs().sum() / (x_feature * item_size - x_feature * (x_feature - 1) / 2))
    b = param("b")
    out["b"] = float((b - r.b.to(b.device)).abs().mean())
    if model_name in ("irt_3pl", "irt_4pl"):
        c = param("c")
        out["c"] = float((c - r.c.to(c.device)).abs().mean())
    if model_name == "irt_4pl":
        d = param("d")
        out["d"] = float((d - r.d.to(d.device)).abs().mean())
    return out


# north_star spelling of the same surface (name map)
def Irt2PL(amortized=False, **kw):
    return (VaeIRT if amortized else VIRT)(model="irt_2pl", **kw)


def Irt4PL(amortized=False, **kw):
    return (VaeIRT if amortized else VIRT)(model="irt_4pl", **kw)


def IrtMultiDim(x_feature, model="irt_2pl", **kw):
    return VaeIRT(model=model, x_feature=x_feature, **kw)


def Dina(amortized=False, enumerate=True, model="dina", **kw):
    """The DINA / DINO surface: pattern-enumerated (`VCCDM` / `VaeCCDM`, what the reference's PaDina tests run,
    test.py:558-566) or the Bernoulli-guide score-function classes (`VCDM` / `VaeCDM`, test.py:520-550)."""
    if enumerate:
        return (VaeCCDM if amortized else VCCDM)(model=model, **kw)
    return (VaeCDM if amortized else VCDM)(model=model, **kw)


def HoDina(amortized=False, **kw):
    return (VaeCHoDina if amortized else VCHoDina)(**kw)

