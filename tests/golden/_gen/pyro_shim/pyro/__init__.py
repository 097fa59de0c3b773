"""Minimal effect-handler shim standing in for pyro-ppl 1.4.0 -- FIXTURE GENERATION ONLY.

pyro-ppl is pinned by the reference (requirements.txt:1) but is not installed in the build
container and cannot be installed (no network).  This package implements just enough of the
pyro surface that ``/root/reference/vi.py`` imports (vi.py:8-14) for the reference's OWN
``model()`` / ``guide()`` / ``SVI.step`` code to execute, so that tests/golden/make_golden.py can
record losses, gradients and Adam trajectories produced by the reference's probabilistic programs.

It is our own code (no reference source is copied), it never travels in the product path and it
is only ever put on sys.path by make_golden.py.  Semantics follow the published pyro 1.4.0
algorithm as summarised in SURVEY.md App. B (plate scaling, replay of the guide trace in the
model, Trace_ELBO surrogate with pathwise / score-function terms, parallel enumeration for
TraceEnum_ELBO, one torch.optim.Adam per parameter tensor).  Anything it gets wrong about pyro
itself would be shared by the fixtures -- hence "parity unpinned w.r.t. pyro internals" in
oracle/README.md.
"""
import weakref

import torch
from torch.distributions import transform_to

from . import distributions  # noqa: F401
from .distributions import constraints
from . import poutine  # noqa: F401
from .poutine import runtime as _rt
from . import infer  # noqa: F401
from . import optim  # noqa: F401

_PARAMS = {}        # name -> unconstrained leaf
_CONSTRAINTS = {}   # name -> constraint
_NAME_OF = {}       # id(leaf) -> name


class _ParamStore(object):
    def param_name(self, p):
        return _NAME_OF.get(id(p))

    def keys(self):
        return _PARAMS.keys()

    def get_unconstrained(self, name):
        return _PARAMS[name]


def get_param_store():
    return _ParamStore()


def clear_param_store():
    _PARAMS.clear()
    _CONSTRAINTS.clear()
    _NAME_OF.clear()


def param(name, init_tensor=None, constraint=constraints.real, event_dim=None):
    if name not in _PARAMS:
        if init_tensor is None:
            raise KeyError(name)
        if callable(init_tensor) and not isinstance(init_tensor, torch.Tensor):
            init_tensor = init_tensor()
        with torch.no_grad():
            un = transform_to(constraint).inv(init_tensor).detach().clone().contiguous()
        un.requires_grad_(True)
        _PARAMS[name] = un
        _CONSTRAINTS[name] = constraint
        _NAME_OF[id(un)] = name
    un = _PARAMS[name]
    value = transform_to(_CONSTRAINTS[name])(un)
    value.unconstrained = weakref.ref(un)
    msg = {"type": "param", "name": name, "value": value, "free": None, "scale": 1.0,
           "cond_indep_stack": (), "infer": {}, "is_observed": False, "fn": None}
    _rt.apply_stack(msg)
    return msg["value"]


def module(name, nn_module, update_module_params=False):
    for pname, p in nn_module.named_parameters():
        full = "{}$$${}".format(name, pname)
        if full not in _PARAMS:
            _PARAMS[full] = p
            _CONSTRAINTS[full] = constraints.real
            _NAME_OF[id(p)] = full
        param(full)
    return nn_module


def sample(name, fn, obs=None, infer=None):
    msg = {"type": "sample", "name": name, "fn": fn, "value": obs, "is_observed": obs is not None,
           "scale": 1.0, "cond_indep_stack": (), "infer": dict(infer or {}), "done": False,
           "free": None}
    _rt.apply_stack(msg)
    return msg["value"]


class plate(poutine.messenger.Messenger):
    """Vectorised plate: subsamples in the guide, reuses the guide's indices when the model is
    replayed, scales log-probs inside by size / subsample_size."""

    def __init__(self, name, size, subsample_size=None, dim=None):
        super().__init__()
        self.name, self.size, self.subsample_size, self.dim = name, size, subsample_size, dim
        self.indices = None

    def __enter__(self):
        msg = {"type": "subsample", "name": self.name, "size": self.size,
               "subsample_size": self.subsample_size, "value": None, "fn": None,
               "is_observed": False, "scale": 1.0, "cond_indep_stack": (), "infer": {},
               "done": False, "free": None}
        _rt.apply_stack(msg)
        self.indices = msg["value"]
        self._scale = float(self.size) / float(len(self.indices))
        super().__enter__()
        return self.indices

    def _process_message(self, msg):
        if msg["type"] == "sample":
            msg["scale"] = msg["scale"] * self._scale
            msg["cond_indep_stack"] = msg["cond_indep_stack"] + (self.name,)
        return None
