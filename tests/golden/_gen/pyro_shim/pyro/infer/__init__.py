"""pyro.infer subset: SVI base, Trace_ELBO, TraceEnum_ELBO, config_enumerate (shim)."""
import functools
import warnings

import torch

from .. import poutine
from ..poutine.handlers import TraceMessenger, ReplayMessenger, EnumMessenger, EnumConfigMessenger
from . import util  # noqa: F401
from .util import torch_item


def config_enumerate(fn=None, default="parallel"):
    if fn is None:
        return functools.partial(config_enumerate, default=default)

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        with EnumConfigMessenger():
            return fn(*args, **kwargs)
    return wrapper


def _site_log_prob(site):
    lp = site["fn"].log_prob(site["value"])
    return lp


class Trace_ELBO(object):
    """Single-plate Trace_ELBO: elbo = sum scale*logp(model) - sum
    scale*logq(guide); reparameterised guide sites contribute -entropy_term (pathwise),
    non-reparameterised ones (log_r.detach() * score_function) with log_r kept per plate index."""

    def __init__(self, num_particles=1, retain_graph=None):
        self.num_particles = num_particles
        self.retain_graph = retain_graph

    def _get_traces(self, model, guide, args, kwargs):
        for _ in range(self.num_particles):
            with TraceMessenger() as gt:
                guide(*args, **kwargs)
            with TraceMessenger() as mt, ReplayMessenger(gt.trace), self._enum_ctx():
                model(*args, **kwargs)
            yield mt.trace, gt.trace

    def _enum_ctx(self):
        return poutine.messenger.Messenger()

    def _particle(self, model_trace, guide_trace):
        elbo = 0.0
        surrogate = 0.0
        per_index_model = None
        for site in model_trace.nodes.values():
            if site["type"] != "sample":
                continue
            lp = _site_log_prob(site) * site["scale"]
            elbo = elbo + torch_item(lp.sum())
            surrogate = surrogate + lp.sum()
            if site["cond_indep_stack"]:
                v = lp.reshape(lp.shape[0], -1).sum(-1) if lp.dim() > 1 else lp
                per_index_model = v if per_index_model is None else per_index_model + v
        for site in guide_trace.nodes.values():
            if site["type"] != "sample":
                continue
            raw = _site_log_prob(site)
            lp = raw * site["scale"]
            elbo = elbo - torch_item(lp.sum())
            if getattr(site["fn"], "has_rsample", False):
                surrogate = surrogate - lp.sum()                      # entropy term (pathwise)
            else:
                # score-function term: downstream cost per plate index, detached, UNscaled score
                v = lp.reshape(lp.shape[0], -1).sum(-1) if lp.dim() > 1 else lp
                log_r = (per_index_model - v).detach()
                score = raw.reshape(raw.shape[0], -1).sum(-1) if raw.dim() > 1 else raw
                surrogate = surrogate + (log_r * score).sum()
        return -elbo, -surrogate

    def loss_and_grads(self, model, guide, *args, **kwargs):
        loss = 0.0
        for mt, gt in self._get_traces(model, guide, args, kwargs):
            l, s = self._particle(mt, gt)
            loss += l / self.num_particles
            if getattr(s, "requires_grad", False):
                (s / self.num_particles).backward(retain_graph=self.retain_graph)
        if loss != loss:
            warnings.warn("Encountered NaN: loss")
        return loss

    def loss(self, model, guide, *args, **kwargs):
        with torch.no_grad():
            loss = 0.0
            for mt, gt in self._get_traces(model, guide, args, kwargs):
                l, _ = self._particle(mt, gt)
                loss += l / self.num_particles
        return loss


class TraceEnum_ELBO(Trace_ELBO):
    """Exact parallel enumeration of model-side discrete sites not in the guide
    (SURVEY.md App. B.5): log-space sum-product per plate element."""

    def _enum_ctx(self):
        return EnumMessenger(first_available_dim=-2)

    def _particle(self, model_trace, guide_trace):
        enum_terms = None
        plain = 0.0
        for site in model_trace.nodes.values():
            if site["type"] != "sample":
                continue
            lp = _site_log_prob(site) * site["scale"]
            if site["infer"].get("_enumerated") or lp.dim() >= 2 and self._depends_on_enum(lp, site):
                enum_terms = lp if enum_terms is None else enum_terms + lp
            else:
                plain = plain + lp.sum()
        total = plain
        if enum_terms is not None:
            # enum dim is leftmost; divide out the scale before logsumexp, re-apply after
            scale = None
            for site in model_trace.nodes.values():
                if site["type"] == "sample" and site["infer"].get("_enumerated"):
                    scale = site["scale"]
            total = total + (torch.logsumexp(enum_terms / scale, dim=0) * scale).sum()
        for site in guide_trace.nodes.values():
            if site["type"] != "sample":
                continue
            assert getattr(site["fn"], "has_rsample", False)
            total = total - (_site_log_prob(site) * site["scale"]).sum()
        return -torch_item(total), -total

    @staticmethod
    def _depends_on_enum(lp, site):
        return bool(site.get("_enum_dependent", False)) or (lp.dim() == 2 and site["is_observed"])


class SVI(object):
    def __init__(self, model, guide, optim, loss, **kwargs):
        self.model, self.guide, self.optim = model, guide, optim
        self.loss = loss.loss
        self.loss_and_grads = loss.loss_and_grads

    def evaluate_loss(self, *args, **kwargs):
        with torch.no_grad():
            return torch_item(self.loss(self.model, self.guide, *args, **kwargs))


class _Unused(object):
    def __init__(self, *a, **k):
        raise NotImplementedError("not part of the hot path")


Predictive = Importance = HMC = NUTS = MCMC = _Unused
