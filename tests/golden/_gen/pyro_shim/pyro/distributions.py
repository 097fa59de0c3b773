"""pyro.distributions subset: thin torch.distributions wrappers with to_event / score parts.
Noise for reparameterised draws is generated explicitly and logged so fixtures can store it."""
import torch
import torch.distributions as td
from torch.distributions import constraints as _c


class _Constraints(object):
    real = _c.real
    positive = _c.positive
    unit_interval = _c.unit_interval
    lower_cholesky = _c.lower_cholesky

    @staticmethod
    def interval(lo, hi):
        return _c.interval(lo, hi)


constraints = _Constraints()


def _log_eps(e):
    from .poutine import runtime
    runtime.EPS_LOG.append(e.detach().clone())
    return e


class _Mixin(object):
    def to_event(self, n=None):
        if not n:
            return self
        return Independent(self, n)


class Normal(td.Normal, _Mixin):
    def __init__(self, loc, scale):
        super().__init__(loc, scale, validate_args=False)

    def rsample(self, sample_shape=torch.Size()):
        eps = _log_eps(torch.randn(self.loc.shape))
        return self.loc + eps * self.scale


class MultivariateNormal(td.MultivariateNormal, _Mixin):
    def __init__(self, loc, covariance_matrix=None, scale_tril=None):
        super().__init__(loc, covariance_matrix=covariance_matrix, scale_tril=scale_tril,
                         validate_args=False)

    def rsample(self, sample_shape=torch.Size()):
        eps = _log_eps(torch.randn(sample_shape + self.loc.shape))
        return self.loc + torch.matmul(self.scale_tril, eps.unsqueeze(-1)).squeeze(-1)


class Bernoulli(td.Bernoulli, _Mixin):
    def __init__(self, probs=None, logits=None):
        super().__init__(probs=probs, logits=logits, validate_args=False)


class Categorical(td.Categorical, _Mixin):
    def __init__(self, probs=None, logits=None):
        super().__init__(probs=probs, logits=logits, validate_args=False)


class Independent(td.Independent, _Mixin):
    def __init__(self, base, n):
        super().__init__(base, n, validate_args=False)

    @property
    def has_rsample(self):
        return self.base_dist.has_rsample

    def rsample(self, sample_shape=torch.Size()):
        return self.base_dist.rsample(sample_shape)
