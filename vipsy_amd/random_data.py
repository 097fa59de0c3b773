"""The reference's synthetic-data generator classes (vi.py:120-412) with the same names, constructor keywords and attributes
(`.y`, `.x`, `.a`, `.b`, `.c`, `.d`, `.q`, `.g`, `.s`, `.attr`, `.theta`, `.lam0`, `.lam1`, `.name`), so that the reference's
demos read the same (test.py:248-257, 497-512):

    ri = RandomIrt2PL(sample_size=100000, item_size=100)
    VaeIRT(data=ri.y, model='irt_2pl', subsample_size=100).fit(random_instance=ri)

Item-side draws (J or D x J values) are made on the host with torch's RNG exactly as the reference makes them; the
N x J response matrix -- and the latent draws behind it -- are synthesised ON THE DEVICE by the HIP kernels of
vipsy_amd/csrc/k_synth.hip (include/vipsy_amd.h: vx_synth_irt / vx_synth_cdm), straight into uint8 (0 / 1).
Deliberate differences: `.y` is generated once and cached (the reference redraws it on every access, vi.py:158-161,
196-199, and HO-DINA even redraws the attributes); it is a uint8 device tensor, which every model class here accepts.
"""
import ctypes
import math
import random

import numpy as np
import torch

from . import _hip
from .engine import MODEL_CODE


def _device(device):
    return torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())


class RandomPsyData(object):
    """vi.py:123-131."""
    name = None

    def __init__(self, sample_size=10000, item_size=100, device=None, seed=None, gid0=0, *args, **kwargs):
        self.item_size, self.sample_size = item_size, sample_size
        self.device = _device(device)
        self.seed = int(seed) if seed is not None else int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        self.gid0 = int(gid0)
        self._y = None


class _RandomIrtBase(RandomPsyData):
    def _synth(self):
        _hip.require_gpu()
        L, dev = _hip.lib(), self.device
        n, J, D = self.sample_size, self.item_size, self.x_feature
        model = self.name
        y = torch.empty((n, J), dtype=torch.uint8, device=dev)
        x = torch.empty((n, D), dtype=torch.float32, device=dev)
        dv = {k: getattr(self, k).to(dev).float().contiguous() for k in ("a", "b", "c", "d") if hasattr(self, k)}
        x_in = None
        if getattr(self, "_x_host", None) is not None:                 # latent draws the host made (correlated traits)
            x_in = self._x_host.to(dev).float().contiguous()
        loc, sc = getattr(self, "_x_affine", (0.0, 1.0))
        if (loc, sc) != (0.0, 1.0):
            # x = loc + sc z with z ~ N(0, 1) drawn by the kernel: the response law is that of z with a' = sc a,
            # b' = b + loc sum_k a_k (1PL: a = 1)
            a_eff = dv["a"] if "a" in dv else torch.ones((D, J), dtype=torch.float32, device=dev)
            dv["b"] = (dv["b"] + loc * a_eff.sum(0, keepdim=True)).contiguous()
            if "a" in dv:
                dv["a"] = (sc * dv["a"]).contiguous()
            elif sc != 1.0:
                # the 1PL link has no slope to fold the scale into: the same response law through the 2PL link with a = sc
                model, dv["a"] = "irt_2pl", (sc * a_eff).contiguous()
        cfg = _hip.IrtCfg(MODEL_CODE[model], D, J, 0, float(self.D), 1.0, self.seed, 0, 0)
        _hip.check(L.vx_synth_irt(ctypes.byref(cfg), n, self.gid0, _hip.ptr(x_in), _hip.ptr(dv.get("a")), _hip.ptr(dv["b"]),
                                  _hip.ptr(dv.get("c")), _hip.ptr(dv.get("d")), 0.0, _hip.ptr(y), _hip.ptr(x),
                                  _hip.stream_ptr()), "vx_synth_irt")
        if (loc, sc) != (0.0, 1.0):
            x.mul_(sc).add_(loc)
        self._y, self._x = y, (x_in if x_in is not None else x)

    @property
    def y(self):
        if self._y is None:
            self._synth()
        return self._y

    @property
    def x(self):
        if self._y is None:
            self._synth()
        return self._x


class RandomIrt1PL(_RandomIrtBase):
    """vi.py:202-233: x ~ N(x_local, x_scale) (drawn on the device as N(0, 1); a location / scale is folded into the item
    parameters the kernel sees and applied to `.x`), b ~ N(b_local, b_scale)."""
    name = "irt_1pl"

    def __init__(self, x_feature=1, x_local=0, x_scale=1, b_local=0, b_scale=1, D=1, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.x_feature = x_feature
        self.b = torch.empty(1, self.item_size).normal_(b_local, b_scale)
        self.D = D
        self._x_host = None
        self._x_affine = (float(x_local), float(x_scale))


class RandomIrt2PL(RandomIrt1PL):
    """vi.py:236-263: a ~ U(a_lower, a_upper) with the identification zeros a[i, J - i:] = 0."""
    name = "irt_2pl"

    def __init__(self, a_lower=0.5, a_upper=3, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.a = torch.empty(self.x_feature, self.item_size).uniform_(a_lower, a_upper)
        for i in range(self.x_feature):
            self.a[i, self.item_size - i:] = 0


class RandomIrt3PL(RandomIrt2PL):
    """vi.py:266-284."""
    name = "irt_3pl"

    def __init__(self, c_unif_lower=0.05, c_unif_upper=0.2, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.c = torch.empty(1, self.item_size).uniform_(c_unif_lower, c_unif_upper)


class RandomIrt4PL(RandomIrt3PL):
    """vi.py:287-304."""
    name = "irt_4pl"

    def __init__(self, d_unif_lower=0.8, d_unif_upper=0.95, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.d = torch.empty(1, self.item_size).uniform_(d_unif_lower, d_unif_upper)


class RandomMilIrt2PL(_RandomIrtBase):
    """vi.py:307-380: multidimensional items by direction cosines -- a[:, j] = mdisc_j cos(omega_j), mdisc ~ LogN, the angles
    uniform on the simplex-like set sum(omega) = (n - 1) pi / 2, floor 0.01, identification zeros; b = -mdiff mdisc."""
    name = "irt_2pl"

    def __init__(self, mdisc_log_local=0, mdisc_log_scale=0.5, mdiff_local=0.5, mdiff_scale=1, x_feature=2, x_local=None,
                 x_cov=None, D=1, *args, **kwargs):
        super().__init__(*args, **kwargs)
        mdisc = torch.empty(self.item_size).log_normal_(mdisc_log_local, mdisc_log_scale)
        mdiff = torch.empty(self.item_size).normal_(mdiff_local, mdiff_scale)
        self.a = self.gen_a(self.item_size, mdisc, x_feature)
        self.b = (-mdiff * mdisc).view(1, -1)
        self.x_feature = x_feature
        self.D = D
        self._x_host = None
        if x_local is not None or x_cov is not None:                   # correlated traits: drawn on the host as the reference does
            loc = torch.zeros(x_feature) if x_local is None else torch.as_tensor(x_local, dtype=torch.float32)
            cov = torch.eye(x_feature) if x_cov is None else torch.as_tensor(x_cov, dtype=torch.float32)
            self._x_host = torch.distributions.MultivariateNormal(loc, cov).sample((self.sample_size,))

    @staticmethod
    def gen_omega(x_feature):
        """Angles in [0, pi/2] that sum to (n - 1) pi / 2 (vi.py:344-364), drawn one after the other inside the range the
        remaining ones leave open (Python's `random`, as in the reference)."""
        lo, up = [0.0] * x_feature, [math.pi / 2] * x_feature
        total, out = math.pi / 2 * (x_feature - 1), []
        while len(lo) > 1:
            t = random.uniform(max(total - sum(up[1:]), lo[0]), min(total - sum(lo[1:]), up[0]))
            out.append(t)
            lo, up, total = lo[1:], up[1:], total - t
        out.append(total)
        return out

    def gen_a(self, item_size, mdisc, x_feature):
        a = torch.zeros((x_feature, item_size))
        for j in range(item_size):
            n = x_feature if j < item_size - x_feature + 1 else item_size - j
            a[:n, j] = mdisc[j] * torch.cos(torch.tensor(self.gen_omega(n), dtype=torch.float32))
        a.clamp_(min=0.01)                                             # vi.py:372
        for i in range(x_feature):
            a[i, item_size - i:] = 0                                   # vi.py:378-379
        return a


class RandomMilIrt3PL(RandomMilIrt2PL):
    """vi.py:383-395: c = sigmoid(N(logit_c_local, logit_c_scale))."""
    name = "irt_3pl"

    def __init__(self, logit_c_local=-1.39, logit_c_scale=0.16, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.c = torch.sigmoid(torch.empty(1, self.item_size).normal_(logit_c_local, logit_c_scale))


class RandomMilIrt4PL(RandomMilIrt3PL):
    """vi.py:398-410: d = 1 / (1 + exp(logit_d)) = sigmoid(-logit_d)."""
    name = "irt_4pl"

    def __init__(self, logit_d_local=-1.39, logit_d_scale=0.16, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.d = torch.sigmoid(-torch.empty(1, self.item_size).normal_(logit_d_local, logit_d_scale))


class RandomDina(RandomPsyData):
    """vi.py:134-161: q ~ Bern(q_p) with the zero-column fix-up, attributes ~ Bern(attr_p), g, s ~ U(0, 0.3)."""
    name = "dina"
    _dino, _hodina = False, False

    def __init__(self, q_size=5, q_p=0.5, attr_p=0.5, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.q_size, self.attr_p = q_size, attr_p
        self.q = torch.empty(q_size, self.item_size).bernoulli_(q_p)
        q_sum = self.q.sum(0)
        if torch.any(q_sum == 0):
            idx = torch.randint(0, q_size, (int((q_sum == 0).sum()),))
            self.q[:, q_sum == 0] = torch.eye(q_size)[idx].T
        self.g = torch.empty(1, self.item_size).uniform_(0, 0.3)
        self.s = torch.empty(1, self.item_size).uniform_(0, 0.3)

    def _synth(self):
        _hip.require_gpu()
        L, dev = _hip.lib(), self.device
        n, J, K = self.sample_size, self.item_size, self.q_size
        cfg = _hip.HoDinaCfg(K, J, 0, 0, 1.0, 0.0, self.seed, 0, 0)
        y = torch.empty((n, J), dtype=torch.uint8, device=dev)
        attr = torch.empty((n, K), dtype=torch.uint8, device=dev)
        theta = torch.empty(n, dtype=torch.float32, device=dev) if self._hodina else None
        dv = {k: getattr(self, k).to(dev).float().contiguous() for k in ("q", "g", "s", "lam0", "lam1") if hasattr(self, k)}
        t_loc, t_sc = getattr(self, "_theta_affine", (0.0, 1.0))
        if self._hodina and (t_loc, t_sc) != (0.0, 1.0):
            # theta = loc + sc z, z ~ N(0, 1) drawn by the kernel: sigmoid(theta lam1 + lam0) = sigmoid(z (sc lam1) + (lam0 + loc lam1))
            dv["lam0"] = (dv["lam0"] + t_loc * dv["lam1"]).contiguous()
            dv["lam1"] = (t_sc * dv["lam1"]).contiguous()
        _hip.check(L.vx_synth_cdm(ctypes.byref(cfg), int(self._dino), int(self._hodina), float(self.attr_p), n, self.gid0,
                                  _hip.ptr(dv["q"]), _hip.ptr(dv["g"]), _hip.ptr(dv["s"]), _hip.ptr(dv.get("lam0")),
                                  _hip.ptr(dv.get("lam1")), 0.0, _hip.ptr(y), _hip.ptr(attr), _hip.ptr(theta),
                                  _hip.stream_ptr()), "vx_synth_cdm")
        self._y, self._attr = y, attr
        if theta is not None:
            self._theta = theta.reshape(n, 1)
            if (t_loc, t_sc) != (0.0, 1.0):
                self._theta.mul_(t_sc).add_(t_loc)

    @property
    def y(self):
        if self._y is None:
            self._synth()
        return self._y

    @property
    def attr(self):
        if self._y is None:
            self._synth()
        return self._attr


class RandomDino(RandomDina):
    """vi.py:164-172 (responses through the reference's dino(), in-place sequencing included)."""
    name = "dino"
    _dino = True


class RandomHoDina(RandomDina):
    """vi.py:175-199: theta ~ N(theta_local, theta_scale), lam0 ~ N, lam1 ~ U(.5, 3); attributes ~
    Bern(sigmoid(theta lam1 + lam0))."""
    name = "ho_dina"
    _hodina = True

    def __init__(self, theta_local=0, theta_scale=1, lam0_local=0, lam0_scale=1, lam1_lower=0.5, lam1_upper=3, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._theta_affine = (float(theta_local), float(theta_scale))
        self.lam0 = torch.empty(1, self.q_size).normal_(lam0_local, lam0_scale)
        self.lam1 = torch.empty(1, self.q_size).uniform_(lam1_lower, lam1_upper)

    @property
    def theta(self):
        if self._y is None:
            self._synth()
        return self._theta


_ = np
