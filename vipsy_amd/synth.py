"""Synthetic response data with the distributions of the reference's Random* generators
(vi.py:120-412) in uint8 (0 / 1 / 255 = missing), from explicit item parameters and a seed -- what bench.py and
the tests feed the engines.  Benchmark / test INPUT only, never inside a timed region.

Item-side draws (J or D x J values) are made on the host exactly as the reference does.  The N x J responses and the
latent draws behind them come from ONE generator per device: on a GPU the HIP kernels of vipsy_amd/csrc/k_synth.hip
(vx_synth_irt / vx_synth_cdm, Philox keyed by the global person id: a data set does not depend on the sharding); on the
CPU -- the `-m "not gpu"` tests only -- torch ops with the same distributions.
"""
import math
import random

import ctypes

import numpy as np
import torch


def _on_gpu(device):
    return torch.device(device).type == "cuda"


def _device_irt(n, gid0, items, model, device, seed, missing, Dc):
    from . import _hip
    from .engine import MODEL_CODE
    J = items["b"].shape[1]
    D = items["a"].shape[0] if "a" in items else 1
    cfg = _hip.IrtCfg(MODEL_CODE[model], D, J, 0, float(Dc), 1.0, int(seed), 0, 0)
    dv = {k: items[k].to(device).float().contiguous() for k in ("a", "b", "c", "d") if k in items}
    y = torch.empty((n, J), dtype=torch.uint8, device=device)
    with torch.cuda.device(device):
        _hip.check(_hip.lib().vx_synth_irt(ctypes.byref(cfg), n, int(gid0), None, _hip.ptr(dv.get("a")), _hip.ptr(dv["b"]),
                                           _hip.ptr(dv.get("c")), _hip.ptr(dv.get("d")), float(missing), _hip.ptr(y), None,
                                           _hip.stream_ptr()), "vx_synth_irt")
    return y


def _device_cdm(n, gid0, prm, device, seed, missing, dino, hodina, attr_p):
    from . import _hip
    K, J = prm["q"].shape
    cfg = _hip.HoDinaCfg(K, J, 0, 0, 1.0, 0.0, int(seed), 0, 0)
    dv = {k: prm[k].to(device).float().contiguous() for k in ("q", "g", "s", "lam0", "lam1") if k in prm}
    y = torch.empty((n, J), dtype=torch.uint8, device=device)
    with torch.cuda.device(device):
        _hip.check(_hip.lib().vx_synth_cdm(ctypes.byref(cfg), int(dino), int(hodina), float(attr_p), n, int(gid0), _hip.ptr(dv["q"]),
                                           _hip.ptr(dv["g"]), _hip.ptr(dv["s"]), _hip.ptr(dv.get("lam0")), _hip.ptr(dv.get("lam1")),
                                           float(missing), _hip.ptr(y), None, None, _hip.stream_ptr()), "vx_synth_cdm")
    return y


def _gen_omega(n, rnd):
    """Angles in [0, pi/2] summing to (n-1) pi/2 (vi.py:344-364): direction cosines of an item."""
    lo = [0.0] * n
    up = [math.pi / 2] * n
    total = math.pi / 2 * (n - 1)
    out = []
    while len(lo) > 1:
        a = max(total - sum(up[1:]), lo[0])
        b = min(total - sum(lo[1:]), up[0])
        t = rnd.uniform(a, b)
        out.append(t)
        lo, up, total = lo[1:], up[1:], total - t
    out.append(total)
    return out


def mirt_item_params(J, D, seed, mdisc_log_scale=0.5, mdiff_loc=0.5, mdiff_scale=1.0):
    """RandomMilIrt2PL item side (vi.py:325-329, 366-380): a (D,J) with the identification zero
    pattern a[i, J-i:] = 0 and floor 0.01; b = -mdiff * mdisc."""
    g = torch.Generator().manual_seed(seed)
    rnd = random.Random(seed)
    mdisc = torch.exp(torch.randn(J, generator=g) * mdisc_log_scale)
    mdiff = torch.randn(J, generator=g) * mdiff_scale + mdiff_loc
    a = torch.zeros(D, J)
    for j in range(J):
        n = D if j < J - D + 1 else J - j
        om = torch.tensor(_gen_omega(n, rnd), dtype=torch.float32)
        a[:n, j] = mdisc[j] * torch.cos(om)
    a.clamp_(min=0.01)                                   # vi.py:372
    for i in range(D):
        a[i, J - i:] = 0                                 # vi.py:378-379
    b = (-mdiff * mdisc).reshape(1, J)
    return a, b


def irt_item_params(J, model, seed, D=1):
    """RandomIrt1PL..4PL item side (vi.py:229, 256-258, 279, 299)."""
    g = torch.Generator().manual_seed(seed)
    out = {"b": torch.randn(1, J, generator=g)}
    if model != "irt_1pl":
        a = torch.rand(D, J, generator=g) * 2.5 + 0.5
        for i in range(D):
            a[i, J - i:] = 0
        out["a"] = a
    if model in ("irt_3pl", "irt_4pl"):
        out["c"] = torch.rand(1, J, generator=g) * 0.15 + 0.05
    if model == "irt_4pl":
        out["d"] = torch.rand(1, J, generator=g) * 0.15 + 0.8
    return out


def simulate_responses(n, gid0, items, model, device, seed, missing=0.0, chunk=65536, Dc=1.0):
    """y[i, j] ~ Bernoulli(P_ij) for persons gid0 .. gid0+n-1 with x_i ~ N(0, I) (vi.py:228, 335);
    MCAR missingness at rate `missing`.  Deterministic per (seed, global person id chunk)."""
    if _on_gpu(device):
        return _device_irt(n, gid0, items, model, device, seed, missing, Dc)
    J = items["b"].shape[1]
    D = items["a"].shape[0] if "a" in items else 1
    a = items["a"].to(device) if "a" in items else torch.ones(1, J, device=device)
    b = items["b"].to(device)
    c = items["c"].to(device) if "c" in items else None
    d = items["d"].to(device) if "d" in items else None
    y = torch.empty(n, J, dtype=torch.uint8, device=device)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        g = torch.Generator(device=device).manual_seed(seed * 1000003 + (gid0 + s))
        x = torch.randn(e - s, D, generator=g, device=device)
        p = torch.sigmoid(Dc * (x @ a + b))
        if c is not None:
            p = c + ((d if d is not None else 1.0) - c) * p
        yy = (torch.rand(e - s, J, generator=g, device=device) < p).to(torch.uint8)
        if missing > 0:
            yy[torch.rand(e - s, J, generator=g, device=device) < missing] = 255
        y[s:e] = yy
    return y


def hodina_params(J, K, seed):
    """RandomHoDina item side (vi.py:148-156, 193-194): q with the zero-column fix-up, g, s, lam0, lam1."""
    g = torch.Generator().manual_seed(seed)
    q = (torch.rand(K, J, generator=g) < 0.5).float()
    empty = q.sum(0) == 0
    if bool(empty.any()):
        idx = torch.randint(0, K, (int(empty.sum()),), generator=g)
        q[:, empty] = torch.eye(K)[idx].T
    return {"q": q, "g": torch.rand(1, J, generator=g) * 0.3, "s": torch.rand(1, J, generator=g) * 0.3,
            "lam0": torch.randn(1, K, generator=g), "lam1": torch.rand(1, K, generator=g) * 2.5 + 0.5}


def simulate_hodina(n, gid0, prm, device, seed, missing=0.0, chunk=262144):
    """theta ~ N(0,1); attr_k ~ Bern(sigmoid(theta lam1_k + lam0_k)); DINA response (vi.py:69-83, 103-116)."""
    if _on_gpu(device):
        return _device_cdm(n, gid0, prm, device, seed, missing, False, True, 0.5)
    q = prm["q"].to(device)
    K, J = q.shape
    need = (q ** 2).sum(0)
    y = torch.empty(n, J, dtype=torch.uint8, device=device)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        g = torch.Generator(device=device).manual_seed(seed * 1000003 + (gid0 + s))
        th = torch.randn(e - s, 1, generator=g, device=device)
        ap = torch.sigmoid(th @ prm["lam1"].to(device) + prm["lam0"].to(device))
        attr = (torch.rand(e - s, K, generator=g, device=device) < ap).float()
        eta = ((attr @ q) == need).float()
        p = eta * (1 - prm["s"].to(device)) + (1 - eta) * prm["g"].to(device)
        yy = (torch.rand(e - s, J, generator=g, device=device) < p).to(torch.uint8)
        if missing > 0:
            yy[torch.rand(e - s, J, generator=g, device=device) < missing] = 255
        y[s:e] = yy
    return y


def dina_params(J, K, seed, q_p=0.5):
    """RandomDina / RandomDino item side (vi.py:134-156): q ~ Bern(q_p) with the zero-column fix-up, g, s ~ U(0, .3)."""
    prm = hodina_params(J, K, seed)
    if q_p != 0.5:
        g = torch.Generator().manual_seed(seed + 1)
        q = (torch.rand(K, J, generator=g) < q_p).float()
        empty = q.sum(0) == 0
        if bool(empty.any()):
            q[:, empty] = torch.eye(K)[torch.randint(0, K, (int(empty.sum()),), generator=g)].T
        prm["q"] = q
    return {"q": prm["q"], "g": prm["g"], "s": prm["s"]}


def simulate_dina(n, gid0, prm, device, seed, cdm="dina", attr_p=0.5, missing=0.0, chunk=262144):
    """attr_k ~ Bern(attr_p); DINA / DINO response (vi.py:69-100, 158-172).  DINO is the reference's dino() INCLUDING its
    in-place sequencing (vi.py:96-100; oracle/vi_oracle.py::dino_eta): eta = 1 iff the item needs more than one attribute
    and at least one of them is mastered -- the definition the kernels, random_data.RandomDino and the fitted models use."""
    if _on_gpu(device):
        return _device_cdm(n, gid0, prm, device, seed, missing, cdm == "dino", False, attr_p)
    q = prm["q"].to(device)
    K, J = q.shape
    need = (q ** 2).sum(0)
    y = torch.empty(n, J, dtype=torch.uint8, device=device)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        g = torch.Generator(device=device).manual_seed(seed * 1000003 + (gid0 + s))
        attr = (torch.rand(e - s, K, generator=g, device=device) < attr_p).float()
        if cdm == "dino":
            eta = ((((1 - attr) @ q) < need) & (need > 1)).float()
        else:
            eta = ((attr @ q) == need).float()
        p = eta * (1 - prm["s"].to(device)) + (1 - eta) * prm["g"].to(device)
        yy = (torch.rand(e - s, J, generator=g, device=device) < p).to(torch.uint8)
        if missing > 0:
            yy[torch.rand(e - s, J, generator=g, device=device) < missing] = 255
        y[s:e] = yy
    return y


def np_u8(y):
    return y.detach().cpu().numpy().astype(np.uint8)
