"""Load tests/golden/*.npz (made by tests/golden/make_golden.py from the reference's own code) and
turn each into (spec, initial unconstrained params, optimiser settings, per-step records)."""
import glob
import os

import numpy as np

from oracle import vi_oracle as vo

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def elbo_cases():
    out = []
    for p in sorted(glob.glob(os.path.join(GOLDEN, "*.npz"))):
        tag = os.path.basename(p)[:-4]
        if tag not in ("functions", "lsat6"):
            out.append(tag)
    return out


def load(tag):
    return dict(np.load(os.path.join(GOLDEN, tag + ".npz"), allow_pickle=False))


def _lr_irt(item, other):
    def fn(module, name):
        return {"lr": item if name in ("a", "b") else other}
    return fn


def _lr_ho(item, other):
    def fn(module, name):
        return {"lr": item if name in ("lam0", "lam1", "g", "s") else other}
    return fn


def _lr_cdm(item, other):
    def fn(module, name):
        return {"lr": item if name in ("g", "s") else other}
    return fn


def build(tag, dtype=np.float32):
    f = load(tag)
    cls = str(f["cls"])
    N, J, B = int(f["N"]), int(f["J"]), int(f["B"])
    enc = {k[len("enc0/"):]: f[k] for k in f if k.startswith("enc0/")}
    amort = cls.startswith("Vae")
    if cls in ("VIRT", "VaeIRT"):
        D = int(f["D"])
        spec = {"family": "irt", "model": str(f["model"]), "D": D, "Dc": float(f["Dc"]) if "Dc" in f else 1.0,
                "N": N, "amortized": amort, "share_cov": bool(f["share_cov"]) if "share_cov" in f else False,
                "a_free": f["a_free"].astype(bool) if "a_free" in f else vo.default_a_free(D, J)}
        params = vo.init_irt_params(spec, J, dtype, encoder=enc if amort else None,
                                    b0=f["b0"] if "b0" in f else None, a0=f["a0"] if "a0" in f else None)
        lr = _lr_irt(float(f["lr_item"]), float(f["lr_other"])) if "lr_item" in f else float(f["lr"])
    elif cls in ("VCDM", "VaeCDM"):
        spec = {"family": "cdm_sf", "cdm": str(f["cdm"]), "K": int(f["K"]), "N": N, "amortized": amort, "q": f["q"],
                "attr_prior": None}
        params = vo.init_cdm_sf_params(spec, J, dtype, encoder=enc if amort else None)
        lr = _lr_cdm(float(f["lr_item"]), float(f["lr_other"])) if "lr_item" in f else float(f["lr"])
    elif cls == "VaeCCDM":
        spec = {"family": "vaeccdm", "cdm": str(f["cdm"]), "K": int(f["K"]), "N": N, "amortized": True, "q": f["q"]}
        params = vo.init_ccdm_params(spec, J, dtype, encoder=enc)
        lr = _lr_cdm(float(f["lr_item"]), float(f["lr_other"]))
    elif cls == "VCCDM":
        spec = {"family": "ccdm", "cdm": str(f["cdm"]), "K": int(f["K"]), "N": N, "amortized": False, "q": f["q"]}
        params = vo.init_ccdm_params(spec, J, dtype)
        lr = float(f["lr"])
    else:
        spec = {"family": "hodina", "K": int(f["K"]), "N": N, "amortized": amort, "q": f["q"]}
        params = vo.init_hodina_params(spec, J, dtype, encoder=enc if amort else None)
        lr = _lr_ho(float(f["lr_item"]), float(f["lr_other"])) if "lr_item" in f else float(f["lr"])
    for k in f:
        if k.startswith("init/"):
            params[k[5:]] = f[k].astype(dtype)
    opt = {"lr": lr, "milestones": tuple(int(m) for m in f["milestones"]) if "milestones" in f else (),
           "gamma": float(f["gamma"]) if "gamma" in f else 0.1}
    steps = []
    for t in range(int(f["steps"])):
        S = int(f["s%d/n_particles" % t])
        n_eps = S
        if S == 0:                                          # a model with no reparameterised site (VCCDM): one particle
            S = sum(1 for k in f if k.startswith("s%d/idx" % t))
        rec = {"loss": float(f["s%d/loss" % t]),
               "idx": [f["s%d/idx%d" % (t, k)] for k in range(S)],
               "eps": [f["s%d/eps%d" % (t, k)] if k < n_eps else
                       (f["s%d/attr%d" % (t, k)] if ("s%d/attr%d" % (t, k)) in f else None) for k in range(S)],
               "grad": {k.split("/grad/")[1]: f[k] for k in f if k.startswith("s%d/grad/" % t)},
               "param": {k.split("/param/")[1]: f[k] for k in f if k.startswith("s%d/param/" % t)}}
        steps.append(rec)
    return spec, params, opt, f["y"], steps, B


def adam_conditioned(steps, t, name, rel=1e-4):
    """Entries of parameter `name` whose Adam trajectory up to step t is well conditioned: Adam's update is
    g / (sqrt(v) + 1e-8), so an entry whose gradient is at float32-noise level (|g| below `rel` of the tensor's largest
    gradient at some step so far) moves by +-lr on the SIGN OF THE NOISE -- in the reference's own float32 run too.
    Such entries (e.g. a slip parameter of an item whose gradient terms cancel) carry no parity information."""
    ok = None
    for u in range(t + 1):
        g = np.abs(steps[u]["grad"][name])
        # a whole tensor can be noise: the fc2 bias of the SoftmaxEncoder has an exactly zero gradient (its softmax runs over
        # the batch, vi.py:478) -- so the floor also looks at the largest gradient of the step
        top = max(float(np.abs(v).max()) for v in steps[u]["grad"].values())
        floor = rel * max(float(g.max()), 1e-2 * top, 1e-30)
        m = (g == 0) | (g >= floor)                                  # exact zeros (frozen / off-batch entries) are exact
        if float(g.max()) < floor:                                   # ... unless the whole tensor is rounding noise
            m = np.zeros(g.shape, bool)
        ok = m if ok is None else (ok & m)
    return ok


def grad_scale(rec, name):
    """Scale a gradient tensor is compared on: its own largest entry, but not below 1e-3 of the step's largest gradient (a
    tensor whose true gradient is zero -- the SoftmaxEncoder's fc2 bias -- holds only rounding noise of the others)."""
    top = max(float(np.abs(v).max()) for v in rec["grad"].values())
    return max(1e-3, float(np.abs(rec["grad"][name]).max()), 1e-3 * top)
