#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own code (/root/reference/vi.py).

Run in the build container only (``python tests/golden/make_golden.py``); the outputs are
committed, the reference and the shim never travel to the GPU box.

What runs where
  * vi.py's response functions, generators, encoders, masks, get_all_attrs  -> imported as-is
    (SURVEY.md section 8c, G1-G5);
  * vi.py's model()/guide()/SVI.step for VIRT / VaeIRT / VCHoDina / VaeCHoDina -> executed as-is
    under tests/golden/_gen/pyro_shim (a minimal effect-handler stand-in for pyro-ppl 1.4.0, which
    is not installable here), with torch autograd producing the gradients and torch.optim.Adam
    the updates.  Every reparameterisation noise draw and every subsample index draw is logged so
    the oracle and the HIP path can be fed the identical eps / idx.

A fixture is DATA: inputs (responses as uint8 with 255 = missing, initial parameters, eps, idx)
and expected outputs (loss, gradients w.r.t. the unconstrained leaves, parameters after each
step).  No reference source text is stored.
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "_gen", "pyro_shim"))
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import pyro  # noqa: E402
from pyro.poutine import runtime as rt  # noqa: E402
from pyro.optim import Adam, MultiStepLR, PyroLRScheduler  # noqa: E402
from pyro.infer import Trace_ELBO, TraceEnum_ELBO  # noqa: E402
import vi  # noqa: E402


def y_to_u8(y):
    y = y.detach().cpu().numpy()
    out = np.where(np.isnan(y), 255, np.nan_to_num(y, nan=0.0)).astype(np.uint8)
    return out


def add_missing(y, rate, gen):
    m = torch.rand(y.shape, generator=gen) < rate
    y = y.clone()
    y[m] = float("nan")
    return y


class CapOptim(object):
    """Wraps the pyro-style optimiser to record the (free-masked) grads SVI.step hands over."""

    def __init__(self, inner):
        self.inner = inner
        self.grads = None

    def __call__(self, params):
        store = pyro.get_param_store()
        self.grads = {store.param_name(p): p.grad.detach().clone().numpy() for p in params}
        self.inner(params)


def snapshot_params():
    out = {}
    for name in list(pyro.get_param_store().keys()):
        un = pyro.get_param_store().get_unconstrained(name)
        out[name] = un.detach().clone().numpy()
    return out


def run_case(tag, model_obj, data, optim, loss, steps, extra):
    rt.EPS_LOG.clear()
    rt.IDX_LOG.clear()
    rt.DISC_LOG.clear()
    cap = CapOptim(optim)
    svi = vi.SVI(model_obj.model, model_obj.guide, optim=cap, loss=loss)
    rec = dict(extra)
    rec["y"] = y_to_u8(data)
    rec["steps"] = np.int64(steps)
    if hasattr(model_obj, "encoder"):
        for k, v in model_obj.encoder.state_dict().items():
            rec["enc0/" + k] = v.detach().clone().numpy()
    for t in range(steps):
        n_eps0, n_idx0, n_disc0 = len(rt.EPS_LOG), len(rt.IDX_LOG), len(rt.DISC_LOG)
        lossv = svi.step(data)
        if isinstance(optim, PyroLRScheduler):
            optim.step()               # vi.py:639-640: scheduler advances every iteration
        eps = rt.EPS_LOG[n_eps0:]
        idx = rt.IDX_LOG[n_idx0:]
        rec["s%d/loss" % t] = np.float64(lossv)
        rec["s%d/n_particles" % t] = np.int64(len(eps))
        for k, e in enumerate(eps):
            rec["s%d/eps%d" % (t, k)] = e.numpy()
        for k, i in enumerate(idx):
            rec["s%d/idx%d" % (t, k)] = i.numpy().astype(np.int64)
        for k, a in enumerate(rt.DISC_LOG[n_disc0:]):              # draws of non-reparameterised guide sites
            rec["s%d/attr%d" % (t, k)] = a.numpy().astype(np.uint8)
        for name, g in cap.grads.items():
            rec["s%d/grad/%s" % (t, name)] = g
        for name, p in snapshot_params().items():
            rec["s%d/param/%s" % (t, name)] = p
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), **rec)
    pyro.clear_param_store()
    print("wrote", tag, "loss0=%.6f" % rec["s0/loss"])


def lr_fn(module_name, param_name):
    if param_name in ("a", "b"):
        return {"lr": 1e-2}
    return {"lr": 1e-3}


def gen_elbo_cases():
    g = torch.Generator().manual_seed(99)

    # ---- D=1 BBVI, 1..4PL (vi.py:588-595, 698-705) ------------------------------------------
    for k, (cls, name) in enumerate([(vi.RandomIrt1PL, "irt_1pl"), (vi.RandomIrt2PL, "irt_2pl"),
                                     (vi.RandomIrt3PL, "irt_3pl"), (vi.RandomIrt4PL, "irt_4pl")]):
        torch.manual_seed(100 + k)
        ri = cls(sample_size=24, item_size=7)
        y = ri.y
        if name in ("irt_2pl", "irt_4pl"):
            y = add_missing(y, 0.25, g)
        m = vi.VIRT(data=y, model=name, subsample_size=24 if name != "irt_4pl" else 10)
        run_case("virt_%s_d1" % name, m, y, Adam({"lr": 5e-2}), Trace_ELBO(num_particles=1), 3,
                 {"model": name, "cls": "VIRT", "N": 24, "J": 7, "D": 1,
                  "B": 24 if name != "irt_4pl" else 10, "lr": 5e-2})

    # ---- D=1 with D-constant 1.702 and b0 -----------------------------------------------------
    torch.manual_seed(110)
    ri = vi.RandomIrt2PL(sample_size=16, item_size=5, D=1.702)
    y = ri.y
    m = vi.VIRT(data=y, model="irt_2pl", D=1.702, b0=torch.full((1, 5), 0.3))
    run_case("virt_irt_2pl_d1_D1702", m, y, Adam({"lr": 1e-2}), Trace_ELBO(num_particles=1), 2,
             {"model": "irt_2pl", "cls": "VIRT", "N": 16, "J": 5, "D": 1, "B": 16, "lr": 1e-2,
              "Dc": 1.702, "b0": np.full((1, 5), 0.3, np.float32)})

    # ---- D=1 amortized (NormEncoder) (vi.py:417-435, 677-684) ----------------------------------
    for k, name in enumerate(["irt_2pl", "irt_4pl"]):
        torch.manual_seed(120 + k)
        cls = {"irt_2pl": vi.RandomIrt2PL, "irt_4pl": vi.RandomIrt4PL}[name]
        ri = cls(sample_size=30, item_size=9)
        y = add_missing(ri.y, 0.3, g)
        m = vi.VaeIRT(data=y, model=name, subsample_size=12, hidden_dim=8)
        run_case("vaeirt_%s_d1" % name, m, y, Adam(lr_fn), Trace_ELBO(num_particles=1), 3,
                 {"model": name, "cls": "VaeIRT", "N": 30, "J": 9, "D": 1, "B": 12, "H": 8,
                  "lr_item": 1e-2, "lr_other": 1e-3})

    # ---- D>1 amortized (MvnEncoder) (vi.py:438-455, 685-693) -----------------------------------
    for k, (cls, name, D) in enumerate([(vi.RandomMilIrt2PL, "irt_2pl", 3),
                                        (vi.RandomMilIrt3PL, "irt_3pl", 2),
                                        (vi.RandomMilIrt4PL, "irt_4pl", 4)]):
        torch.manual_seed(130 + k)
        np.random.seed(130 + k)
        import random
        random.seed(130 + k)
        ri = cls(sample_size=20, item_size=11, x_feature=D)
        y = ri.y
        if name != "irt_3pl":
            y = add_missing(y, 0.2, g)
        m = vi.VaeIRT(data=y, model=name, subsample_size=8, x_feature=D, hidden_dim=8)
        sched = MultiStepLR({"optimizer": torch.optim.Adam, "optim_args": lr_fn,
                             "milestones": [2], "gamma": 0.1})
        run_case("vaeirt_%s_d%d" % (name, D), m, y, sched, Trace_ELBO(num_particles=1), 4,
                 {"model": name, "cls": "VaeIRT", "N": 20, "J": 11, "D": D, "B": 8, "H": 8,
                  "lr_item": 1e-2, "lr_other": 1e-3, "milestones": np.array([2]), "gamma": 0.1,
                  "a_true": ri.a.numpy()})

    # ---- D>1 BBVI per-person and shared Cholesky (vi.py:706-723) -------------------------------
    for share in (False, True):
        torch.manual_seed(140 + int(share))
        ri = vi.RandomIrt2PL(sample_size=18, item_size=8, x_feature=3)
        y = add_missing(ri.y, 0.15, g)
        m = vi.VIRT(data=y, model="irt_2pl", x_feature=3, share_cov=share, subsample_size=7)
        run_case("virt_irt_2pl_d3_%s" % ("share" if share else "perperson"), m, y,
                 Adam({"lr": 2e-2}), Trace_ELBO(num_particles=1), 3,
                 {"model": "irt_2pl", "cls": "VIRT", "N": 18, "J": 8, "D": 3, "B": 7, "lr": 2e-2,
                  "share_cov": share})

    # ---- two particles (fresh subsample + eps per particle; App. A.2) ---------------------------
    torch.manual_seed(150)
    ri = vi.RandomIrt2PL(sample_size=20, item_size=6)
    y = ri.y
    m = vi.VIRT(data=y, model="irt_2pl", subsample_size=9)
    run_case("virt_irt_2pl_d1_particles2", m, y, Adam({"lr": 1e-2}), Trace_ELBO(num_particles=2), 2,
             {"model": "irt_2pl", "cls": "VIRT", "N": 20, "J": 6, "D": 1, "B": 9, "lr": 1e-2,
              "S": 2})

    # ---- HO-DINA enumerated (vi.py:897-934, 968-981) -------------------------------------------
    torch.manual_seed(160)
    ri = vi.RandomHoDina(sample_size=20, item_size=8, q_size=3)
    y = add_missing(ri.y, 0.2, g)
    m = vi.VCHoDina(data=y, q=ri.q, subsample_size=20)
    run_case("vchodina_k3", m, y, Adam({"lr": 1e-1}), TraceEnum_ELBO(num_particles=1), 3,
             {"cls": "VCHoDina", "N": 20, "J": 8, "K": 3, "B": 20, "lr": 1e-1, "q": ri.q.numpy()})
    torch.manual_seed(161)
    ri = vi.RandomHoDina(sample_size=26, item_size=10, q_size=4)
    y = ri.y
    m = vi.VCHoDina(data=y, q=ri.q, subsample_size=11)
    run_case("vchodina_k4_sub", m, y, Adam({"lr": 1e-1}), TraceEnum_ELBO(num_particles=1), 3,
             {"cls": "VCHoDina", "N": 26, "J": 10, "K": 4, "B": 11, "lr": 1e-1, "q": ri.q.numpy()})
    # Categorical(probs) clamps pattern probabilities below float32 eps (torch probs_to_logits):
    # pre-seed steep lam1 / wide theta so some of the 2^K patterns fall under the clamp.
    torch.manual_seed(163)
    ri = vi.RandomHoDina(sample_size=22, item_size=9, q_size=4)
    y = add_missing(ri.y, 0.1, g)
    pyro.param("lam1", torch.full((1, 4), 3.0), constraint=pyro.distributions.constraints.positive)
    pyro.param("theta_local", torch.linspace(-2.5, 2.5, 22).reshape(22, 1))
    m = vi.VCHoDina(data=y, q=ri.q, subsample_size=22)
    run_case("vchodina_k4_clamp", m, y, Adam({"lr": 1e-1}), TraceEnum_ELBO(num_particles=1), 2,
             {"cls": "VCHoDina", "N": 22, "J": 9, "K": 4, "B": 22, "lr": 1e-1, "q": ri.q.numpy(),
              "init/lam1": np.log(np.full((1, 4), 3.0, np.float32)),
              "init/theta_local": np.linspace(-2.5, 2.5, 22, dtype=np.float32).reshape(22, 1)})
    torch.manual_seed(162)
    ri = vi.RandomHoDina(sample_size=24, item_size=9, q_size=3)
    y = add_missing(ri.y, 0.2, g)
    m = vi.VaeCHoDina(data=y, q=ri.q, subsample_size=10, hidden_dim=8)

    def lr_ho(module_name, param_name):
        if param_name in ("lam0", "lam1", "g", "s"):
            return {"lr": 1e-1}
        return {"lr": 1e-3}
    run_case("vaechodina_k3", m, y, Adam(lr_ho), TraceEnum_ELBO(num_particles=1), 3,
             {"cls": "VaeCHoDina", "N": 24, "J": 9, "K": 3, "B": 10, "H": 8, "q": ri.q.numpy(),
              "lr_item": 1e-1, "lr_other": 1e-3})


def gen_ccdm_cases():
    """Pattern-enumerated DINA / DINO with a uniform prior over the 2^K patterns (VCCDM, vi.py:819-865)."""
    g = torch.Generator().manual_seed(177)
    torch.manual_seed(170)
    ri = vi.RandomDina(sample_size=22, item_size=9, q_size=3)
    y = add_missing(ri.y, 0.2, g)
    m = vi.VCCDM(data=y, q=ri.q, model="dina", subsample_size=22)
    run_case("vccdm_dina_k3", m, y, Adam({"lr": 1e-1}), TraceEnum_ELBO(num_particles=1), 3,
             {"cls": "VCCDM", "cdm": "dina", "N": 22, "J": 9, "K": 3, "B": 22, "lr": 1e-1, "q": ri.q.numpy()})
    torch.manual_seed(171)
    ri = vi.RandomDina(sample_size=30, item_size=12, q_size=4)
    y = ri.y
    m = vi.VCCDM(data=y, q=ri.q, model="dina", subsample_size=13)
    run_case("vccdm_dina_k4_sub", m, y, Adam({"lr": 1e-1}), TraceEnum_ELBO(num_particles=1), 3,
             {"cls": "VCCDM", "cdm": "dina", "N": 30, "J": 12, "K": 4, "B": 13, "lr": 1e-1, "q": ri.q.numpy()})
    torch.manual_seed(172)
    ri = vi.RandomDino(sample_size=24, item_size=10, q_size=3)
    y = add_missing(ri.y, 0.15, g)
    m = vi.VCCDM(data=y, q=ri.q, model="dino", subsample_size=24)
    run_case("vccdm_dino_k3", m, y, Adam({"lr": 1e-1}), TraceEnum_ELBO(num_particles=1), 3,
             {"cls": "VCCDM", "cdm": "dino", "N": 24, "J": 10, "K": 3, "B": 24, "lr": 1e-1, "q": ri.q.numpy()})


def gen_round2_cases():
    """Fixtures added in round 2: the CFA call pattern (custom a_free / a0 masks, several particles; test.py:418-430),
    an encoder of the reference's default width with more than 128 items, a width that is neither 8 nor 64, and
    HO-DINA with more than 128 items."""
    import random
    g = torch.Generator().manual_seed(277)
    # ---- CFA: VIRT(x_feature=2, a_free=mask.T, a0=mask.T), Trace_ELBO(num_particles=3) on the first rows of the
    # reference's own ex5.2.dat (float64 in the reference, test.py:419,428; responses are 0/1)
    data = np.loadtxt("/root/reference/ex5.2.dat")[:36]
    y = torch.from_numpy(data).float()
    mask = torch.FloatTensor([[1, 0], [1, 0], [1, 0], [0, 1], [0, 1], [0, 1]])
    torch.manual_seed(200)
    m = vi.VIRT(data=y, model="irt_2pl", subsample_size=12, x_feature=2, a_free=mask.T, a0=mask.T.clone())
    run_case("virt_cfa_d2_particles3", m, y, Adam({"lr": 1e-2}), Trace_ELBO(num_particles=3), 2,
             {"model": "irt_2pl", "cls": "VIRT", "N": 36, "J": 6, "D": 2, "B": 12, "lr": 1e-2, "S": 3,
              "a_free": mask.T.numpy(), "a0": mask.T.numpy()})
    # ---- VaeIRT, hidden_dim = 64 (the reference default, vi.py:661), J = 130 > 128, D = 2
    torch.manual_seed(201)
    np.random.seed(201)
    random.seed(201)
    ri = vi.RandomMilIrt2PL(sample_size=24, item_size=130, x_feature=2)
    y = add_missing(ri.y, 0.1, g)
    m = vi.VaeIRT(data=y, model="irt_2pl", subsample_size=9, x_feature=2, hidden_dim=64)
    run_case("vaeirt_irt_2pl_d2_h64_j130", m, y, Adam(lr_fn), Trace_ELBO(num_particles=1), 2,
             {"model": "irt_2pl", "cls": "VaeIRT", "N": 24, "J": 130, "D": 2, "B": 9, "H": 64,
              "lr_item": 1e-2, "lr_other": 1e-3})
    # ---- VaeIRT, hidden_dim = 24 (neither 8 nor 64), D = 3, 4PL
    torch.manual_seed(202)
    np.random.seed(202)
    random.seed(202)
    ri = vi.RandomMilIrt4PL(sample_size=20, item_size=13, x_feature=3)
    y = add_missing(ri.y, 0.15, g)
    m = vi.VaeIRT(data=y, model="irt_4pl", subsample_size=20, x_feature=3, hidden_dim=24)
    run_case("vaeirt_irt_4pl_d3_h24", m, y, Adam(lr_fn), Trace_ELBO(num_particles=1), 2,
             {"model": "irt_4pl", "cls": "VaeIRT", "N": 20, "J": 13, "D": 3, "B": 20, "H": 24,
              "lr_item": 1e-2, "lr_other": 1e-3})
    # ---- NormEncoder width 24, D = 1
    torch.manual_seed(203)
    ri = vi.RandomIrt2PL(sample_size=28, item_size=10)
    y = add_missing(ri.y, 0.2, g)
    m = vi.VaeIRT(data=y, model="irt_2pl", subsample_size=11, hidden_dim=24)
    run_case("vaeirt_irt_2pl_d1_h24", m, y, Adam(lr_fn), Trace_ELBO(num_particles=1), 2,
             {"model": "irt_2pl", "cls": "VaeIRT", "N": 28, "J": 10, "D": 1, "B": 11, "H": 24,
              "lr_item": 1e-2, "lr_other": 1e-3})
    # ---- HO-DINA with J = 130 > 128 items
    torch.manual_seed(204)
    ri = vi.RandomHoDina(sample_size=48, item_size=130, q_size=3)
    y = add_missing(ri.y, 0.1, g)
    m = vi.VCHoDina(data=y, q=ri.q, subsample_size=48)
    run_case("vchodina_k3_j130", m, y, Adam({"lr": 1e-1}), TraceEnum_ELBO(num_particles=1), 2,
             {"cls": "VCHoDina", "N": 48, "J": 130, "K": 3, "B": 48, "lr": 1e-1, "q": ri.q.numpy()})


def gen_score_function_cases():
    """Bernoulli-guide CDMs with the score-function estimator (VCDM / VaeCDM, vi.py:726-816), incl. the reference's
    Bernoulli(1.5) prior (vi.py:753) and the DINO in-place quirk; complete data (the model does not mask NaN)."""
    torch.manual_seed(300)
    ri = vi.RandomDina(sample_size=26, item_size=9, q_size=3)
    y = ri.y
    m = vi.VCDM(data=y, q=ri.q, model="dina", subsample_size=26)
    run_case("vcdm_dina_k3", m, y, Adam({"lr": 5e-2}), Trace_ELBO(num_particles=1), 3,
             {"cls": "VCDM", "cdm": "dina", "N": 26, "J": 9, "K": 3, "B": 26, "lr": 5e-2, "q": ri.q.numpy()})
    torch.manual_seed(301)
    ri = vi.RandomDino(sample_size=30, item_size=11, q_size=4)
    y = ri.y
    m = vi.VCDM(data=y, q=ri.q, model="dino", subsample_size=12)
    run_case("vcdm_dino_k4_sub_particles2", m, y, Adam({"lr": 5e-2}), Trace_ELBO(num_particles=2), 3,
             {"cls": "VCDM", "cdm": "dino", "N": 30, "J": 11, "K": 4, "B": 12, "lr": 5e-2, "S": 2, "q": ri.q.numpy()})
    torch.manual_seed(302)
    ri = vi.RandomDina(sample_size=24, item_size=10, q_size=3)
    y = ri.y
    m = vi.VaeCDM(data=y, q=ri.q, model="dina", subsample_size=10, hidden_dim=8)

    def lr_cdm(module_name, param_name):
        return {"lr": 1e-1 if param_name in ("g", "s") else 1e-2}
    run_case("vaecdm_dina_k3", m, y, Adam(lr_cdm), Trace_ELBO(num_particles=1), 3,
             {"cls": "VaeCDM", "cdm": "dina", "N": 24, "J": 10, "K": 3, "B": 10, "H": 8, "q": ri.q.numpy(),
              "lr_item": 1e-1, "lr_other": 1e-2})


def gen_vaeccdm_cases():
    """VaeCCDM (vi.py:866-891): pattern prior from the SoftmaxEncoder (softmax over the batch), missing responses as -1."""
    g = torch.Generator().manual_seed(377)
    torch.manual_seed(310)
    ri = vi.RandomDina(sample_size=22, item_size=9, q_size=3)
    y = add_missing(ri.y, 0.15, g)
    m = vi.VaeCCDM(data=y, q=ri.q, model="dina", subsample_size=10, hidden_dim=8)

    def lr_cc(module_name, param_name):
        return {"lr": 1e-1 if param_name in ("g", "s") else 1e-2}
    run_case("vaeccdm_dina_k3", m, y, Adam(lr_cc), TraceEnum_ELBO(num_particles=1), 3,
             {"cls": "VaeCCDM", "cdm": "dina", "N": 22, "J": 9, "K": 3, "B": 10, "H": 8, "q": ri.q.numpy(),
              "lr_item": 1e-1, "lr_other": 1e-2})
    torch.manual_seed(311)
    ri = vi.RandomDino(sample_size=20, item_size=8, q_size=2)
    y = ri.y
    m = vi.VaeCCDM(data=y, q=ri.q, model="dino", subsample_size=20, hidden_dim=8)
    run_case("vaeccdm_dino_k2", m, y, Adam(lr_cc), TraceEnum_ELBO(num_particles=1), 2,
             {"cls": "VaeCCDM", "cdm": "dino", "N": 20, "J": 8, "K": 2, "B": 20, "H": 8, "q": ri.q.numpy(),
              "lr_item": 1e-1, "lr_other": 1e-2})


def gen_function_cases():
    """G1-G5 of SURVEY.md section 8c: pure-torch pieces of vi.py imported and evaluated."""
    rec = {}
    torch.manual_seed(7)
    x1 = torch.randn(6, 1)
    x3 = torch.randn(6, 3)
    a1 = torch.rand(1, 5) * 2 + 0.5
    a3 = torch.rand(3, 5) * 2
    b = torch.randn(1, 5)
    c = torch.rand(1, 5) * 0.2
    d = 1 - torch.rand(1, 5) * 0.2
    rec.update(x1=x1, x3=x3, a1=a1, a3=a3, b=b, c=c, d=d)
    for Dc in (1, 1.702):
        t = "D%s" % ("1" if Dc == 1 else "1702")
        rec["irt_1pl_" + t] = vi.irt_1pl(x1, b, Dc)
        rec["irt_2pl_1d_" + t] = vi.irt_2pl(x1, a1, b, Dc)
        rec["irt_2pl_3d_" + t] = vi.irt_2pl(x3, a3, b, Dc)
        rec["irt_3pl_3d_" + t] = vi.irt_3pl(x3, a3, b, c, Dc)
        rec["irt_4pl_3d_" + t] = vi.irt_4pl(x3, a3, b, c, d, Dc)
    # DINA / DINO incl. edge cases (all-zero attribute pattern, item needing every attribute)
    q = torch.tensor([[1., 0, 1, 1, 0, 1], [0, 1, 1, 0, 0, 1], [0, 0, 0, 1, 1, 1]])
    attr = torch.tensor([[0., 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 1], [1, 1, 1]])
    gg = torch.tensor([[0.1, 0.2, 0.05, 0.3, 0.15, 0.25]])
    ss = torch.tensor([[0.2, 0.1, 0.15, 0.05, 0.3, 0.12]])
    rec.update(cdm_q=q, cdm_attr=attr, cdm_g=gg, cdm_s=ss)
    rec["dina_p"] = vi.dina(attr, q, gg, ss)
    rec["dino_p"] = vi.dino(attr, q, gg, ss)
    # G4 get_all_attrs (LSB-first)
    for K in (1, 2, 3, 4):
        obj = vi.VCCDM.__new__(vi.VCCDM)
        obj.attr_size = K
        rec["all_attrs_K%d" % K] = vi.VCCDM.get_all_attrs(obj)
    # G3 encoders
    torch.manual_seed(11)
    ne = vi.NormEncoder(7, 1, 8)
    me = vi.MvnEncoder(7, 3, 8)
    yin = torch.tensor(np.random.RandomState(3).choice([-1., 0., 1.], size=(5, 7)), dtype=torch.float32)
    rec["enc_in"] = yin
    for k, v in ne.state_dict().items():
        rec["norm_enc/" + k] = v
    for k, v in me.state_dict().items():
        rec["mvn_enc/" + k] = v
    loc, scale = ne(yin)
    rec["norm_enc_loc"], rec["norm_enc_scale"] = loc, scale
    loc, M = me(yin)
    rec["mvn_enc_loc"], rec["mvn_enc_M"] = loc, M
    rec["mvn_enc_L"] = torch.distributions.LowerCholeskyTransform()(M)
    # G5 _get_p_data masking
    torch.manual_seed(13)
    ri = vi.RandomIrt2PL(sample_size=8, item_size=5)
    y = ri.y
    y[1, 2] = float("nan")
    y[4, 0] = float("nan")
    obj = vi.VIRT(data=y, model="irt_2pl")
    idx = torch.tensor([4, 1, 6])
    p, data_ = obj._get_p_data(y, idx, {"x": ri.x[idx], "a": ri.a, "b": ri.b, "D": 1})
    rec.update(mask_y=torch.tensor(y_to_u8(y)), mask_idx=idx, mask_x=ri.x, mask_a=ri.a, mask_b=ri.b,
               mask_p=p, mask_data=data_)
    # Bernoulli log-prob conventions that the oracle restates
    probs = torch.tensor([0.0, 1e-9, 0.3, 0.999999, 1.0])
    bern = torch.distributions.Bernoulli(probs=probs, validate_args=False)
    rec["bern_probs"] = probs
    rec["bern_lp1"] = bern.log_prob(torch.ones(5))
    rec["bern_lp0"] = bern.log_prob(torch.zeros(5))
    # G2 generator shape / zero-pattern pins (values are torch-RNG-version specific)
    torch.manual_seed(17)
    import random
    random.seed(17)
    mi = vi.RandomMilIrt2PL(sample_size=12, item_size=9, x_feature=4)
    rec["mil_a"] = mi.a
    rec["mil_b"] = mi.b
    rec["mil_y"] = torch.tensor(y_to_u8(mi.y))
    r2 = vi.RandomIrt2PL(sample_size=5, item_size=6, x_feature=3)
    rec["irt2pl_a_d3"] = r2.a
    hd = vi.RandomHoDina(sample_size=6, item_size=7, q_size=3)
    rec["hodina_q"] = hd.q
    out = {k: (v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in rec.items()}
    np.savez_compressed(os.path.join(HERE, "functions.npz"), **out)
    print("wrote functions")


if __name__ == "__main__":
    torch.set_num_threads(1)
    if len(sys.argv) > 1 and sys.argv[1] == "ccdm":     # only the cases added after the first batch
        gen_ccdm_cases()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "r2":       # only the round-2 additions
        gen_round2_cases()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "vaeccdm":
        gen_vaeccdm_cases()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "sf":
        gen_score_function_cases()
        sys.exit(0)
    gen_function_cases()
    gen_elbo_cases()
    gen_ccdm_cases()
    gen_round2_cases()
