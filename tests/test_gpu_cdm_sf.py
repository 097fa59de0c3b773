"""GPU parity of the Bernoulli-guide CDMs with the score-function estimator (VCDM / VaeCDM, vi.py:726-816) through the C
ABI: golden replays of the reference's own steps, random problems against the oracle, the in-kernel Philox draws, the
control-variate baselines and the class surface."""
import numpy as np
import pytest
import torch

from oracle import vi_oracle as vo
from tests import golden_util as gu

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _engine(tag):
    from vipsy_amd.engine import CdmSfEngine, LrSpec
    spec, params, opt, y, steps, B = gu.build(tag, np.float64)
    enc = {k.split("$$$")[1]: v for k, v in params.items() if k.startswith("encoder$$$")}
    eng = CdmSfEngine(torch.from_numpy(y).to(_dev()), spec["q"], cdm=spec["cdm"], amortized=spec["amortized"],
                      H=(enc["fc1.weight"].shape[0] if enc else 0), encoder_init=enc if enc else None, seed=1)
    lrs = LrSpec(opt["lr"], milestones=opt["milestones"], gamma=opt["gamma"])
    return eng, lrs, spec, params, opt, y, steps


@pytest.mark.parametrize("tag", ["vcdm_dina_k3", "vcdm_dino_k4_sub_particles2", "vaecdm_dina_k3"])
def test_hip_replays_reference_score_function_steps(tag):
    """Same y / subsample / attribute draws as the reference run -> same loss, gradients and Adam trajectory."""
    eng, lrs, spec, params, opt, y, steps = _engine(tag)
    adam = vo.Adam(opt["lr"], milestones=opt["milestones"], gamma=opt["gamma"])
    for t, rec in enumerate(steps):
        S = len(rec["idx"])
        rows = [torch.from_numpy(i).to(_dev()) for i in rec["idx"]]
        full = [len(i) == spec["N"] and (i == np.arange(spec["N"])).all() for i in rec["idx"]]
        rows = [None if f else r for f, r in zip(full, rows)]
        attrs = [torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint8)).to(_dev()) for a in rec["eps"]]
        loss_h = float(eng.step(lrs, rows=rows if S > 1 else rows[0], b_global=len(rec["idx"][0]),
                                eps=attrs if S > 1 else attrs[0], num_particles=S).item())
        lrs.scheduler_step()
        torch.cuda.synchronize()
        loss_o, g_o = vo.loss_and_grads(spec, params, y, rec["idx"], rec["eps"])
        assert loss_h == pytest.approx(loss_o, rel=2e-5), (tag, t)
        assert loss_h == pytest.approx(rec["loss"], rel=2e-4), (tag, t)
        for name, go in g_o.items():
            gh = eng.unconstrained(name, eng.GP if (eng.per_person and name in eng.pp_off) else eng.G).cpu().numpy()
            sc = max(1e-3, float(np.abs(go).max()))
            np.testing.assert_allclose(gh / sc, go / sc, atol=3e-5, err_msg="%s step %d grad %s" % (tag, t, name))
            np.testing.assert_allclose(gh / sc, rec["grad"][name] / sc, atol=1e-3, err_msg="golden %s %s" % (tag, name))
        adam.step(params, g_o)
        adam.scheduler_step()
        for name, p in rec["param"].items():
            ok = gu.adam_conditioned(steps, t, name)
            ph = eng.unconstrained(name).cpu().numpy()
            np.testing.assert_allclose(ph[ok], params[name][ok], atol=2e-5, rtol=1e-4, err_msg="%s step %d param %s" % (tag, t, name))
            np.testing.assert_allclose(ph[ok], p[ok], atol=2e-4, rtol=1e-3, err_msg="golden %s step %d %s" % (tag, t, name))
            if not ok.all():
                params[name][~ok] = p[~ok]
                ph[~ok] = p[~ok]
                eng.unconstrained(name).copy_(torch.from_numpy(ph).to(eng.dev))


@pytest.mark.parametrize("N,J,K,cdm,B,amort,H", [
    (1000, 30, 8, "dina", None, False, 0),           # config-5-like shape (1M x 30 x 8 scaled down)
    (333, 100, 5, "dino", 77, False, 0),
    (257, 65, 10, "dina", None, True, 64),
    (130, 9, 1, "dino", 50, True, 24),
    (700, 200, 4, "dina", None, True, 8),
    (301, 40, 6, "dina", None, True, 96),            # hidden_dim > 64
    (150, 33, 3, "dino", 60, True, 128),
])
def test_cdm_sf_step_vs_oracle(N, J, K, cdm, B, amort, H):
    """Random problems; the attribute draws are made IN the kernel (Philox keyed by the global person id) and must equal
    the oracle's restatement of the draw rule."""
    from vipsy_amd.engine import CdmSfEngine, BIN_ENC_KEYS
    rng = np.random.RandomState(N + J + K)
    q = (rng.rand(K, J) < 0.4).astype(np.float32)
    q[rng.randint(0, K, size=J), np.arange(J)] = 1.0
    y = rng.randint(0, 2, size=(N, J)).astype(np.uint8)
    eng = CdmSfEngine(torch.from_numpy(y).to(_dev()), q, cdm=cdm, amortized=amort, H=H, seed=9, gid0=1 << 33,
                      n_global=N, attr_prior=0.4)
    eng.unconstrained("g").copy_(torch.from_numpy(vo.logit(0.05 + 0.3 * rng.rand(1, J))).float())
    eng.unconstrained("s").copy_(torch.from_numpy(vo.logit(0.05 + 0.3 * rng.rand(1, J))).float())
    if not amort:
        eng.unconstrained("attr_p").copy_(torch.from_numpy(rng.randn(N, K)).float())
    idx = np.arange(N) if B is None else np.sort(rng.permutation(N)[:B])
    rows = None if B is None else torch.from_numpy(idx).to(_dev())
    eng.t = 3
    eng.loss_and_grads(rows, len(idx), None, 2)
    torch.cuda.synchronize()
    spec = {"family": "cdm_sf", "cdm": cdm, "K": K, "N": N, "amortized": amort, "q": q.astype(np.float64), "attr_prior": 0.4}
    params = {n: eng.unconstrained(n).cpu().numpy().astype(np.float64) for n in eng.all_names()}
    # the kernel's draw rule: u_ik < p_ik with p in float32
    if amort:
        W = {k: params["encoder$$$" + k] for k in BIN_ENC_KEYS}
        u, _ = vo.bin_enc_forward(W, y[idx].astype(np.float64))
        p = vo.sigmoid(u)
    else:
        p = np.clip(vo.sigmoid(params["attr_p"][idx]), vo.TINY32, 1 - vo.EPS32)
    uni = vo.philox_uniforms(9, 3, 2, (1 << 33) + idx, K)
    attr = (uni < p.astype(np.float32)).astype(np.float64)
    near = np.abs(uni - p) < 1e-6                        # a draw that sits on the threshold may round either way
    assert near.mean() < 0.01
    loss_o, g_o, lr_o = vo.cdm_sf_particle(spec, params, y, idx, attr)
    lr_h = eng.last["log_r"][:len(idx)].cpu().numpy()
    bad = near.any(axis=1)
    np.testing.assert_allclose(lr_h[~bad], lr_o[~bad], rtol=3e-5, atol=3e-4)
    if not bad.any():
        assert float(eng.G[eng.n_params].item()) == pytest.approx(loss_o, rel=3e-5)
        for name, go in g_o.items():
            gh = eng.unconstrained(name, eng.GP if (eng.per_person and name in eng.pp_off) else eng.G).cpu().numpy()
            sc = max(1e-6, float(np.abs(go).max()))
            assert np.abs(gh - go).max() / sc < 2e-4, name


def test_baselines_on_hip_match_oracle():
    """'avg': the per-person decaying average is used before it is updated; 'loo': leave-one-out mean over the particles."""
    from vipsy_amd.engine import CdmSfEngine, LrSpec
    rng = np.random.RandomState(3)
    N, J, K, S = 200, 20, 3, 3
    q = (rng.rand(K, J) < 0.5).astype(np.float32)
    q[0, q.sum(0) == 0] = 1
    y = rng.randint(0, 2, size=(N, J)).astype(np.uint8)
    spec = {"family": "cdm_sf", "cdm": "dina", "K": K, "N": N, "amortized": False, "q": q.astype(np.float64), "attr_prior": None}
    idx = np.arange(N)
    attrs = [rng.randint(0, 2, size=(N, K)).astype(np.uint8) for _ in range(S)]
    dev_attrs = [torch.from_numpy(a).to(_dev()) for a in attrs]
    # ---- avg
    eng = CdmSfEngine(torch.from_numpy(y).to(_dev()), q, baseline="avg", baseline_beta=0.8, seed=2)
    eng.unconstrained("attr_p").copy_(torch.from_numpy(0.5 * rng.randn(N, K)).float())
    params = {n: eng.unconstrained(n).cpu().numpy().astype(np.float64) for n in eng.all_names()}
    base = np.zeros(N)
    for s in range(2):
        eng.loss_and_grads(None, N, dev_attrs[s], 0)
        torch.cuda.synchronize()
        _, g_o, lr_o = vo.cdm_sf_particle(spec, params, y, idx, attrs[s], baseline=base)
        gh = eng.GP.reshape(N, K).cpu().numpy()
        np.testing.assert_allclose(gh, g_o["attr_p"], rtol=2e-4, atol=2e-4 * np.abs(g_o["attr_p"]).max())
        base = 0.8 * base + 0.2 * lr_o
        np.testing.assert_allclose(eng.base[:N].cpu().numpy(), base, rtol=1e-4, atol=1e-3)
    # ---- loo through engine.step (lr = 0 keeps the parameters so that the averaged gradient can be checked)
    eng = CdmSfEngine(torch.from_numpy(y).to(_dev()), q, baseline="loo", seed=2)
    eng.unconstrained("attr_p").copy_(torch.from_numpy(params["attr_p"]).float())
    eng.step(LrSpec(0.0), rows=None, b_global=N, eps=dev_attrs, num_particles=S)
    torch.cuda.synchronize()
    lrs_o = [vo.cdm_sf_particle(spec, params, y, idx, a)[2] for a in attrs]
    g_ref = np.mean([vo.cdm_sf_particle(spec, params, y, idx, a, baseline=(sum(lrs_o) - lrs_o[s]) / (S - 1))[1]["attr_p"]
                     for s, a in enumerate(attrs)], axis=0)
    gh = eng.GP.reshape(N, K).cpu().numpy()
    np.testing.assert_allclose(gh, g_ref, rtol=3e-4, atol=3e-4 * np.abs(g_ref).max())


def test_vcdm_class_surface_recovers_guess_and_slip():
    """VCDM / VaeCDM with the reference's call pattern (test.py:520-550); with a sane prior (attr_prior=0.5) and the
    leave-one-out baseline the guess / slip parameters move towards the truth."""
    from vipsy_amd import vi, synth
    dev = _dev()
    prm = synth.dina_params(20, 3, seed=11)
    y = synth.simulate_dina(4000, 0, prm, dev, seed=12)
    vi.clear_param_store()
    m = vi.VCDM(data=y, q=prm["q"], model="dina", subsample_size=2000, attr_prior=0.5, baseline="loo")
    err0 = float((torch.sigmoid(m.engine.unconstrained("g")) - prm["g"].to(dev)).abs().mean())
    m.fit(optim=vi.Adam({"lr": 5e-2}), loss=vi.Trace_ELBO(num_particles=4), max_iter=300, progress=False)
    err1 = float((vi.param("g") - prm["g"].to(dev)).abs().mean())
    assert err1 < err0 and err1 < 0.12
    vi.clear_param_store()
    with pytest.raises(ValueError):
        yy = y.clone()
        yy[0, 0] = 255
        vi.VaeCDM(data=yy, q=prm["q"], model="dina")
    m2 = vi.VaeCDM(data=y, q=prm["q"], model="dino", subsample_size=500, hidden_dim=32)
    assert np.isfinite(m2.fit(max_iter=5, progress=False))
    vi.clear_param_store()


# ---- VaeCCDM (vi.py:866-891): SoftmaxEncoder prior over the patterns, softmax over the batch, -1 for missing -------------
def _vaeccdm_engine(tag):
    from vipsy_amd.engine import VaeCcdmEngine, LrSpec
    spec, params, opt, y, steps, B = gu.build(tag, np.float64)
    enc = {k.split("$$$")[1]: v for k, v in params.items() if k.startswith("encoder$$$")}
    eng = VaeCcdmEngine(torch.from_numpy(y).to(_dev()), spec["q"], cdm=spec["cdm"], H=enc["fc1.weight"].shape[0],
                        encoder_init=enc, seed=1)
    return eng, LrSpec(opt["lr"], milestones=opt["milestones"], gamma=opt["gamma"]), spec, params, opt, y, steps


@pytest.mark.parametrize("tag", ["vaeccdm_dina_k3", "vaeccdm_dino_k2"])
def test_hip_replays_reference_vaeccdm_steps(tag):
    eng, lrs, spec, params, opt, y, steps = _vaeccdm_engine(tag)
    adam = vo.Adam(opt["lr"], milestones=opt["milestones"], gamma=opt["gamma"])
    for t, rec in enumerate(steps):
        idx = rec["idx"][0]
        full = len(idx) == spec["N"] and (idx == np.arange(spec["N"])).all()
        eng.loss_and_grads(None if full else torch.from_numpy(idx).to(_dev()), len(idx))
        torch.cuda.synchronize()
        loss_o, g_o = vo.loss_and_grads(spec, params, y, [idx], [None])
        loss_h = float(eng.G[eng.n_params].item())
        assert loss_h == pytest.approx(loss_o, rel=2e-5), (tag, t)
        assert loss_h == pytest.approx(rec["loss"], rel=2e-4), (tag, t)
        for name, go in g_o.items():
            gh = eng.unconstrained(name, eng.G).cpu().numpy()
            sc = gu.grad_scale(rec, name)
            if name == "encoder$$$fc2.bias":            # exactly zero in exact arithmetic (the softmax over the batch is shift
                sc = max(sc, float(np.abs(g_o["encoder$$$fc2.weight"]).max()))   # invariant): float32 leaves the rounding of its terms
            np.testing.assert_allclose(gh / sc, go / sc, atol=3e-5, err_msg="%s step %d grad %s" % (tag, t, name))
            np.testing.assert_allclose(gh / sc, rec["grad"][name] / sc, atol=1e-3, err_msg="golden %s %s" % (tag, name))
        eng.allreduce()
        eng.apply_optim(lrs)
        lrs.scheduler_step()
        adam.step(params, g_o)
        adam.scheduler_step()
        torch.cuda.synchronize()
        for name, p in rec["param"].items():
            ok = gu.adam_conditioned(steps, t, name)
            ph = eng.unconstrained(name).cpu().numpy()
            np.testing.assert_allclose(ph[ok], params[name][ok], atol=2e-5, rtol=1e-4, err_msg="%s step %d param %s" % (tag, t, name))
            np.testing.assert_allclose(ph[ok], p[ok], atol=2e-4, rtol=1e-3, err_msg="golden %s step %d %s" % (tag, t, name))
            if not ok.all():
                params[name][~ok] = p[~ok]
                ph[~ok] = p[~ok]
                eng.unconstrained(name).copy_(torch.from_numpy(ph).to(eng.dev))


@pytest.mark.parametrize("N,J,K,cdm,miss,B,H", [(600, 30, 8, "dina", 0.1, None, 64), (257, 70, 5, "dino", 0.0, 100, 24),
                                                (130, 12, 2, "dina", 0.3, None, 8), (300, 140, 9, "dina", 0.05, 77, 32),
                                                (203, 25, 4, "dina", 0.1, None, 96), (150, 30, 7, "dino", 0.0, 64, 128)])
def test_vaeccdm_step_vs_oracle(N, J, K, cdm, miss, B, H):
    from vipsy_amd.engine import VaeCcdmEngine
    rng = np.random.RandomState(N + J + K)
    q = (rng.rand(K, J) < 0.4).astype(np.float32)
    q[rng.randint(0, K, size=J), np.arange(J)] = 1.0
    y = rng.randint(0, 2, size=(N, J)).astype(np.uint8)
    y[rng.rand(N, J) < miss] = 255
    eng = VaeCcdmEngine(torch.from_numpy(y).to(_dev()), q, cdm=cdm, H=H, seed=4)
    eng.unconstrained("g").copy_(torch.from_numpy(vo.logit(0.05 + 0.3 * rng.rand(1, J))).float())
    eng.unconstrained("s").copy_(torch.from_numpy(vo.logit(0.05 + 0.3 * rng.rand(1, J))).float())
    eng.unconstrained("encoder$$$fc2.weight").mul_(3.0)             # spread the pattern priors
    idx = np.arange(N) if B is None else np.sort(rng.permutation(N)[:B])
    eng.loss_and_grads(None if B is None else torch.from_numpy(idx).to(_dev()), len(idx))
    torch.cuda.synchronize()
    spec = {"family": "vaeccdm", "cdm": cdm, "K": K, "N": N, "amortized": True, "q": q.astype(np.float64)}
    params = {n: eng.unconstrained(n).cpu().numpy().astype(np.float64) for n in eng.names()}
    loss_o, g_o = vo.loss_and_grads(spec, params, y, [idx], [None])
    assert float(eng.G[eng.n_params].item()) == pytest.approx(loss_o, rel=3e-5)
    top = max(float(np.abs(v).max()) for v in g_o.values())
    for name, go in g_o.items():
        gh = eng.unconstrained(name, eng.G).cpu().numpy()
        sc = max(1e-6, float(np.abs(go).max()), 1e-3 * top)
        if name == "encoder$$$fc2.bias":
            sc = max(sc, float(np.abs(g_o["encoder$$$fc2.weight"]).max()))
        assert np.abs(gh - go).max() / sc < 3e-4, (name, np.abs(gh - go).max() / sc)


def test_vaeccdm_class_surface():
    from vipsy_amd import vi, synth
    dev = _dev()
    prm = synth.dina_params(16, 3, seed=21)
    y = synth.simulate_dina(3000, 0, prm, dev, seed=22, missing=0.05)
    vi.clear_param_store()
    m = vi.VaeCCDM(data=y, q=prm["q"], model="dina", subsample_size=500, hidden_dim=32)
    err0 = float((torch.sigmoid(m.engine.unconstrained("g")) - prm["g"].to(dev)).abs().mean())
    m.fit(optim=vi.Adam({"lr": 2e-2}), max_iter=300, progress=False)
    err1 = float((vi.param("g") - prm["g"].to(dev)).abs().mean())
    assert np.isfinite(err1) and err1 < err0
    vi.clear_param_store()
