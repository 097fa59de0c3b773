"""The vi.py-compatible class surface (vipsy_amd/vi.py): constructor / fit signatures of the reference's
demos (test.py:265-662), parameter recovery with the reference's own error metric (test.py:70-91)."""
import numpy as np
import pytest
import torch


def test_surface_matches_reference_names():
    from vipsy_amd import vi
    for n in ("VIRT", "VaeIRT", "VCHoDina", "VaeCHoDina", "VCDM", "VaeCDM", "VCCDM", "VaeCCDM", "Adam", "MultiStepLR",
              "Trace_ELBO", "TraceEnum_ELBO", "param", "clear_param_store", "rmse_", "Irt2PL", "Irt4PL", "IrtMultiDim",
              "HoDina", "Dina", "RandomIrt1PL", "RandomIrt2PL", "RandomIrt3PL", "RandomIrt4PL", "RandomMilIrt2PL", "RandomMilIrt3PL",
              "RandomMilIrt4PL", "RandomDina", "RandomDino", "RandomHoDina"):
        assert hasattr(vi, n), n
    spec2 = vi.Adam(lambda m, n: {"lr": 1e-2, "betas": (0.8, 0.9), "eps": 1e-6} if n == "a" else {"lr": 1e-3}).spec()
    assert spec2.hyper_of("a") == ((0.8, 0.9), 1e-6) and spec2.hyper_of("b") == ((0.9, 0.999), 1e-8)
    with pytest.raises(NotImplementedError):
        vi.VaeCCDM(q=None, data=None)                   # needs q and data, like every BaseCDM (vi.py:733-743)
    spec = vi.MultiStepLR({"optimizer": torch.optim.Adam, "optim_args": lambda m, n: {"lr": 1e-2 if n == "a" else 1e-3},
                           "milestones": [2], "gamma": 0.1}).spec()
    assert spec.lr_of("a") == 1e-2 and spec.lr_of("encoder$$$fc1.weight") == 1e-3
    spec.scheduler_step(); spec.scheduler_step()
    assert spec.lr_of("a") == pytest.approx(1e-3)


def test_nan_float_input_is_converted_to_u8():
    from vipsy_amd.vi import to_u8
    y = torch.tensor([[0., 1., float("nan")], [1., float("nan"), 0.]])
    out = to_u8(y, torch.device("cpu"))
    assert out.dtype == torch.uint8 and out.tolist() == [[0, 1, 255], [1, 255, 0]]
    with pytest.raises(ValueError):
        to_u8(torch.tensor([[0.5]]), torch.device("cpu"))


@pytest.mark.gpu
def test_virt_2pl_recovers_item_parameters():
    """Irt2PLTestCase.test_bbvi (test.py:281-284), shortened: N=4000, J=20, full batch."""
    from vipsy_amd import vi, synth
    vi.clear_param_store()
    dev = torch.device("cuda:0")
    items = synth.irt_item_params(20, "irt_2pl", seed=3)
    y = synth.simulate_responses(4000, 0, items, "irt_2pl", dev, seed=4)

    class RI(object):
        a, b = items["a"], items["b"]
    m = vi.VIRT(data=y, model="irt_2pl")
    l0 = m.fit(optim=vi.Adam({"lr": 5e-2}), max_iter=1, progress=False)
    l1 = m.fit(optim=vi.Adam({"lr": 5e-2}), max_iter=600, random_instance=RI, progress=False)
    assert l1 < l0
    err = vi.rmse_(20, "irt_2pl", RI, 1)
    assert err["b"] < 0.12 and err["a"] < 0.25, err


@pytest.mark.gpu
def test_vaeirt_multidim_runs_reference_call_pattern():
    """IrtMultiDimTestCase.test_ai_100_dim_2pl call pattern (test.py:336-361) at D=4, J=40."""
    from vipsy_amd import vi, synth
    vi.clear_param_store()
    dev = torch.device("cuda:0")
    a, b = synth.mirt_item_params(40, 4, seed=5)
    y = synth.simulate_responses(3000, 0, {"a": a, "b": b}, "irt_2pl", dev, seed=6)

    def optim(_, param_name):
        return {"lr": 1e-2} if param_name in ("a", "b") else {"lr": 1e-3}
    sched = vi.MultiStepLR({"optimizer": torch.optim.Adam, "optim_args": optim, "milestones": [250], "gamma": 0.1})
    m = vi.VaeIRT(data=y, model="irt_2pl", subsample_size=500, x_feature=4)
    first = m.fit(optim=sched, max_iter=1, loss=vi.Trace_ELBO(num_particles=1), progress=False)
    last = m.fit(optim=sched, max_iter=300, loss=vi.Trace_ELBO(num_particles=1), progress=False)
    assert np.isfinite(last) and last < first
    ahat = vi.param("a")
    assert ahat.shape == (4, 40)
    assert float(ahat[1, -1]) == 0.0 and float(ahat[3, -3:].abs().sum()) == 0.0     # identification zeros stay frozen
    err = float((vi.param("b").cpu() - b).abs().mean())
    assert err < 0.6


@pytest.mark.gpu
def test_vaeirt_with_sizes_the_mfma_kernels_do_not_take():
    """The same call pattern with 37 items, 3 latent dimensions, 32 hidden units and 50 rows a step: the engine pads every one
    of them with phantoms for the MFMA kernels (engine.py: IrtEngine.__init__, _pad_batch); the caller sees its own shapes."""
    from vipsy_amd import vi, synth
    vi.clear_param_store()
    dev = torch.device("cuda:0")
    a, b = synth.mirt_item_params(37, 3, seed=5)
    y = synth.simulate_responses(3001, 0, {"a": a, "b": b}, "irt_2pl", dev, seed=6)

    def optim(_, param_name):
        return {"lr": 1e-2} if param_name in ("a", "b") else {"lr": 1e-3}
    sched = vi.MultiStepLR({"optimizer": torch.optim.Adam, "optim_args": optim, "milestones": [250], "gamma": 0.1})
    m = vi.VaeIRT(data=y, model="irt_2pl", subsample_size=50, x_feature=3, hidden_dim=32)
    eng = m.engine
    assert (eng.J, eng.D, eng.H) == (40, 4, 64) and (eng.J_items, eng.D_model, eng.H_model) == (37, 3, 32)
    first = m.fit(optim=sched, max_iter=1, loss=vi.Trace_ELBO(num_particles=1), progress=False)
    last = m.fit(optim=sched, max_iter=300, loss=vi.Trace_ELBO(num_particles=1), progress=False)
    assert np.isfinite(last) and last < first
    ahat = vi.param("a")
    assert ahat.shape == (3, 37) and vi.param("b").shape[-1] == 37
    assert float(ahat[1, -1]) == 0.0 and float(ahat[2, -2:].abs().sum()) == 0.0     # identification zeros stay frozen
    assert tuple(vi.param("encoder$$$fc1.weight").shape) == (32, 37) and tuple(vi.param("encoder$$$fc22.weight").shape) == (6, 32)
    # the phantoms have not moved: zero head rows / columns, zero a row, zero fc1 rows and columns
    from vipsy_amd.engine import _EngineBase
    w22 = _EngineBase.unconstrained(eng, "encoder$$$fc22.weight")
    assert float(w22[6:].abs().max()) == 0.0 and float(w22[:, 32:].abs().max()) == 0.0
    w1 = _EngineBase.unconstrained(eng, "encoder$$$fc1.weight")
    assert float(w1[32:].abs().max()) == 0.0 and float(w1[:, 37:].abs().max()) == 0.0
    assert float(_EngineBase.unconstrained(eng, "a")[3:].abs().max()) == 0.0
    err = float((vi.param("b").cpu() - b).abs().mean())
    assert err < 0.8


@pytest.mark.gpu
def test_vchodina_recovers_guess_and_slip():
    """PaHoDinaTestCase.test_bbvi (test.py:638-641) at N=3000, J=30, K=3."""
    from vipsy_amd import vi, synth
    vi.clear_param_store()
    dev = torch.device("cuda:0")
    prm = synth.hodina_params(30, 3, seed=8)
    y = synth.simulate_hodina(3000, 0, prm, dev, seed=9)
    m = vi.VCHoDina(data=y, q=prm["q"], subsample_size=3000)
    m.fit(optim=vi.Adam({"lr": 1e-1}), max_iter=300, progress=False)
    g_err = float((vi.param("g").cpu() - prm["g"]).abs().mean())
    s_err = float((vi.param("s").cpu() - prm["s"]).abs().mean())
    assert g_err < 0.06 and s_err < 0.08, (g_err, s_err)


@pytest.mark.gpu
def test_two_particles_and_amortized_1d():
    from vipsy_amd import vi, synth
    vi.clear_param_store()
    dev = torch.device("cuda:0")
    items = synth.irt_item_params(24, "irt_4pl", seed=10)
    y = synth.simulate_responses(2000, 0, items, "irt_4pl", dev, seed=11, missing=0.3)
    m = vi.VaeIRT(data=y, model="irt_4pl", subsample_size=200)
    l = m.fit(optim=vi.Adam({"lr": 1e-2}), loss=vi.Trace_ELBO(num_particles=2), max_iter=50, progress=False)
    assert np.isfinite(l)
    c = vi.param("c")
    assert ((c > 0) & (c < 1)).all()


@pytest.mark.gpu
def test_config1_lsat6_bbvi_lands_near_the_known_answer():
    """BASELINE config 1: VIRT('irt_2pl') on the reference's lsat.dat with the fit() defaults (vi.py:627);
    known answer = exact marginal ML, VI lands within the Gaussian-q bias."""
    import os
    from vipsy_amd import vi
    vi.clear_param_store()
    y = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lsat6.npz"))["y"]
    m = vi.VIRT(data=torch.from_numpy(y), model="irt_2pl")
    m.fit(max_iter=1500, progress=False)
    a = vi.param("a").cpu().numpy()[0]
    b = vi.param("b").cpu().numpy()[0]
    assert np.abs(a - np.array([0.8257, 0.7227, 0.8909, 0.6884, 0.6569])).max() < 0.15, a
    assert np.abs(b - np.array([2.7732, 0.9902, 0.2491, 1.2848, 2.0533])).max() < 0.15, b


@pytest.mark.gpu
def test_vccdm_dina_recovers_guess_and_slip():
    """DinaTestCase-style call pattern (VCCDM(data=, q=).fit(...), test.py): N=3000, J=20, K=3."""
    from vipsy_amd import vi
    vi.clear_param_store()
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(11)
    K, J, N = 3, 20, 3000
    q = (rng.rand(K, J) < 0.5).astype(np.float32)
    q[rng.randint(0, K, size=J), np.arange(J)] = 1.0
    g, s = rng.uniform(0.05, 0.25, J), rng.uniform(0.05, 0.25, J)
    attr = (rng.rand(N, K) < 0.5).astype(np.float32)
    eta = ((attr @ q) == (q ** 2).sum(0)).astype(np.float32)
    p = (1 - s) ** eta * g ** (1 - eta)
    y = torch.from_numpy((rng.rand(N, J) < p).astype(np.float32)).to(dev)

    class RI(object):
        pass
    RI.g, RI.s = torch.tensor(g, dtype=torch.float32).reshape(1, J), torch.tensor(s, dtype=torch.float32).reshape(1, J)
    m = vi.VCCDM(data=y, q=torch.from_numpy(q), model="dina")
    l0 = m.fit(optim=vi.Adam({"lr": 5e-2}), max_iter=1, progress=False)
    l1 = m.fit(optim=vi.Adam({"lr": 5e-2}), max_iter=400, random_instance=RI, progress=False)
    assert l1 < l0
    assert float((vi.param("g").cpu() - RI.g).abs().mean()) < 0.03
    assert float((vi.param("s").cpu() - RI.s).abs().mean()) < 0.04


@pytest.mark.gpu
def test_score_function_estimator_through_the_class_surface():
    """estimator='score' (north_star's REINFORCE mode; no reference counterpart) through the reference-style classes, for the
    1-D guide, the amortized multivariate guide and the shared-covariance guide, with each baseline: the fits run, stay finite
    and move the item parameters."""
    from vipsy_amd import vi, synth
    dev = torch.device("cuda:0")
    a, b = synth.mirt_item_params(40, 4, seed=5)
    y4 = synth.simulate_responses(2000, 0, {"a": a, "b": b}, "irt_2pl", dev, seed=6)
    items = synth.irt_item_params(24, "irt_2pl", seed=7)
    y1 = synth.simulate_responses(2000, 0, items, "irt_2pl", dev, seed=8)
    for make, loss in ((lambda: vi.VaeIRT(data=y4, model="irt_2pl", x_feature=4, subsample_size=200, estimator="score", baseline="avg"),
                        vi.Trace_ELBO(num_particles=1)),
                       (lambda: vi.VIRT(data=y4, model="irt_2pl", x_feature=4, share_cov=True, estimator="score", baseline="loo"),
                        vi.Trace_ELBO(num_particles=3)),
                       (lambda: vi.VIRT(data=y1, model="irt_2pl", subsample_size=500, estimator="score", baseline="none"),
                        vi.Trace_ELBO(num_particles=1))):
        vi.clear_param_store()
        m = make()
        b0 = vi.param("b").clone()
        l = m.fit(optim=vi.Adam({"lr": 1e-2}), loss=loss, max_iter=30, progress=False)
        assert np.isfinite(float(l))
        bb = vi.param("b")
        assert torch.isfinite(bb).all() and float((bb - b0).abs().max()) > 1e-3
