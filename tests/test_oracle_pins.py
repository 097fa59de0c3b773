"""Independent pins of the oracle: the LSAT-6 known answer (exact marginal ML by
Gauss-Hermite quadrature) and ELBO <= log-marginal."""
import os

import numpy as np
import pytest

from oracle import vi_oracle as vo

Y = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lsat6.npz"))["y"]   # reference lsat.dat
KAT_A = np.array([0.8257, 0.7227, 0.8909, 0.6884, 0.6569])
KAT_B = np.array([2.7732, 0.9902, 0.2491, 1.2848, 2.0533])
KAT_LL = -2466.6534


def log_marginal(a, b, y, n=61):
    """sum_i log int prod_j Bern(y_ij; sigma(a_j x + b_j)) N(x;0,1) dx by Gauss-Hermite quadrature."""
    t, w = np.polynomial.hermite.hermgauss(n)
    x = np.sqrt(2.0) * t
    z = x[:, None] * a[None, :] + b[None, :]                       # (n, J)
    lp1, lp0 = -np.logaddexp(0, -z), -np.logaddexp(0, z)
    ll = y[:, None, :] * lp1[None] + (1 - y[:, None, :]) * lp0[None]
    s = ll.sum(-1) + np.log(w / np.sqrt(np.pi))[None, :]
    m = s.max(1, keepdims=True)
    return float((m[:, 0] + np.log(np.exp(s - m).sum(1))).sum())


def test_quadrature_reproduces_the_known_answer():
    assert Y.shape == (1000, 5)
    assert log_marginal(KAT_A, KAT_B, Y.astype(np.float64)) == pytest.approx(KAT_LL, abs=2e-3)


def test_oracle_bbvi_on_lsat6_lands_near_the_known_answer():
    """BASELINE config 1: VIRT('irt_2pl') on lsat.dat, fit defaults (Adam lr 5e-2, vi.py:627)."""
    N, J = Y.shape
    spec = {"family": "irt", "model": "irt_2pl", "D": 1, "Dc": 1.0, "N": N, "amortized": False, "share_cov": False,
            "a_free": None}
    params = vo.init_irt_params(spec, J, np.float64)
    adam = vo.Adam(5e-2)
    idx = np.arange(N)
    losses = []
    for t in range(1500):
        eps = vo.philox_normals(2024, t, 0, idx, 1).astype(np.float64)
        loss, g = vo.loss_and_grads(spec, params, Y, [idx], [eps])
        adam.step(params, g)
        losses.append(loss)
    a, b = params["a"][0], params["b"][0]
    # VI sits near, not on, the ML solution (Gaussian-q bias); measured gap here is < 0.1
    assert np.abs(a - KAT_A).max() < 0.15, a
    assert np.abs(b - KAT_B).max() < 0.15, b
    elbo = -np.mean(losses[-300:])
    assert elbo <= log_marginal(a, b, Y.astype(np.float64)) + 1.0          # ELBO <= log-marginal (MC noise margin)
    # measured: ELBO ~ -2500 vs log-marginal -2466.65, i.e. a Gaussian-q gap of ~0.033 nats per person
    assert elbo <= KAT_LL + 1.0 and elbo > KAT_LL - 60.0


def test_elbo_gradient_is_unbiased_direction_check():
    """Analytic pathwise gradients against central finite differences of the oracle's own loss (float64)."""
    rng = np.random.RandomState(0)
    N, J = 12, 6
    y = rng.randint(0, 2, size=(N, J)).astype(np.uint8)
    y[0, 1] = 255
    spec = {"family": "irt", "model": "irt_4pl", "D": 1, "Dc": 1.702, "N": N, "amortized": False, "share_cov": False,
            "a_free": None}
    params = vo.init_irt_params(spec, J, np.float64)
    for k in params:
        params[k] = params[k] + 0.3 * rng.randn(*params[k].shape)
    idx = np.arange(N)
    eps = rng.randn(N, 1)
    _, g = vo.loss_and_grads(spec, params, y, [idx], [eps])
    for name in ("a", "b", "c", "d", "x_local", "x_scale"):
        flat = params[name].reshape(-1)
        for pos in (0, flat.size // 2, flat.size - 1):
            old = flat[pos]
            flat[pos] = old + 1e-6
            lp, _ = vo.loss_and_grads(spec, params, y, [idx], [eps])
            flat[pos] = old - 1e-6
            lm, _ = vo.loss_and_grads(spec, params, y, [idx], [eps])
            flat[pos] = old
            assert g[name].reshape(-1)[pos] == pytest.approx((lp - lm) / 2e-6, rel=1e-5, abs=1e-6), (name, pos)


def test_score_function_baselines_keep_the_estimator_unbiased_and_cut_its_variance():
    """north_star: REINFORCE with a control-variate baseline.  On a small VCDM problem the mean of the guide-logit gradient
    over many draws is the same with and without a baseline that does not depend on the person's own draw (here: the
    leave-one-out mean of log_r over the particles), and its variance is smaller."""
    from oracle import vi_oracle as vo
    rng = np.random.RandomState(4)
    N, J, K, S, reps = 12, 8, 2, 4, 600
    q = (rng.rand(K, J) < 0.6).astype(np.float64)
    q[0, q.sum(0) == 0] = 1
    y = rng.randint(0, 2, size=(N, J)).astype(np.uint8)
    spec = {"family": "cdm_sf", "cdm": "dina", "K": K, "N": N, "amortized": False, "q": q, "attr_prior": 0.5}
    params = vo.init_cdm_sf_params(spec, J, np.float64)
    params["attr_p"] = 0.3 * rng.randn(N, K)
    idx = np.arange(N)
    p = vo.sigmoid(params["attr_p"])
    plain, loo = [], []
    for _ in range(reps):
        attrs = [(rng.rand(N, K) < p).astype(np.float64) for _ in range(S)]
        lrs = [vo.cdm_sf_particle(spec, params, y, idx, a)[2] for a in attrs]
        g0 = np.mean([vo.cdm_sf_particle(spec, params, y, idx, a)[1]["attr_p"] for a in attrs], axis=0)
        g1 = np.mean([vo.cdm_sf_particle(spec, params, y, idx, a, baseline=(sum(lrs) - lrs[s]) / (S - 1))[1]["attr_p"]
                      for s, a in enumerate(attrs)], axis=0)
        plain.append(g0)
        loo.append(g1)
    plain, loo = np.array(plain), np.array(loo)
    se = plain.std(0) / np.sqrt(reps)
    assert np.all(np.abs(plain.mean(0) - loo.mean(0)) < 6 * se + 1e-9)          # same expectation
    assert loo.var(0).mean() < 0.5 * plain.var(0).mean()                        # the control variate pays


def test_irt_score_function_estimator_is_unbiased_and_baseline_cuts_its_variance():
    """north_star's 'REINFORCE score-function gradient with control-variate baseline' for the IRT guide (SURVEY.md App. A.5;
    the reference itself is pathwise there): over many draws the score-function gradient of (loc, log scale) has the same
    mean as the pathwise one, with and without a leave-one-out baseline, and the baseline cuts its variance."""
    from oracle import vi_oracle as vo
    rng = np.random.RandomState(7)
    N, J, S, reps = 6, 12, 4, 4000
    y = rng.randint(0, 2, size=(N, J)).astype(np.uint8)
    y[rng.rand(N, J) < 0.1] = 255
    spec = {"family": "irt", "model": "irt_2pl", "D": 1, "Dc": 1.0, "N": N, "amortized": False, "share_cov": False, "a_free": None}
    params = vo.init_irt_params(spec, J, np.float64)
    params["a"] = 0.5 + rng.rand(1, J)
    params["b"] = 0.5 * rng.randn(1, J)
    params["x_local"] = 0.3 * rng.randn(N, 1)
    params["x_scale"] = -0.5 + 0.2 * rng.randn(N, 1)
    idx = np.arange(N)
    sc = dict(spec, estimator="score")
    path, plain, loo = [], [], []
    for _ in range(reps):
        eps = [rng.randn(N, 1) for _ in range(S)]
        path.append(np.mean([np.concatenate([vo.irt_particle(spec, params, y, idx, e)[1][k].ravel() for k in ("x_local", "x_scale")])
                             for e in eps], axis=0))
        outs = [vo.irt_particle(sc, params, y, idx, e, want_log_r=True) for e in eps]
        lrs = [o[2] for o in outs]
        plain.append(np.mean([np.concatenate([o[1][k].ravel() for k in ("x_local", "x_scale")]) for o in outs], axis=0))
        loo.append(np.mean([np.concatenate([vo.irt_particle(sc, params, y, idx, e, baseline=(sum(lrs) - lrs[s]) / (S - 1))[1][k].ravel()
                                            for k in ("x_local", "x_scale")]) for s, e in enumerate(eps)], axis=0))
        # the item gradients do not depend on the estimator
    path, plain, loo = np.array(path), np.array(plain), np.array(loo)
    se = np.sqrt(plain.var(0) / reps + path.var(0) / reps)
    assert np.all(np.abs(plain.mean(0) - path.mean(0)) < 6 * se + 1e-9)            # same expectation as the pathwise gradient
    se2 = np.sqrt(loo.var(0) / reps + path.var(0) / reps)
    assert np.all(np.abs(loo.mean(0) - path.mean(0)) < 6 * se2 + 1e-9)
    assert loo.var(0).mean() < 0.7 * plain.var(0).mean()                          # the control variate pays
    l0, g0 = vo.irt_particle(spec, params, y, idx, eps[0])
    l1, g1 = vo.irt_particle(sc, params, y, idx, eps[0])
    assert l0 == l1 and all(np.array_equal(g0[k], g1[k]) for k in ("a", "b"))


def test_mvn_score_terms_are_the_gradient_of_log_q():
    """The score of MultivariateNormal(loc, scale_tril = L(M)) at a fixed point, as the oracle's score mode for x_feature > 1
    uses it (L as vi.py:452-454 / :711-714 build it): central differences of log q itself."""
    rng = np.random.RandomState(0)
    B, D = 3, 5
    M, loc, eps = 0.3 * rng.randn(B, D, D), rng.randn(B, D), rng.randn(B, D)

    def l_of(m):
        return np.tril(m, -1) + np.einsum("bi,ij->bij", np.exp(np.einsum("bii->bi", m)), np.eye(D))

    x = loc + np.einsum("bij,bj->bi", l_of(M), eps)

    def logq(lc, m):
        L = l_of(m)
        e = np.stack([np.linalg.solve(L[i], x[i] - lc[i]) for i in range(B)])
        return -np.log(np.einsum("bii->bi", L)).sum(1) - 0.5 * (e ** 2).sum(1)

    s_loc, s_m = vo.mvn_score_terms(l_of(M), eps)
    h = 1e-6
    for i in range(B):
        for k in range(D):
            d = np.zeros_like(loc)
            d[i, k] = h
            assert s_loc[i, k] == pytest.approx((logq(loc + d, M)[i] - logq(loc - d, M)[i]) / (2 * h), abs=1e-6)
            for c in range(D):
                dm = np.zeros_like(M)
                dm[i, k, c] = h
                fd = (logq(loc, M + dm)[i] - logq(loc, M - dm)[i]) / (2 * h)
                assert s_m[i, k, c] == pytest.approx(fd, abs=1e-6), (i, k, c)
