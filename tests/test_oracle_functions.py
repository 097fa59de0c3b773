"""Oracle building blocks against values computed by importing the reference's own pure-torch
functions (tests/golden/functions.npz)."""
import os

import numpy as np

from oracle import vi_oracle as vo

F = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "functions.npz")))


def _p(model, x, a, b, c, d, Dc):
    z = Dc * ((x + b) if model == "irt_1pl" else (x @ a + b))
    sg = vo.sigmoid(z)
    lo = c if model in ("irt_3pl", "irt_4pl") else 0.0
    hi = d if model == "irt_4pl" else 1.0
    return lo + (hi - lo) * sg


def test_response_functions_match_reference():            # vi.py:22-66
    for Dc, t in ((1.0, "D1"), (1.702, "D1702")):
        np.testing.assert_allclose(_p("irt_1pl", F["x1"], None, F["b"], None, None, Dc), F["irt_1pl_" + t], rtol=2e-6)
        np.testing.assert_allclose(_p("irt_2pl", F["x1"], F["a1"], F["b"], None, None, Dc), F["irt_2pl_1d_" + t], rtol=2e-6)
        np.testing.assert_allclose(_p("irt_2pl", F["x3"], F["a3"], F["b"], None, None, Dc), F["irt_2pl_3d_" + t], rtol=2e-6)
        np.testing.assert_allclose(_p("irt_3pl", F["x3"], F["a3"], F["b"], F["c"], None, Dc), F["irt_3pl_3d_" + t], rtol=2e-6)
        np.testing.assert_allclose(_p("irt_4pl", F["x3"], F["a3"], F["b"], F["c"], F["d"], Dc), F["irt_4pl_3d_" + t], rtol=2e-6)


def test_dina_and_pattern_table():                         # vi.py:69-83, 825-837
    for K in (1, 2, 3, 4):
        np.testing.assert_array_equal(vo.all_attrs(K), F["all_attrs_K%d" % K])
    q, attr, g, s = F["cdm_q"], F["cdm_attr"], F["cdm_g"], F["cdm_s"]
    eta, al = vo.dina_eta(3, q.astype(np.float64))
    # rows of `attr` are patterns; find their index in the LSB-first table
    idx = (attr * (2 ** np.arange(3))).sum(1).astype(int)
    p = np.where(eta[idx] == 1, 1 - s, g)
    np.testing.assert_allclose(p, F["dina_p"], rtol=1e-6)


def test_encoders_and_cholesky_transform():                # vi.py:417-455, 686
    yin = F["enc_in"].astype(np.float64)
    W = {k: F["norm_enc/" + k].astype(np.float64) for k in vo.ENC_KEYS}
    loc, raw, _ = vo.enc_forward(W, yin)
    np.testing.assert_allclose(loc, F["norm_enc_loc"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(np.exp(raw), F["norm_enc_scale"], rtol=1e-5)
    W = {k: F["mvn_enc/" + k].astype(np.float64) for k in vo.ENC_KEYS}
    loc, raw, _ = vo.enc_forward(W, yin)
    r, c = vo.tril_rows_cols(3)
    M = np.zeros((5, 3, 3))
    M[:, r, c] = raw
    np.testing.assert_allclose(loc, F["mvn_enc_loc"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(M, F["mvn_enc_M"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(vo.constrained("x_scale", M), F["mvn_enc_L"], rtol=1e-5, atol=1e-6)


def test_missing_mask_and_bernoulli_conventions():         # vi.py:617-625; torch clamp_probs
    y = F["mask_y"]
    idx = F["mask_idx"]
    p = _p("irt_2pl", F["mask_x"][idx], F["mask_a"], F["mask_b"], None, None, 1.0)
    p = np.where(y[idx] == 255, 0.0, p)
    np.testing.assert_allclose(p, F["mask_p"], rtol=2e-6)
    np.testing.assert_array_equal(np.where(y[idx] == 255, 0, y[idx]), F["mask_data"])
    probs = F["bern_probs"].astype(np.float32)
    lp1, _ = vo.bernoulli_logprob_probs(probs, np.ones(5, np.uint8))
    lp0, _ = vo.bernoulli_logprob_probs(probs, np.zeros(5, np.uint8))
    # atol 1.2e-7: at the clamp, -BCE_with_logits is -log1p(eps) = -1.19e-7 with the max-based
    # formula of the pinned torch 1.6 (requirements.txt:2) and -0.0 with torch 2.10's log_sigmoid
    # formula that produced the fixture ("in {-0.0, -1.2e-7}"); the oracle keeps
    # the mathematical value.
    np.testing.assert_allclose(lp1, F["bern_lp1"], rtol=1e-6, atol=1.2e-7)
    np.testing.assert_allclose(lp0, F["bern_lp0"], rtol=1e-6, atol=1.2e-7)
    # a missing cell contributes log_prob(0 | P=0) = -log1p(eps) and no gradient (vi.py:621-624)
    lpm, dm = vo.bernoulli_logprob_probs(np.array([0.7], np.float32), np.array([255], np.uint8))
    assert abs(lpm[0] - (-np.log1p(vo.EPS32))) < 1e-12 and dm[0] == 0.0


def test_generator_identification_patterns():              # vi.py:257-258, 378-379, 150-153
    a = F["irt2pl_a_d3"]
    assert (a[1, -1:] == 0).all() and (a[2, -2:] == 0).all() and (a[0] != 0).all()
    a = F["mil_a"]
    for i in range(4):
        assert (a[i, 9 - i:] == 0).all()
    assert (a[np.tril(np.ones((4, 9)), 9 - 4).astype(bool) & (a != 0)] >= 0.01 - 1e-7).all()
    assert (F["hodina_q"].sum(0) > 0).all()
    assert (vo.default_a_free(4, 9) == (a != 0)).all()
