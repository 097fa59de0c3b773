"""GPU parity: the HIP path (through the C ABI) against the oracle and the golden vectors."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import vi_oracle as vo
from tests import golden_util as gu

pytestmark = pytest.mark.gpu


# every gradient against the oracle, as a fraction of its tensor's largest magnitude: the tolerance of the golden replays
# (round 3: the largest value any case of this file reaches is 2.2e-6; rounds 1-2 accepted 2e-4 to 3e-4 here)
GRAD_TOL = 3e-5


def _dev():
    return torch.device("cuda:0")


def _engine_from_fixture(tag):
    from vipsy_amd.engine import IrtEngine, HoDinaEngine, LrSpec
    spec, params, opt, y, steps, B = gu.build(tag, np.float64)
    enc = {k.split("$$$")[1]: v for k, v in params.items() if k.startswith("encoder$$$")}
    yt = torch.from_numpy(y).to(_dev())
    if spec["family"] == "ccdm":
        from vipsy_amd.engine import CcdmEngine
        eng = CcdmEngine(yt, spec["q"], cdm=spec["cdm"], seed=1)
    elif spec["family"] == "hodina":
        eng = HoDinaEngine(yt, spec["q"], amortized=spec["amortized"], H=(enc["fc1.weight"].shape[0] if enc else 0),
                           encoder_init=enc if enc else None, seed=1)
    else:
        custom = spec["D"] > 1 and not np.array_equal(spec["a_free"], vo.default_a_free(spec["D"], y.shape[1]))
        eng = IrtEngine(yt, model=spec["model"], D=spec["D"], Dc=spec["Dc"], amortized=spec["amortized"],
                        H=(enc["fc1.weight"].shape[0] if enc else 0), encoder_init=enc if enc else None,
                        b0=params["b"], share_cov=spec["share_cov"], seed=1,
                        a_free=spec["a_free"].astype(np.float32) if custom else None,     # the CFA call pattern, test.py:426-429
                        a0=params["a"] if custom else None)
    for name in eng.all_names():                      # fixture-specific initial values (e.g. pre-seeded lam1)
        eng.unconstrained(name).copy_(torch.from_numpy(np.asarray(params[name], np.float32)).reshape(
            eng.unconstrained(name).shape))
    lrs = LrSpec(opt["lr"], milestones=opt["milestones"], gamma=opt["gamma"])
    return eng, lrs, spec, params, opt, y, steps


def test_philox_words_bit_exact():
    from vipsy_amd import _hip
    L = _hip.lib()
    n = 1000
    out = torch.empty(n, 4, dtype=torch.int32, device=_dev())
    for seed, step, stream, gid0 in [(1234, 0, 0, 0), (0xDEADBEEFCAFE, 77, 3, (1 << 33) + 5)]:
        _hip.check(L.vx_philox_raw(_hip.ptr(out), gid0, n, seed, step, stream, _hip.stream_ptr()), "philox_raw")
        torch.cuda.synchronize()
        got = out.cpu().numpy().view(np.uint32)
        gids = gid0 + np.arange(n, dtype=np.int64)
        w = vo.philox4x32_10((gids & 0xFFFFFFFF).astype(np.uint32), ((gids >> 32) & 0xFFFFFFFF).astype(np.uint32),
                             np.uint32(step), np.uint32(stream << 16), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
        np.testing.assert_array_equal(got, np.stack(w, axis=1))


@pytest.mark.parametrize("D", [1, 3, 100])
def test_philox_normals_match_oracle(D):
    from vipsy_amd import _hip
    L = _hip.lib()
    n = 4096
    out = torch.empty(n, D, dtype=torch.float32, device=_dev())
    gids = torch.randperm(1 << 20, device=_dev())[:n].to(torch.int64) + (1 << 32)
    _hip.check(L.vx_philox_normals(_hip.ptr(out), _hip.ptr(gids), 0, n, D, 99, 5, 2, _hip.stream_ptr()), "normals")
    torch.cuda.synchronize()
    ref = vo.philox_normals(99, 5, 2, gids.cpu().numpy(), D)
    np.testing.assert_allclose(out.cpu().numpy(), ref, atol=2e-5, rtol=1e-5)
    assert abs(float(out.mean())) < 0.05 and abs(float(out.std()) - 1) < 0.05


GOLDEN_HIP = ["vaeirt_irt_2pl_d3", "vaeirt_irt_3pl_d2", "vaeirt_irt_4pl_d4",
              "virt_irt_1pl_d1", "virt_irt_2pl_d1", "virt_irt_3pl_d1", "virt_irt_4pl_d1", "virt_irt_2pl_d1_D1702",
              "vaeirt_irt_2pl_d1", "vaeirt_irt_4pl_d1",
              "vchodina_k3", "vchodina_k4_sub", "vchodina_k4_clamp", "vaechodina_k3",
              "virt_irt_2pl_d3_perperson", "virt_irt_2pl_d3_share",
              "vccdm_dina_k3", "vccdm_dina_k4_sub", "vccdm_dino_k3",
              # round 2: the reference's default encoder width with J > 128, widths that are neither 8 nor 64, HO-DINA J > 128
              "vaeirt_irt_2pl_d2_h64_j130", "vaeirt_irt_4pl_d3_h24", "vaeirt_irt_2pl_d1_h24", "vchodina_k3_j130"]
GOLDEN_HIP_PARTICLES = ["virt_irt_2pl_d1_particles2", "virt_cfa_d2_particles3"]


@pytest.mark.parametrize("tag", GOLDEN_HIP)
def test_hip_replays_reference_steps(tag):
    """Same y / eps / idx as the reference run -> same loss, gradients and Adam trajectory."""
    eng, lrs, spec, params, opt, y, steps = _engine_from_fixture(tag)
    adam = vo.Adam(opt["lr"], milestones=opt["milestones"], gamma=opt["gamma"])
    for t, rec in enumerate(steps):
        idx, eps = rec["idx"][0], rec["eps"][0]
        rows = torch.from_numpy(idx).to(_dev())
        full = len(idx) == spec["N"] and (idx == np.arange(spec["N"])).all()
        eng.loss_and_grads(None if full else rows, len(idx),
                           None if eps is None else torch.from_numpy(np.ascontiguousarray(eps, dtype=np.float32)).to(_dev()))
        torch.cuda.synchronize()
        loss_o, g_o = vo.loss_and_grads(spec, params, y, [idx], [eps])
        loss_h = float(eng.G[eng.n_params].item())
        assert loss_h == pytest.approx(loss_o, rel=2e-5), (tag, t)
        assert loss_h == pytest.approx(rec["loss"], rel=2e-4), (tag, t)
        for name, go in g_o.items():
            gh = eng.unconstrained(name, eng.GP if (eng.per_person and name in eng.pp_off) else eng.G).cpu().numpy()
            if name == "a" and spec.get("a_free") is not None:
                gh = gh * spec["a_free"]
            sc = max(1e-3, float(np.abs(go).max()))
            np.testing.assert_allclose(gh / sc, go / sc, atol=3e-5, err_msg="%s step %d grad %s" % (tag, t, name))
            np.testing.assert_allclose(gh / sc, rec["grad"][name] / sc, atol=1e-3, err_msg="golden %s %s" % (tag, name))
        eng.allreduce()
        eng.apply_optim(lrs)
        lrs.scheduler_step()
        adam.step(params, g_o)
        adam.scheduler_step()
        torch.cuda.synchronize()
        _check_params_after_step(eng, params, steps, t, tag)


def _check_params_after_step(eng, params, steps, t, tag):
    """Parameters after the Adam step: HIP vs oracle vs golden.  Entries whose gradient is float32 rounding noise (Adam
    moves them by +-lr on the sign of the noise, in the reference too: tests/golden_util.py::adam_conditioned) are not
    compared; engine and oracle follow the reference's value there so that later steps compare like for like."""
    rec = steps[t]
    for name, p in rec["param"].items():
        ok = gu.adam_conditioned(steps, t, name) if name in rec["grad"] else np.ones(p.shape, bool)
        ph = eng.unconstrained(name).cpu().numpy()
        np.testing.assert_allclose(ph[ok], params[name][ok], atol=2e-5, rtol=1e-4, err_msg="%s step %d param %s" % (tag, t, name))
        np.testing.assert_allclose(ph[ok], p[ok], atol=2e-4, rtol=1e-3, err_msg="golden %s step %d param %s" % (tag, t, name))
        if not ok.all():
            params[name][~ok] = p[~ok]
            ph[~ok] = p[~ok]
            eng.unconstrained(name).copy_(torch.from_numpy(ph).to(eng.dev))


@pytest.mark.parametrize("tag", GOLDEN_HIP_PARTICLES)
def test_hip_replays_reference_steps_particles(tag):
    """num_particles > 1 (Trace_ELBO(num_particles=S); test.py:430 uses 20): every particle draws its own subsample and
    eps, the surrogate is the mean over particles -- replayed through engine.step, which owns the
    accumulation; includes the CFA call pattern with custom a_free / a0 masks (test.py:418-430)."""
    eng, lrs, spec, params, opt, y, steps = _engine_from_fixture(tag)
    adam = vo.Adam(opt["lr"], milestones=opt["milestones"], gamma=opt["gamma"])
    for t, rec in enumerate(steps):
        S = len(rec["idx"])
        assert S > 1
        rows = [torch.from_numpy(i).to(_dev()) for i in rec["idx"]]
        eps = [torch.from_numpy(np.ascontiguousarray(e, dtype=np.float32)).to(_dev()) for e in rec["eps"]]
        loss_h = float(eng.step(lrs, rows=rows, b_global=len(rec["idx"][0]), eps=eps, num_particles=S).item())
        lrs.scheduler_step()
        torch.cuda.synchronize()
        loss_o, g_o = vo.loss_and_grads(spec, params, y, rec["idx"], rec["eps"])
        assert loss_h == pytest.approx(loss_o, rel=2e-5), (tag, t)
        assert loss_h == pytest.approx(rec["loss"], rel=2e-4), (tag, t)
        for name, go in g_o.items():
            gh = eng.unconstrained(name, eng.GP if (eng.per_person and name in eng.pp_off) else eng.G).cpu().numpy()
            if name == "a" and spec.get("a_free") is not None:
                gh = gh * spec["a_free"]
            sc = max(1e-3, float(np.abs(go).max()))
            np.testing.assert_allclose(gh / sc, go / sc, atol=3e-5, err_msg="%s step %d grad %s" % (tag, t, name))
            np.testing.assert_allclose(gh / sc, rec["grad"][name] / sc, atol=1e-3, err_msg="golden %s %s" % (tag, name))
        adam.step(params, g_o)
        adam.scheduler_step()
        _check_params_after_step(eng, params, steps, t, tag)


def _random_problem(N, J, D, H, model, miss, seed):
    rng = np.random.RandomState(seed)
    y = rng.randint(0, 2, size=(N, J)).astype(np.uint8)
    y[rng.rand(N, J) < miss] = 255
    enc = {"fc1.weight": rng.randn(H, J) / np.sqrt(J), "fc1.bias": 0.1 * rng.randn(H),
           "fc21.weight": rng.randn(D, H) / np.sqrt(H), "fc21.bias": 0.1 * rng.randn(D),
           "fc22.weight": 0.3 * rng.randn(D * (D + 1) // 2, H) / np.sqrt(H), "fc22.bias": 0.05 * rng.randn(D * (D + 1) // 2)}
    return y, enc, rng


@pytest.mark.parametrize("N,J,D,H,model,miss,B", [
    (300, 500, 100, 64, "irt_2pl", 0.0, None),      # headline shape (test_ai_100_dim_2pl, test.py:336)
    (257, 130, 33, 40, "irt_2pl", 0.3, 101),        # ragged everything, subsample
    (129, 260, 7, 64, "irt_4pl", 0.2, None),
    (64, 33, 2, 16, "irt_3pl", 0.5, 17),
    (1000, 600, 12, 64, "irt_2pl", 0.59, None),     # J > 512: two item groups
    (200, 131, 101, 32, "irt_4pl", 0.2, 77),        # D >= 64: register-resident likelihood kernel, ragged staging
    (150, 260, 64, 64, "irt_3pl", 0.1, None),
    (130, 516, 127, 64, "irt_2pl", 0.3, None),
    (320, 500, 100, 64, "irt_2pl", 0.1, None),      # N % 16 == 0: item-major responses, dimension-major fc1 gradient
    (1040, 516, 64, 64, "irt_3pl", 0.2, None),      # ... with two 512-item groups and a ragged last person tile
    (64, 40, 8, 64, "irt_2pl", 0.0, None),
    (36, 40, 8, 64, "irt_2pl", 0.1, None),          # a single ragged person tile in every dimension-major kernel
    (9000, 40, 8, 64, "irt_2pl", 0.1, None),        # more than 8192 persons: whole 128-person workgroups (fwd_b<false>)
    (40000, 40, 8, 64, "irt_2pl", 0.1, None),       # over half a chip round: the 64-persons-per-wave forward (k_mvn_fwd_b2.hip), ragged end
    (40008, 72, 12, 64, "irt_2pl", 0.1, None),      # ... whose last wave holds 8 persons (no LDS transposes, x rows as staging)
    (70200, 72, 12, 64, "irt_3pl", 0.2, 70000),     # one full round there + a 4 464-person tail on fwd_b<false>, row gather
    (200, 90, 6, 96, "irt_2pl", 0.2, None),         # hidden_dim > 64 (vi.py:417-455 takes any width): generic kernels
    (150, 260, 33, 128, "irt_4pl", 0.1, 60),        # ... up to 128
    (4, 36, 4, 64, "irt_2pl", 0.0, None),           # the smallest batch those kernels accept
    (2000, 500, 100, 64, "irt_2pl", 0.0, 1000),     # headline shape with a row gather (subsample): FAST == 2 staging
    # shapes aimed at the static structure of the bf16x3 kernels: k-blocks of 16 in the hidden gradient (one block,
    # two blocks, all eight), a partial last DIAG / LOC tile, fc1 k-steps with a ragged tail, a ragged last person tile
    (260, 48, 16, 64, "irt_2pl", 0.1, None),
    (200, 64, 20, 64, "irt_3pl", 0.2, None),
    (136, 100, 124, 64, "irt_2pl", 0.1, None),
    (300, 260, 112, 64, "irt_4pl", 0.3, 104),
    (264, 36, 36, 64, "irt_2pl", 0.0, None),
    # the MFMA likelihood kernels (96 <= D <= 111, full batch): f16x2 (k_irt_lik_h, 1PL / 2PL link) at the ends of its D
    # range, ragged item chunks and person tiles; bf16x3 (k_irt_lik_b) for the 3PL / 4PL links, which make their own image
    (1104, 260, 96, 64, "irt_2pl", 0.1, None),
    (464, 132, 108, 64, "irt_2pl", 0.2, None),
    (320, 500, 100, 64, "irt_4pl", 0.1, None),
    (208, 260, 104, 64, "irt_3pl", 0.2, None),
    # item counts that are no multiple of 4: phantom items (IrtEngine.__init__) keep the MFMA kernels -- absent for the
    # likelihood, zero for the encoder; loss and every gradient over the problem's own items as if they were not there
    (300, 499, 100, 64, "irt_2pl", 0.1, None),      # ... k_irt_lik_h through the item-major copy
    (320, 501, 100, 64, "irt_4pl", 0.1, None),      # ... k_irt_lik_b
    (2000, 498, 100, 64, "irt_2pl", 0.0, 1000),     # ... a subsample: the likelihood reads the person-major copy with byte 254
    (9000, 37, 8, 64, "irt_2pl", 0.1, None),        # ... a small shape, whole workgroups of the forward
    # latent dimensions that are no multiple of 4: phantom dimensions (a_k = 0 for ever, zero head rows kept at zero)
    (1000, 45, 3, 64, "irt_3pl", 0.2, None),        # ... with phantom items
    (320, 500, 99, 64, "irt_2pl", 0.1, None),       # ... 99 + 1 dimensions on k_irt_lik_h and the dimension-major backward
    (2000, 500, 98, 64, "irt_2pl", 0.0, 1000),      # ... a subsample
    (9001, 40, 6, 64, "irt_2pl", 0.1, None),        # ... with phantom persons
    (333, 131, 10, 64, "irt_4pl", 0.2, 77),         # ... everything ragged, a batch that is no multiple of 4 (person-major backward)
])
def test_mvn_amortized_step_vs_oracle(N, J, D, H, model, miss, B):
    from vipsy_amd.engine import IrtEngine, ENC_KEYS
    y, enc, rng = _random_problem(N, J, D, H, model, miss, seed=N + J + D)
    eng = IrtEngine(torch.from_numpy(y).to(_dev()), model=model, D=D, amortized=True, H=H,
                    encoder_init={k: v.astype(np.float32) for k, v in enc.items()}, seed=11)
    # randomise the item parameters a little so every term is exercised
    a0 = (eng.unconstrained("a") * torch.from_numpy(1 + 0.3 * rng.randn(D, J)).float().to(_dev()))
    eng.unconstrained("a").copy_(a0 * eng.unconstrained("a", eng.free))
    eng.unconstrained("b").copy_(torch.from_numpy(0.5 * rng.randn(1, J)).float())
    if model in ("irt_3pl", "irt_4pl"):
        eng.unconstrained("c").add_(torch.from_numpy(0.3 * rng.randn(1, J)).float().to(_dev()))
    if model == "irt_4pl":
        eng.unconstrained("d").add_(torch.from_numpy(0.3 * rng.randn(1, J)).float().to(_dev()))
    idx = np.arange(N) if B is None else np.sort(rng.permutation(N)[:B])
    rows = None if B is None else torch.from_numpy(idx).to(_dev())
    eng.loss_and_grads(rows, len(idx))
    torch.cuda.synchronize()
    nb = len(idx)
    fw = eng.last["fw"]
    eps = fw["eps"][:nb * eng.D].reshape(nb, eng.D)[:, :D].cpu().numpy()      # (eng.D: D + phantom dimensions, if any)
    # RNG inside the fused kernel == vx_philox_normals == oracle spec, keyed by global person id
    np.testing.assert_allclose(eps, vo.philox_normals(11, 0, 0, idx, D), atol=2e-5)
    spec = {"family": "irt", "model": model, "D": D, "Dc": 1.0, "N": N, "amortized": True, "share_cov": False,
            "a_free": vo.default_a_free(D, J)}
    params = {n: eng.unconstrained(n).cpu().numpy().astype(np.float64) for n in eng.names()}
    loss_o, g_o = vo.loss_and_grads(spec, params, y, [idx], [eps])
    loss_h = float(eng.G[eng.n_params].item())
    assert loss_h == pytest.approx(loss_o, rel=3e-5)
    for name, go in g_o.items():
        gh = eng.unconstrained(name, eng.G).cpu().numpy() * eng.unconstrained(name, eng.free).cpu().numpy()
        sc = max(1e-6, float(np.abs(go).max()))
        err = np.abs(gh - go).max() / sc
        assert err < GRAD_TOL, (name, err)


def _oracle_headline_chunked(eng, y, eps, model="irt_2pl", chunk=2048):
    """The oracle on a full batch too large for one call ((B, D, D) temporaries): a full-batch loss and every gradient
    are sums over the persons (plate scale N / B = 1), so person chunks are evaluated with spec N = chunk size and added.
    Also returns the per-person forward values x, h and ent = 0.5 |eps|^2 + sum_k M_kk of the guide (vi.py:448-455)."""
    N, J = y.shape
    D, H = eng.D, eng.H
    params = {n: eng.unconstrained(n).cpu().numpy().astype(np.float64) for n in eng.names()}
    W = {k: params["encoder$$$" + k] for k in vo.ENC_KEYS}
    r_, c_ = vo.tril_rows_cols(D)
    dsel = np.flatnonzero(r_ == c_)
    loss, grads = 0.0, None
    x_o, h_o, ent_o = np.empty((N, D)), np.empty((N, H)), np.empty(N)
    for lo in range(0, N, chunk):
        hi = min(N, lo + chunk)
        yc, ec = y[lo:hi], eps[lo:hi].astype(np.float64)
        spec = {"family": "irt", "model": model, "D": D, "Dc": 1.0, "N": hi - lo, "amortized": True, "share_cov": False,
                "a_free": vo.default_a_free(D, J)}
        l, g = vo.loss_and_grads(spec, params, yc, [np.arange(hi - lo)], [ec])
        loss += l
        grads = g if grads is None else {k: grads[k] + g[k] for k in g}
        loc, raw, cache = vo.enc_forward(W, vo.enc_input(yc, np.float64))
        h_o[lo:hi] = cache[2]
        xc = loc.copy()
        col0 = 0
        for k in range(D):                             # row k of L: raw[(k, 0..k-1)] off the diagonal, exp on it (vi.py:452-454)
            xc[:, k] += (raw[:, col0:col0 + k] * ec[:, :k]).sum(1) + np.exp(raw[:, col0 + k]) * ec[:, k]
            col0 += k + 1
        x_o[lo:hi] = xc
        ent_o[lo:hi] = 0.5 * (ec ** 2).sum(1) + raw[:, dsel].sum(1)
    return loss, grads, x_o, h_o, ent_o


Z_CLAMP = float(np.log((1.0 - vo.EPS32) / vo.EPS32))       # 15.9424: where torch's clamp_probs cuts the Bernoulli logit


def _oracle_latents_chunked(params, y, eps, D, chunk=4096):
    """x = loc + L eps of the amortized guide (vi.py:448-455, 692-693) for every person, float64, in person chunks."""
    W = {k: params["encoder$$$" + k] for k in vo.ENC_KEYS}
    x = np.empty((y.shape[0], D))
    for lo in range(0, y.shape[0], chunk):
        hi = min(y.shape[0], lo + chunk)
        loc, raw, _ = vo.enc_forward(W, vo.enc_input(y[lo:hi], np.float64))
        ec = eps[lo:hi].astype(np.float64)
        xc, col0 = loc.copy(), 0
        for k in range(D):
            xc[:, k] += (raw[:, col0:col0 + k] * ec[:, :k]).sum(1) + np.exp(raw[:, col0 + k]) * ec[:, k]
            col0 += k + 1
        x[lo:hi] = xc
    return x


@pytest.mark.parametrize("N,slopes,J", [
    (33024, "small", 499),  # the same kernels on 499 items + one phantom item (IrtEngine.__init__: absent / zero)
    (33021, "small", 498),  # ... and three phantom persons
])
def test_headline_large_batch_kernels_with_phantom_items_vs_oracle(N, slopes, J):
    test_headline_large_batch_kernels_vs_oracle(N, slopes, J)


@pytest.mark.parametrize("N,slopes", [
    (33024, "small"),   # 129 workgroups of 256 persons: every person on k_mvn_enc_fwd_b2 (64 per wave) and bwd_h_b2
    (70016, "small"),   # one full chip round (65 536) on fwd_b2 + a 4 480-person tail on k_mvn_enc_fwd_b<false>
    (33024, "ones"),    # the same kernels in the BENCH's regime: a as vi.py:567-572 initialises it (ones + the zero pattern),
    (70016, "ones"),    # |z| up to 40, every ninth cell beyond the clamp; the few cells ON the clamp are marked missing
    (33021, "small"),   # a shard whose size is no multiple of 8 (nor of 4): three phantom persons (engine.py::_pad_persons) keep
    (70010, "small"),   # it on these kernels -- six of them, a chip round + tail; every gradient and the loss as if they were not there
])
def test_headline_large_batch_kernels_vs_oracle(N, slopes, J=500):
    """The kernels that run the judged 1M x 500 x 100 step -- the large-batch forms of the forward (k_mvn_fwd_b2.hip,
    k_mvn_enc_fwd_b<false>), the hidden gradient (k_mvn_enc_bwd_h_b2 / _b<false>), bwd_w_b, lik_h, fc1_bwd_b -- at the
    headline's own J = 500, D = 100, H = 64 (all eight k-blocks of 16, 176 head tiles, multi-block DIAG / LOC sections),
    full batch, 10 % missing, against the oracle: loss, every gradient, and x / h / ent of every person.

    slopes = 'small': slopes of 0.05 (1 +- 0.3), |z| stays under 15 in all 33 M cells.  slopes = 'ones' (VERDICT round 3,
    weak #1): the reference's own initial a (vi.py:567-572), the regime bench.py runs in.  There the reference's gradient
    JUMPS from -1 to 0 where |z| crosses logit(1 - eps) = 15.9424 (torch clamp_probs), and a cell within
    float32 rounding of that point may fall on either side in ANY float32 evaluation (the reference's included;
    tools/lik_err_probe.py: a dozen of 33 M cells, each moving an entry of G_a by |x|).  Those cells -- oracle |z| within
    1e-3 |z| of the clamp point -- are set to 255 (missing) in y BEFORE either side runs; since the responses are also the
    encoder's input, the marking is iterated until no unmarked cell is left in the band.  Same tolerance as the rest."""
    from vipsy_amd.engine import IrtEngine
    D, H = 100, 64
    y, _, rng = _random_problem(N, J, D, H, "irt_2pl", 0.1, seed=N)
    a_mult = 0.05 * (1 + 0.3 * rng.randn(D, J))
    b_new = 0.5 * rng.randn(1, J)

    def make_engine(yy):
        # encoder: the nn.Linear default initialisation, as in the judged run
        e = IrtEngine(torch.from_numpy(yy).to(_dev()), model="irt_2pl", D=D, amortized=True, H=H, seed=11)
        if slopes == "small":
            a0 = e.unconstrained("a") * torch.from_numpy(a_mult).float().to(_dev())
            e.unconstrained("a").copy_(a0 * e.unconstrained("a", e.free))
        e.unconstrained("b").copy_(torch.from_numpy(b_new).float())
        return e

    eng = make_engine(y)
    n_marked = 0
    if slopes == "ones":
        params = {n: eng.unconstrained(n).cpu().numpy().astype(np.float64) for n in eng.names()}
        eps_o = vo.philox_normals(11, 0, 0, np.arange(N), D)
        z = _oracle_latents_chunked(params, y, eps_o, D) @ params["a"] + params["b"]
        dirty = np.arange(N)
        for _ in range(60):
            # a marked response changes its person's encoder input, hence all 500 logits of that person: every marked person is
            # evaluated again (a new cell lands in the band with probability ~0.35 per marked person: geometric decay)
            band = np.zeros(y.shape, dtype=bool)
            zd = z[dirty]
            band[dirty] = (np.abs(np.abs(zd) - Z_CLAMP) < 1e-3 * np.abs(zd)) & (y[dirty] != 255)
            if not band.any():
                break
            y[band] = 255
            n_marked += int(band.sum())
            dirty = np.flatnonzero(band.any(1))
            z[dirty] = _oracle_latents_chunked(params, y[dirty], eps_o[dirty], D) @ params["a"] + params["b"]
        else:
            raise AssertionError("marking the cells on the clamp did not settle")
        assert not ((np.abs(np.abs(z) - Z_CLAMP) < 1e-3 * np.abs(z)) & (y != 255)).any()
        assert (np.abs(z) > Z_CLAMP).mean() > 0.05 and n_marked < 3e-3 * y.size       # the bench's regime; under 0.3 % marked
        print("slope-1 variant, N = %d: %d of %d cells on the clamp marked missing" % (N, n_marked, y.size))
        eng = make_engine(y)
    eng.loss_and_grads()
    torch.cuda.synchronize()
    fw = eng.last["fw"]
    eps = fw["eps"][:N * D].reshape(N, D).cpu().numpy()
    np.testing.assert_allclose(eps, vo.philox_normals(11, 0, 0, np.arange(N), D), atol=2e-5)
    loss_o, g_o, x_o, h_o, ent_o = _oracle_headline_chunked(eng, y, eps)
    z_o = x_o @ eng.unconstrained("a").double().cpu().numpy() + eng.unconstrained("b").double().cpu().numpy()
    if slopes == "small":
        assert np.abs(z_o).max() < 15.0                                                 # no cell near the clamp
    else:                                               # (with the kernel's own draws, 2e-5 off the oracle's: a band still clear)
        assert not ((np.abs(np.abs(z_o) - Z_CLAMP) < 2e-4 * np.abs(z_o)) & (y != 255)).any()
    # per-person forward values, every person
    x_h = fw["x"][:N * D].reshape(N, D).cpu().numpy()
    h_h = fw["h"][:N * H].reshape(N, H).cpu().numpy()
    ent_h = fw["ent"][:N].cpu().numpy()
    np.testing.assert_allclose(x_h, x_o, atol=2e-5 * max(1.0, np.abs(x_o).max()), rtol=1e-5)
    np.testing.assert_allclose(h_h, h_o, atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(ent_h, ent_o, atol=2e-5 * max(1.0, np.abs(ent_o).max()), rtol=1e-5)
    # the dimension-major copies the backward kernels read
    nbk = eng.last["nb"]                                 # persons the kernels were launched over (N, or N + phantoms)
    assert nbk == (N + 7) // 8 * 8 and eng.last["n_valid"] == N
    hT = fw["hT"][:H * nbk].reshape(H, nbk)[:, :N].cpu().numpy()
    epsT = fw["epsT"][:D * nbk].reshape(D, nbk)[:, :N].cpu().numpy()
    assert np.array_equal(hT, h_h.T) and np.array_equal(epsT, eps.T)
    loss_h = float(eng.G[eng.n_params].item())
    assert loss_h == pytest.approx(loss_o, rel=3e-5)
    errs = {}
    for name, go in g_o.items():
        gh = eng.unconstrained(name, eng.G).cpu().numpy() * eng.unconstrained(name, eng.free).cpu().numpy()
        sc = max(1e-6, float(np.abs(go).max()))
        errs[name] = float(np.abs(gh - go).max() / sc)
    print("large-batch gradient errors (of the tensor's max), N = %d: %s" % (N, errs))
    assert max(errs.values()) < GRAD_TOL_LARGE, errs


GRAD_TOL_LARGE = 3e-5       # the tolerance of the golden replays


@pytest.mark.parametrize("N,J,D,H,B", [(2000, 37, 6, 32, None), (1203, 130, 3, 24, 100), (4500, 499, 1, 40, None),
                                       (3000, 500, 100, 64, 50),      # a subsample that is no multiple of 4: two phantom rows
                                       (1500, 45, 7, 64, 10)])        # ... with everything else
def test_phantom_items_dimensions_and_hidden_units_train_like_the_problem_itself(N, J, D, H, B, monkeypatch):
    """Item, dimension and hidden-unit counts the MFMA kernels do not take, padded with phantoms on the host
    (IrtEngine.__init__), against the same problem handed to the kernels as it is (the generic tier): five Adam steps, the same
    losses and the same parameters -- the phantoms never move and never reach the problem's own parameters."""
    from vipsy_amd.engine import IrtEngine, LrSpec
    rng = np.random.RandomState(N + J)
    y = rng.randint(0, 2, size=(N, J)).astype(np.uint8)
    y[rng.rand(N, J) < 0.2] = 255
    draws = np.random.RandomState(3)
    rows_all = [None if B is None else torch.from_numpy(np.sort(draws.choice(N, size=B, replace=False)).astype(np.int64)).to(_dev())
                for _ in range(5)]
    out = []
    for pad in (True, False):
        for name in ("pad_items", "pad_dims", "pad_hidden", "pad_batch"):
            monkeypatch.setattr(IrtEngine, name, pad)
        eng = IrtEngine(torch.from_numpy(y).to(_dev()), model="irt_2pl", D=D, amortized=True, H=H, seed=11)
        assert (eng.J, eng.D, eng.H) == (((J + 3) // 4 * 4, (D + 3) // 4 * 4 if D > 1 else 1, 64) if pad else (J, D, H))
        assert tuple(eng.unconstrained("a").shape) == (D, J) and tuple(eng.unconstrained("encoder$$$fc1.weight").shape) == (H, J)
        lrs = LrSpec(lambda m, p: {"lr": 1e-2})
        losses = [float(eng.step(lrs, rows=rows_all[t], b_global=B)) for t in range(5)]
        torch.cuda.synchronize()
        out.append((np.array(losses), {n: eng.unconstrained(n).double().cpu().numpy().copy() for n in eng.names()}))
    np.testing.assert_allclose(out[0][0], out[1][0], rtol=3e-5)
    for n in out[0][1]:
        # (Adam's first steps move every entry by ~lr whatever its gradient's size, so entries whose gradient is rounding
        # noise may differ by a step: compare where it matters, on the scale of the tensor)
        sc = max(1e-3, float(np.abs(out[1][1][n]).max()))
        assert np.abs(out[0][1][n] - out[1][1][n]).max() <= 2e-3 * sc, n


@pytest.mark.parametrize("N,J,K,H,B", [(3000, 30, 5, 32, None), (2500, 30, 8, 64, 20), (1000, 13, 3, 24, 100)])
def test_amortized_hodina_encoder_with_phantoms_trains_like_the_encoder_itself(N, J, K, H, B, monkeypatch):
    """VaeCHoDina's NormEncoder over item counts / widths the MFMA encoder kernels do not take (the reference's HO-DINA has 30
    items): only the ENCODER sees phantom items and hidden units (HoDinaEngine.__init__), the HO-DINA kernel the problem as it is.
    Five Adam steps against the encoder handed over in its own shape: same losses, same parameters."""
    from vipsy_amd.engine import HoDinaEngine, LrSpec
    rng = np.random.RandomState(N + J)
    q = (rng.rand(K, J) < 0.4).astype(np.float32)
    q[0, q.sum(0) == 0] = 1
    y = rng.randint(0, 2, size=(N, J)).astype(np.uint8)
    y[rng.rand(N, J) < 0.1] = 255
    draws = np.random.RandomState(3)
    rows_all = [None if B is None else torch.from_numpy(np.sort(draws.choice(N, size=B, replace=False)).astype(np.int64)).to(_dev())
                for _ in range(5)]
    out = []
    for pad in (True, False):
        monkeypatch.setattr(HoDinaEngine, "pad_encoder", pad)
        eng = HoDinaEngine(torch.from_numpy(y).to(_dev()), torch.from_numpy(q), amortized=True, H=H, seed=11)
        assert (eng.J_enc, eng.H) == (((J + 3) // 4 * 4, 64) if pad else (J, H)) and eng.J == J
        assert tuple(eng.unconstrained("encoder$$$fc1.weight").shape) == (H, J)
        lrs = LrSpec(lambda m, p: {"lr": 1e-2})
        losses = [float(eng.step(lrs, rows=rows_all[t], b_global=B)) for t in range(5)]
        torch.cuda.synchronize()
        out.append((np.array(losses), {n: eng.unconstrained(n).double().cpu().numpy().copy() for n in eng.names()}))
    np.testing.assert_allclose(out[0][0], out[1][0], rtol=3e-5)
    for n in out[0][1]:
        sc = max(1e-3, float(np.abs(out[1][1][n]).max()))
        assert np.abs(out[0][1][n] - out[1][1][n]).max() <= 2e-3 * sc, n


@pytest.mark.parametrize("N", [4099, 33021])
def test_phantom_persons_add_nothing(N):
    """A full batch whose size is no multiple of 8, launched over the next multiple with phantom persons (all responses missing;
    taken out behind the likelihood: engine.py::_pad_persons) against the same batch launched over its own size (the fp32-MFMA
    generation of the backward kernels): loss and every gradient agree to the parity tolerance, three steps of Adam keep the
    parameters together, and the padded step replayed from its HIP graph equals the padded eager one bit for bit."""
    from vipsy_amd.engine import IrtEngine, LrSpec
    J, D, H = 500, 100, 64
    y, _, _ = _random_problem(N, J, D, H, "irt_2pl", 0.1, seed=N)
    out = []
    for pad, graph in ((True, False), (False, False), (True, True)):      # (the third: the padded step replayed from its graph)
        eng = IrtEngine(torch.from_numpy(y).to(_dev()), model="irt_2pl", D=D, amortized=True, H=H, seed=11)
        eng.pad_persons = pad
        eng.use_graph = graph
        eng.loss_and_grads()
        torch.cuda.synchronize()
        assert eng.last["nb"] == ((N + 7) // 8 * 8 if pad else N) and eng.last["n_valid"] == N
        g = eng.G[:eng.n_params + 1].double().cpu().numpy().copy()
        lrs = LrSpec(lambda m, p: {"lr": 1e-3})
        losses = [float(eng.step(lrs)) for _ in range(3)]
        out.append((g, np.array(losses), eng.P.double().cpu().numpy().copy()))
    assert np.array_equal(out[0][0], out[2][0]) and np.array_equal(out[0][2], out[2][2])
    g_pad, g_own = out[0][0], out[1][0]
    assert g_pad[-1] == pytest.approx(g_own[-1], rel=3e-5)
    assert np.abs(g_pad[:-1] - g_own[:-1]).max() <= GRAD_TOL_LARGE * np.abs(g_own[:-1]).max()
    np.testing.assert_allclose(out[0][1], out[1][1], rtol=3e-5)
    assert np.abs(out[0][2] - out[1][2]).max() <= 1e-4 * max(1.0, np.abs(out[1][2]).max())


@pytest.mark.parametrize("N,J,model,miss,B", [
    (1000, 5, "irt_2pl", 0.0, None),                 # config 1 shape (lsat.dat)
    (5000, 100, "irt_4pl", 0.0, None),               # config 2 shape, scaled down
    (3000, 500, "irt_2pl", 0.9, None),               # config 4 shape, scaled down, 90 % MCAR
    (777, 1000, "irt_2pl", 0.59, 100),               # Irt2PLMissing.test_ai shape (test.py:311-319)
    (100, 65, "irt_1pl", 0.1, 33),
    (100, 129, "irt_3pl", 0.1, None),
    (2000, 500, "irt_2pl", 0.9, None),               # >= 50 % missing, full batch: observed-cell lists (k_irt1d_sparse.hip)
    (777, 130, "irt_4pl", 0.7, None),
    (300, 37, "irt_1pl", 0.6, None),
    (640, 200, "irt_3pl", 0.95, None),
    (4500, 60, "irt_2pl", 0.8, None),                # two sort windows; a few |x| > 8 (full-range slots of the list kernel)
    (300, 1000, "irt_3pl", 0.8, None),               # list kernel at J near its limit (1024), asymptote gradients
    (200, 7, "irt_4pl", 0.7, None),                  # ... and with lists shorter than one quad
])
def test_irt1d_step_vs_oracle(N, J, model, miss, B):
    from vipsy_amd.engine import IrtEngine
    rng = np.random.RandomState(N + J)
    y = rng.randint(0, 2, size=(N, J)).astype(np.uint8)
    y[rng.rand(N, J) < miss] = 255
    eng = IrtEngine(torch.from_numpy(y).to(_dev()), model=model, D=1, seed=5)
    eng.unconstrained("b").copy_(torch.from_numpy(0.7 * rng.randn(1, J)).float())
    if model != "irt_1pl":
        eng.unconstrained("a").copy_(torch.from_numpy(0.5 + 2 * rng.rand(1, J)).float())
    if model in ("irt_3pl", "irt_4pl"):
        eng.unconstrained("c").add_(torch.from_numpy(0.3 * rng.randn(1, J)).float().to(_dev()))
    loc0 = rng.randn(N)
    if N == 4500:
        loc0[::500] = 11.0 * np.sign(loc0[::500])
    eng.PP.copy_(torch.from_numpy(np.concatenate([loc0, 0.3 * rng.randn(N)])).float())
    idx = np.arange(N) if B is None else np.sort(rng.permutation(N)[:B])
    rows = None if B is None else torch.from_numpy(idx).to(_dev())
    eps = vo.philox_normals(5, 0, 0, idx, 1)
    eng.loss_and_grads(rows, len(idx))
    torch.cuda.synchronize()
    spec = {"family": "irt", "model": model, "D": 1, "Dc": 1.0, "N": N, "amortized": False, "share_cov": False,
            "a_free": None}
    params = {n: eng.unconstrained(n).cpu().numpy().astype(np.float64) for n in eng.names() + ["x_local", "x_scale"]}
    loss_o, g_o = vo.loss_and_grads(spec, params, y, [idx], [eps])
    assert float(eng.G[eng.n_params].item()) == pytest.approx(loss_o, rel=3e-5)
    for name, go in g_o.items():
        gh = eng.unconstrained(name, eng.GP if (eng.per_person and name in eng.pp_off) else eng.G).cpu().numpy()
        sc = max(1e-6, float(np.abs(go).max()))
        assert np.abs(gh - go).max() / sc < GRAD_TOL, name


def test_entry_points_reject_bad_arguments():
    from vipsy_amd import _hip
    L = _hip.lib()
    cfg = _hip.IrtCfg(2, 200, 10, 64, 1.0, 1.0, 0, 0, 0)                  # D = 200 > supported 127
    assert L.vx_irt_lik_workspace_floats(ctypes.byref(cfg), 10) == -1
    cfg = _hip.IrtCfg(2, 1, 2000, 0, 1.0, 1.0, 0, 0, 0)                   # J > 1024 for the D=1 kernel
    assert L.vx_irt1d_workspace_floats(ctypes.byref(cfg), 10) == -1
    assert L.vx_philox_normals(None, None, 0, 10, 1, 0, 0, 0, _hip.stream_ptr()) == -1


@pytest.mark.parametrize("N", [512, 33024, 70016])
def test_generic_and_fast_kernels_agree(N, tmp_path):
    """The specialised kernels (bf16x3 and fp32-MFMA generations) against the shape-generic ones (VX_FORCE_GENERIC=1 in a
    child process) on the headline shape: N = 512 (the small-batch forms, every generation), and the two batch sizes of
    test_headline_large_batch_kernels_vs_oracle (the large-batch forms that run the judged step; default kernels only)."""
    import os
    import subprocess
    import sys
    code = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, %r)
from vipsy_amd.engine import IrtEngine
rng = np.random.RandomState(5)
N, J, D, H = int(sys.argv[1]), 500, 100, 64
y = rng.randint(0, 2, size=(N, J)).astype(np.uint8); y[rng.rand(N, J) < 0.2] = 255
eng = IrtEngine(torch.from_numpy(y).cuda(), model="irt_2pl", D=D, amortized=True, H=H, seed=21)
eng.unconstrained("b").copy_(torch.from_numpy(0.5 * rng.randn(1, J)).float())
eng.loss_and_grads()
torch.cuda.synchronize()
np.savez(sys.argv[2], loss=float(eng.G[eng.n_params].item()), g=eng.G[:eng.n_params].double().cpu().numpy(),
         x=eng.last["fw"]["x"][:N * D].cpu().numpy())
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    # the two test seams of the library: VX_FORCE_GENERIC=1 (shape-generic kernels only) and VX_MFMA16 (0: the fp32-MFMA
    # kernels; f / w / h / g: one of the four guide kernels on the bf16 MFMA, the rest fp32)
    switches = {"0": {}, "1": {"VX_FORCE_GENERIC": "1"}}
    if N == 512:
        switches.update({"fp32": {"VX_MFMA16": "0"}, "b3f": {"VX_MFMA16": "f"}, "b3w": {"VX_MFMA16": "w"},
                         "b3h": {"VX_MFMA16": "h"}, "b3g": {"VX_MFMA16": "g"}})
    for mode, extra in switches.items():
        env = dict(os.environ, VX_FORCE_GENERIC="0")
        env.update(extra)
        out = str(tmp_path / ("r_%s.npz" % mode))
        p = subprocess.run([sys.executable, "-c", code, str(N), out], env=env, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        res[mode] = dict(np.load(out))
    # every kernel generation -- the default bf16x3 kernels ("0"), the fp32-MFMA ones and the mixed selections --
    # against the shape-generic ones
    for m in switches:
        if m == "1":
            continue
        assert float(res[m]["loss"]) == pytest.approx(float(res["1"]["loss"]), rel=1e-6)
        np.testing.assert_allclose(res[m]["x"], res["1"]["x"], atol=2e-5, rtol=1e-5)
        g0, g1 = res[m]["g"], res["1"]["g"]
        assert np.abs(g0 - g1).max() <= 2e-5 * max(1.0, np.abs(g1).max())


@pytest.mark.parametrize("N,D", [(33024, 100), (70016, 100), (33024, 92), (33024, 84)])
def test_forward_ring_and_plain_forms_agree(N, D, tmp_path):
    """The large-batch guide forward takes its head tiles through an LDS ring shared by a workgroup's four waves
    (k_mvn_fwd_b2.hip, SH); VX_FWD_RING=0 in a child process selects the form in which every wave pulls them itself.
    The arithmetic and its order are the same: every output of the forward -- and with it the loss and every gradient of
    the step -- has to agree BIT FOR BIT (33 024: whole workgroups only; 70 016: a ragged last tile beside the whole
    rounds; D = 92 and 84: the other latent widths whose tiles fit the ring's LDS budget, eps rows of a different bank
    phase).  Both are checked against the oracle by test_headline_large_batch_kernels_vs_oracle."""
    import os
    import subprocess
    import sys
    code = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, %r)
from vipsy_amd.engine import IrtEngine
rng = np.random.RandomState(9)
N, J, D, H = int(sys.argv[1]), 500, int(sys.argv[3]), 64
y = rng.randint(0, 2, size=(N, J)).astype(np.uint8); y[rng.rand(N, J) < 0.2] = 255
eng = IrtEngine(torch.from_numpy(y).cuda(), model="irt_2pl", D=D, amortized=True, H=H, seed=23)
eng.unconstrained("b").copy_(torch.from_numpy(0.5 * rng.randn(1, J)).float())
eng.loss_and_grads()
torch.cuda.synchronize()
fw = eng.last["fw"]
np.savez(sys.argv[2], loss=eng.G[eng.n_params:eng.n_params + 1].cpu().numpy(), g=eng.G[:eng.n_params].cpu().numpy(),
         **{k: fw[k][:N * n].cpu().numpy() for k, n in (("x", D), ("eps", D), ("h", H), ("ldT", D), ("ent", 1), ("hT", H),
                                                         ("epsT", D))})
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for mode in ("1", "0"):
        env = dict(os.environ, VX_FWD_RING=mode, VX_FORCE_GENERIC="0")
        out = str(tmp_path / ("ring_%s.npz" % mode))
        p = subprocess.run([sys.executable, "-c", code, str(N), out, str(D)], env=env, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        res[mode] = dict(np.load(out))
    assert set(res["1"]) == set(res["0"]) and {"x", "eps", "h", "hT", "loss", "g"} <= set(res["1"])
    for k in res["1"]:
        a, b = res["1"][k], res["0"][k]
        assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), k


@pytest.mark.parametrize("N,J,K,miss,B,amort", [
    (500, 30, 8, 0.0, None, False),                  # config 5 shape (PaHoDina), scaled down
    (333, 100, 5, 0.2, 77, False),                   # reference defaults q_size=5, item_size=100 (vi.py:125,138)
    (200, 70, 9, 0.1, None, False),                  # 512 patterns: 8 per lane
    (150, 40, 10, 0.0, 50, True),                    # 1024 patterns, amortized guide
    (100, 9, 1, 0.3, None, True),
    (333, 32, 5, 0.2, 77, False),                    # 5 <= K <= 8, J <= 32: the MFMA kernel (k_hodina_m.hip), one pattern tile
    (260, 17, 6, 0.3, None, True),                   # ... two tiles, amortized guide
    (131, 25, 7, 0.1, None, False),                  # ... four tiles; large lambda: clamped prior probabilities
    (900, 30, 6, 0.15, None, False),                 # ... 29 waves, most on the no-clamp form, a few with a clamped prior
    (2048, 30, 8, 0.0, None, False),                 # ... eight tiles, every wave full
])
def test_hodina_step_vs_oracle(N, J, K, miss, B, amort):
    from vipsy_amd.engine import HoDinaEngine, ENC_KEYS
    rng = np.random.RandomState(N + J + K)
    q = (rng.rand(K, J) < 0.4).astype(np.float32)
    q[rng.randint(0, K, size=J), np.arange(J)] = 1.0
    y = rng.randint(0, 2, size=(N, J)).astype(np.uint8)
    y[rng.rand(N, J) < miss] = 255
    eng = HoDinaEngine(torch.from_numpy(y).to(_dev()), q, amortized=amort, H=64, seed=9)
    lam_sc = 4.0 if K == 7 else 1.0
    eng.unconstrained("lam0").copy_(torch.from_numpy(lam_sc * 0.5 * rng.randn(1, K)).float())
    eng.unconstrained("lam1").copy_(torch.from_numpy(0.4 * rng.randn(1, K)).float())
    eng.unconstrained("g").add_(torch.from_numpy(0.5 * rng.randn(1, J)).float().to(_dev()))
    eng.unconstrained("s").add_(torch.from_numpy(0.5 * rng.randn(1, J)).float().to(_dev()))
    if not amort:
        eng.PP.copy_(torch.from_numpy(np.concatenate([rng.randn(N), 0.3 * rng.randn(N)])).float())
    idx = np.arange(N) if B is None else np.sort(rng.permutation(N)[:B])
    rows = None if B is None else torch.from_numpy(idx).to(_dev())
    eps = vo.philox_normals(9, 0, 0, idx, 1)
    eng.loss_and_grads(rows, len(idx))
    torch.cuda.synchronize()
    spec = {"family": "hodina", "K": K, "N": N, "amortized": amort, "q": q}
    params = {n: eng.unconstrained(n).cpu().numpy().astype(np.float64) for n in eng.all_names()}
    loss_o, g_o = vo.loss_and_grads(spec, params, y, [idx], [eps])
    assert float(eng.G[eng.n_params].item()) == pytest.approx(loss_o, rel=5e-5)
    errs = {}
    for name, go in g_o.items():
        gh = eng.unconstrained(name, eng.GP if (eng.per_person and name in eng.pp_off) else eng.G).cpu().numpy()
        sc = max(1e-6, float(np.abs(go).max()))
        errs[name] = float(np.abs(gh - go).max() / sc)
    print("hodina gradient errors (of the tensor's max), N=%d J=%d K=%d amort=%s: %s" % (N, J, K, amort, errs))
    assert max(errs.values()) < HODINA_TOL, errs


HODINA_TOL = 3e-5        # (measured: <= 5.2e-6, theta_local of the K = 8 case)


@pytest.mark.parametrize("N,J,model,miss,B", [
    (700, 500, "irt_2pl", 0.59, 100),                # Irt2PLMissing.test_ai (test.py:311-327), J scaled
    (300, 100, "irt_4pl", 0.0, None),                # Irt4PL.test_ai
    (129, 37, "irt_1pl", 0.2, 64),
    (320, 500, "irt_2pl", 0.3, None),                # N % 16 == 0, full batch: item-major responses in the fc1 gradient
    (1040, 516, "irt_3pl", 0.1, None),               # ... two 512-item groups, ragged last person tile
    (4500, 500, "irt_2pl", 0.9, None),               # >= 4 096 persons: fc1 from the fp16-pair images (k_norm_enc_fwd_h), rows by DMA
    (4300, 260, "irt_4pl", 0.2, 4200),               # ... gathered rows, a ragged last tile
    (4128, 504, "irt_1pl", 0.5, None),               # ... rows staged word by word (J / 4 even)
    (4501, 500, "irt_2pl", 0.9, None),               # no multiple of 4: the encoder over 4 504 persons, three of them phantoms
    (4503, 500, "irt_2pl", 0.2, None),               # (engine.py, D = 1 branch); observed-cell lists / the dense step kernel
    (4500, 499, "irt_2pl", 0.9, None),               # an item count that is no multiple of 4: one phantom item (absent / zero);
    (4301, 257, "irt_4pl", 0.2, 4200),               # ... observed-cell lists, gathered rows, a small shape
    (320, 37, "irt_2pl", 0.3, None),
])
def test_irt1d_amortized_step_vs_oracle(N, J, model, miss, B, H=64):
    from vipsy_amd.engine import IrtEngine
    rng = np.random.RandomState(N + J)
    y = rng.randint(0, 2, size=(N, J)).astype(np.uint8)
    y[rng.rand(N, J) < miss] = 255
    eng = IrtEngine(torch.from_numpy(y).to(_dev()), model=model, D=1, amortized=True, H=H, seed=5)
    eng.unconstrained("b").copy_(torch.from_numpy(0.7 * rng.randn(1, J)).float())
    if model != "irt_1pl":
        eng.unconstrained("a").copy_(torch.from_numpy(0.5 + 2 * rng.rand(1, J)).float())
    idx = np.arange(N) if B is None else np.sort(rng.permutation(N)[:B])
    rows = None if B is None else torch.from_numpy(idx).to(_dev())
    eps = vo.philox_normals(5, 0, 0, idx, 1)
    eng.loss_and_grads(rows, len(idx))
    torch.cuda.synchronize()
    spec = {"family": "irt", "model": model, "D": 1, "Dc": 1.0, "N": N, "amortized": True, "share_cov": False,
            "a_free": None}
    params = {n: eng.unconstrained(n).cpu().numpy().astype(np.float64) for n in eng.names()}
    loss_o, g_o = vo.loss_and_grads(spec, params, y, [idx], [eps])
    assert float(eng.G[eng.n_params].item()) == pytest.approx(loss_o, rel=3e-5)
    for name, go in g_o.items():
        gh = eng.unconstrained(name, eng.G).cpu().numpy()
        sc = max(1e-6, float(np.abs(go).max()))
        assert np.abs(gh - go).max() / sc < GRAD_TOL, (name, np.abs(gh - go).max() / sc)


@pytest.mark.parametrize("N,J,D,share,B", [(300, 64, 20, False, None), (257, 40, 100, False, 50), (200, 33, 9, True, 64)])
def test_mvn_bbvi_step_vs_oracle(N, J, D, share, B):
    """VIRT with x_feature > 1: per-person / shared Cholesky rows (vi.py:706-723)."""
    from vipsy_amd.engine import IrtEngine
    rng = np.random.RandomState(N + D)
    y = rng.randint(0, 2, size=(N, J)).astype(np.uint8)
    y[rng.rand(N, J) < 0.15] = 255
    eng = IrtEngine(torch.from_numpy(y).to(_dev()), model="irt_2pl", D=D, share_cov=share, seed=13)
    eng.unconstrained("b").copy_(torch.from_numpy(0.5 * rng.randn(1, J)).float())
    eng.unconstrained("x_local").copy_(torch.from_numpy(0.5 * rng.randn(N, D)).float())
    eng.unconstrained("x_scale").copy_(torch.from_numpy(0.2 * rng.randn(*eng.unconstrained("x_scale").shape)).float())
    idx = np.arange(N) if B is None else np.sort(rng.permutation(N)[:B])
    rows = None if B is None else torch.from_numpy(idx).to(_dev())
    eng.loss_and_grads(rows, len(idx))
    torch.cuda.synchronize()
    eps = eng.last["fw"]["eps"][:len(idx) * D].reshape(len(idx), D).cpu().numpy()
    np.testing.assert_allclose(eps, vo.philox_normals(13, 0, 0, idx, D), atol=2e-5)
    spec = {"family": "irt", "model": "irt_2pl", "D": D, "Dc": 1.0, "N": N, "amortized": False, "share_cov": share,
            "a_free": vo.default_a_free(D, J)}
    params = {n: eng.unconstrained(n).cpu().numpy().astype(np.float64) for n in eng.all_names()}
    loss_o, g_o = vo.loss_and_grads(spec, params, y, [idx], [eps])
    assert float(eng.G[eng.n_params].item()) == pytest.approx(loss_o, rel=3e-5)
    for name, go in g_o.items():
        gh = eng.unconstrained(name, eng.GP if (eng.per_person and name in eng.pp_off) else eng.G).cpu().numpy()
        if name == "a":
            gh = gh * spec["a_free"]
        sc = max(1e-6, float(np.abs(go).max()))
        assert np.abs(gh - go).max() / sc < GRAD_TOL, (name, np.abs(gh - go).max() / sc)


@pytest.mark.parametrize("N,J,K,cdm,miss,B", [
    (500, 30, 8, "dina", 0.0, None),                 # cfg5 shape without the higher-order layer
    (333, 70, 5, "dino", 0.2, 100),
    (129, 9, 10, "dina", 0.1, None),                 # K = 10: 16 patterns per lane
    (64, 130, 3, "dino", 0.0, None),                 # J > 128: three item slots per lane
])
def test_ccdm_step_vs_oracle(N, J, K, cdm, miss, B):
    """VCCDM (vi.py:819-865): enumerated DINA / DINO, uniform pattern prior, empty guide."""
    from vipsy_amd.engine import CcdmEngine
    rng = np.random.RandomState(N + J + K)
    q = (rng.rand(K, J) < 0.4).astype(np.float32)
    q[rng.randint(0, K, size=J), np.arange(J)] = 1.0          # every item needs at least one attribute
    y = rng.randint(0, 2, size=(N, J)).astype(np.uint8)
    y[rng.rand(N, J) < miss] = 255
    eng = CcdmEngine(torch.from_numpy(y).to(_dev()), q, cdm=cdm, seed=3)
    eng.unconstrained("g").add_(torch.from_numpy(0.5 * rng.randn(1, J)).float().to(_dev()))
    eng.unconstrained("s").add_(torch.from_numpy(0.5 * rng.randn(1, J)).float().to(_dev()))
    idx = np.arange(N) if B is None else np.sort(rng.permutation(N)[:B])
    rows = None if B is None else torch.from_numpy(idx).to(_dev())
    eng.loss_and_grads(rows, len(idx))
    torch.cuda.synchronize()
    spec = {"family": "ccdm", "cdm": cdm, "K": K, "N": N, "amortized": False, "q": q}
    params = {n: eng.unconstrained(n).cpu().numpy().astype(np.float64) for n in eng.names()}
    loss_o, g_o = vo.loss_and_grads(spec, params, y, [idx], [None])
    assert float(eng.G[eng.n_params].item()) == pytest.approx(loss_o, rel=2e-5)
    for name, go in g_o.items():
        gh = eng.unconstrained(name, eng.G).cpu().numpy()
        sc = max(1e-6, float(np.abs(go).max()))
        assert np.abs(gh - go).max() / sc < GRAD_TOL, (name, np.abs(gh - go).max() / sc)


@pytest.mark.parametrize("amortized,D,model,J,steps", [(True, 4, "irt_2pl", 24, 120), (False, 1, "irt_2pl", 24, 120),
                                                       (False, 1, "irt_4pl", 24, 120),
                                                       (True, 100, "irt_2pl", 500, 24)])     # the headline's model (f16x2 guide + likelihood)
def test_trained_item_parameters_within_1e3_of_cpu_reference(amortized, D, model, J, steps):
    """north_star: "item-parameter RMSE within 1e-3 of CPU reference" on identical simulated responses: Adam steps on the
    HIP path and on the oracle (float64) with the same Philox draws; compare the recovered a, b(, c, d).  The last case is the
    headline's shape (J = 500, D = 100: the f16x2 guide and likelihood kernels) with slopes of 0.05, so that no cell sits on
    the clamp of the Bernoulli log-probability (see test_headline_large_batch_kernels_vs_oracle)."""
    from vipsy_amd.engine import IrtEngine, LrSpec, ENC_KEYS
    N, H = (256 if D >= 64 else 512), 64
    a_level = 0.05 if D >= 64 else 1.0
    rng = np.random.RandomState(99 + D)
    x = rng.randn(N, D)
    af = vo.default_a_free(D, J)
    a_true = a_level * rng.uniform(0.5, 2.0, size=(D, J)) * (1.0 if af is None else af)
    b_true = rng.randn(1, J)
    pz = 1 / (1 + np.exp(-(x @ a_true + b_true)))
    if model == "irt_4pl":
        pz = 0.15 + (0.9 - 0.15) * pz
    y = (rng.rand(N, J) < pz).astype(np.uint8)
    y[rng.rand(N, J) < 0.1] = 255
    eng = IrtEngine(torch.from_numpy(y).to(_dev()), model=model, D=D, amortized=amortized, H=H, seed=17)
    spec = {"family": "irt", "model": model, "D": D, "Dc": 1.0, "N": N, "amortized": amortized, "share_cov": False,
            "a_free": vo.default_a_free(D, J)}
    enc0 = {k: eng.unconstrained("encoder$$$" + k).cpu().numpy().astype(np.float64) for k in ENC_KEYS} if amortized else None
    params = vo.init_irt_params(spec, J, np.float64, encoder=enc0)
    if a_level != 1.0:
        eng.unconstrained("a").mul_(a_level)
        params["a"] = params["a"] * a_level

    def lr_fn(module, name):
        return {"lr": (2e-2 if a_level == 1.0 else 2e-3) if name in ("a", "b", "c", "d") else (1e-3 if amortized else 2e-2)}
    lrs = LrSpec(lr_fn)
    adam = vo.Adam(lr_fn)
    idx = np.arange(N)
    for t in range(steps):
        eps = vo.philox_normals(17, t, 0, idx, D)
        _, g = vo.loss_and_grads(spec, params, y, [idx], [eps])
        adam.step(params, g)
        eng.step(lrs)
    torch.cuda.synchronize()
    for name in ("a", "b") + (("c", "d") if model == "irt_4pl" else ()):
        ph = eng.param(name).double().cpu().numpy()
        po = vo.constrained(name, params[name])
        rmse = float(np.sqrt(np.mean((ph - po) ** 2)))
        assert rmse < 1e-3 * a_level, (name, rmse)
        assert float(np.abs(po - (a_level if name == "a" else 0.0)).max()) > 0.05 * a_level     # the parameters did move


def _oracle_grads_chunked(params, y, eps, D, chunk=2048):
    """loss and every gradient of the amortized multivariate full-batch step from explicit parameters, float64, in person
    chunks (the sums over the persons of a full batch are additive: plate scale 1)."""
    N, J = y.shape
    loss, grads = 0.0, None
    for lo in range(0, N, chunk):
        hi = min(N, lo + chunk)
        spec = {"family": "irt", "model": "irt_2pl", "D": D, "Dc": 1.0, "N": hi - lo, "amortized": True, "share_cov": False,
                "a_free": vo.default_a_free(D, J)}
        l, g = vo.loss_and_grads(spec, params, y[lo:hi], [np.arange(hi - lo)], [eps[lo:hi].astype(np.float64)])
        loss += l
        grads = g if grads is None else {k: grads[k] + g[k] for k in g}
    return loss, grads


def test_trajectory_through_the_judged_kernels_within_1e3_of_cpu_reference():
    """north_star's accuracy clause ("item-parameter RMSE within 1e-3 of CPU reference") THROUGH THE
    KERNELS THE BENCH TIMES -- the large-batch forms k_mvn_enc_fwd_b2, k_irt_lik_h + k_lik_reduce_parts, k_mvn_enc_bwd_h_b2,
    k_mvn_enc_bwd_w_b, k_fc1_bwd_c (N = 33 024 = 129 workgroups of 256 persons, J = 500, D = 100, H = 64) -- from the reference's
    own initial slopes (a = ones + the zero pattern, vi.py:567-572; b = 0, vi.py:577), the bench's optimiser settings
    (test.py:345-350: lr 1e-2 for a, b, 1e-3 for the encoder), ten Adam steps on the HIP path and on the float64 oracle with the
    same Philox draws.  Nothing is marked missing here: cells within float32 rounding of the Bernoulli clamp (|z| = 15.9424,
    where the reference's gradient jumps from -+1 to 0) fall on either side in ANY float32 evaluation, the reference's own
    included; the test COUNTS the cells whose float32 logit (from the HIP path's x) and float64 logit lie on different sides
    and reports them -- Adam's normalisation keeps their effect on the parameters orders of magnitude under the tolerance."""
    from vipsy_amd.engine import IrtEngine, LrSpec, ENC_KEYS
    N, J, D, H, steps = 33024, 500, 100, 64, 10
    y, _, rng = _random_problem(N, J, D, H, "irt_2pl", 0.1, seed=4242)
    eng = IrtEngine(torch.from_numpy(y).to(_dev()), model="irt_2pl", D=D, amortized=True, H=H, seed=23)
    spec = {"family": "irt", "model": "irt_2pl", "D": D, "Dc": 1.0, "N": N, "amortized": True, "share_cov": False,
            "a_free": vo.default_a_free(D, J)}
    enc0 = {k: eng.unconstrained("encoder$$$" + k).cpu().numpy().astype(np.float64) for k in ENC_KEYS}
    params = vo.init_irt_params(spec, J, np.float64, encoder=enc0)
    assert np.array_equal(params["a"], eng.unconstrained("a").double().cpu().numpy())       # the reference's initial slopes
    a_init, b_init = params["a"].copy(), params["b"].copy()

    def lr_fn(module, name):
        return {"lr": 1e-2 if name in ("a", "b") else 1e-3}
    lrs, adam, idx = LrSpec(lr_fn), vo.Adam(lr_fn), np.arange(N)
    crossed, beyond, losses = [], [], []
    for t in range(steps):
        eps = vo.philox_normals(23, t, 0, idx, D)
        # the logits both sides see at this step's parameters: float64 from the oracle's latents, float32 from the HIP path's
        a32 = eng.unconstrained("a").cpu().numpy().copy()
        b32 = eng.unconstrained("b").cpu().numpy().copy()
        z_o = _oracle_latents_chunked(params, y, eps, D) @ params["a"] + params["b"]
        loss_o, g = _oracle_grads_chunked(params, y, eps, D)
        adam.step(params, g)
        loss_h = eng.step(lrs)
        torch.cuda.synchronize()
        x_h = eng.last["fw"]["x"][:N * D].reshape(N, D).cpu().numpy()
        z_h = x_h @ a32 + b32
        obs = y != 255
        crossed.append(int((((np.abs(z_h) > Z_CLAMP) != (np.abs(z_o) > Z_CLAMP)) & obs).sum()))
        beyond.append(float(((np.abs(z_o) > Z_CLAMP) & obs).mean()))
        losses.append((float(loss_h), loss_o))
        assert float(loss_h) == pytest.approx(loss_o, rel=1e-4), (t, losses)
    if t >= 1:
        assert eng._graph is not None and eng._graph["graph"] is not None                   # the shard-sized step replayed its graph
    print("trajectory through the judged kernels: cells on different sides of the clamp per step %s (of %d observed; "
          "%.1f %% beyond the clamp at step 0, %.1f %% at step %d)" % (crossed, int((y != 255).sum()), 100 * beyond[0],
                                                                       100 * beyond[-1], steps - 1))
    assert beyond[0] > 0.05                                     # the bench's regime
    rm = {}
    for name, init in (("a", a_init), ("b", b_init)):
        ph = eng.param(name).double().cpu().numpy()
        po = vo.constrained(name, params[name])
        rm[name] = float(np.sqrt(np.mean((ph - po) ** 2)))
        assert float(np.abs(po - init).max()) > 0.05                                         # the parameters did move
    print("trajectory through the judged kernels: RMSE(a) %.3g, RMSE(b) %.3g after %d steps" % (rm["a"], rm["b"], steps))
    assert rm["a"] < 1e-3 and rm["b"] < 1e-3, rm
    for k in ENC_KEYS:                                          # the encoder rode along: within the same tolerance of its own scale
        ph = eng.unconstrained("encoder$$$" + k).double().cpu().numpy()
        po = params["encoder$$$" + k]
        assert float(np.sqrt(np.mean((ph - po.reshape(ph.shape)) ** 2))) < 1e-3 * max(1.0, float(np.abs(po).max())), k


@pytest.mark.parametrize("miss,model", [(0.0, "irt_4pl"), (0.9, "irt_2pl")])
def test_captured_step_equals_eager_step(miss, model):
    """The D = 1 full-batch step replayed from a HIP graph (step counters in device memory) against the same step
    launched kernel by kernel: same bits, across a scheduler milestone (which captures again)."""
    from vipsy_amd.engine import IrtEngine, LrSpec
    rng = np.random.RandomState(3)
    N, J = 3000, 40
    y = rng.randint(0, 2, size=(N, J)).astype(np.uint8)
    y[rng.rand(N, J) < miss] = 255
    out = []
    for graph in (True, False):
        eng = IrtEngine(torch.from_numpy(y).to(_dev()), model=model, D=1, seed=11)
        eng.use_graph = graph
        lrs = LrSpec(lambda m, p: {"lr": 1e-2 if p in ("a", "b") else 1e-3}, milestones=(3,), gamma=0.5)
        losses = []
        for _ in range(7):
            losses.append(eng.step(lrs))                  # slots of the loss ring: seven steps, seven distinct values kept
            lrs.scheduler_step()
        torch.cuda.synchronize()
        assert eng.t == 7
        assert (getattr(eng, "_graph", None) or {}).get("graph") is not None if graph else getattr(eng, "_graph", None) is None
        out.append((torch.stack(losses).cpu().numpy(), eng.P.cpu().numpy().copy(), eng.PP.cpu().numpy().copy()))
    for u, v in zip(out[0], out[1]):
        assert np.array_equal(u, v)


@pytest.mark.parametrize("N,J,miss,B", [(5000, 500, 0.9, None), (3000, 64, 0.2, 100), (4500, 260, 0.5, None)])
def test_captured_amortized_1d_step_equals_eager_step(N, J, miss, B):
    """VaeIRT with one latent dimension (NormEncoder, vi.py:417-435, 677-684) replayed from its HIP graph -- full batches (the
    observed-cell lists at 90 % missing; the f16 forward from 4 096 persons on) and the reference's own subsample of 100
    (Irt2PLMissing.test_ai, test.py:311-327) -- against the same steps launched kernel by kernel: same bits."""
    from vipsy_amd.engine import IrtEngine, LrSpec
    rng = np.random.RandomState(N + J)
    y = rng.randint(0, 2, size=(N, J)).astype(np.uint8)
    y[rng.rand(N, J) < miss] = 255
    n_steps = 9
    rows_all = [None if B is None else torch.from_numpy(rng.choice(N, size=B, replace=False).astype(np.int64)) for _ in range(n_steps)]
    out = []
    for mode in ("graph", "eager", "steps"):
        eng = IrtEngine(torch.from_numpy(y).to(_dev()), model="irt_2pl", D=1, amortized=True, H=64, seed=11)
        eng.use_graph = mode != "eager"
        lrs = LrSpec(lambda m, p: {"lr": 1e-2 if p in ("a", "b") else 1e-3}, milestones=(4,), gamma=0.5)
        if mode == "steps":
            losses = eng.steps(lrs, rows_all, b_global=B, scheduler=True)
        else:
            losses = []
            for t in range(n_steps):
                losses.append(eng.step(lrs, rows=rows_all[t], b_global=B))
                lrs.scheduler_step()
        torch.cuda.synchronize()
        assert eng.t == n_steps
        assert ((getattr(eng, "_graph", None) or {}).get("graph") is not None) == (mode != "eager")
        out.append((torch.stack(losses).cpu().numpy(), eng.P.cpu().numpy().copy()))
    assert np.isfinite(out[0][0]).all() and len(set(out[0][0].tolist())) == n_steps
    for other in out[1:]:
        for u, v in zip(out[0], other):
            assert np.array_equal(u, v)


@pytest.mark.parametrize("kind", ["cfa_bbvi_d2_rows", "bbvi_d3_share_full", "irt1d_full", "amortized_d100_rows", "hodina_full",
                                  "vaechodina_rows", "irt1d_rows"])
def test_captured_particles_equal_eager_particles(kind):
    """Trace_ELBO(num_particles = S) (the reference's CFA demo: 20 particles of 100 rows, test.py:420-430): the S passes -- every
    particle its own subsample and its own Philox stream --, their mean and the one optimiser step replayed as ONE HIP graph,
    against the host loop over the particles: same bits in every loss and every parameter, across a scheduler milestone."""
    from vipsy_amd.engine import IrtEngine, HoDinaEngine, LrSpec
    rng = np.random.RandomState(23)
    n_steps = 7
    lr_fn = lambda m, p: {"lr": 1e-2 if p in ("a", "b", "g", "s") else 1e-3}
    if kind == "cfa_bbvi_d2_rows":
        N, J, S, B = 900, 6, 5, 100
        y = rng.randint(0, 2, size=(N, J)).astype(np.uint8)
        a_free = np.array([[1, 1, 1, 0, 0, 0], [0, 0, 0, 1, 1, 1]], dtype=np.float32)
        mk = lambda: IrtEngine(torch.from_numpy(y).to(_dev()), model="irt_2pl", D=2, a_free=torch.from_numpy(a_free), a0=torch.from_numpy(a_free), seed=11)
    elif kind == "bbvi_d3_share_full":
        N, J, S, B = 400, 20, 3, None
        y = rng.randint(0, 2, size=(N, J)).astype(np.uint8)
        y[rng.rand(N, J) < 0.1] = 255
        mk = lambda: IrtEngine(torch.from_numpy(y).to(_dev()), model="irt_2pl", D=3, share_cov=True, seed=11)
    elif kind in ("irt1d_full", "irt1d_rows"):
        N, J, S, B = (3000, 40, 3, None) if kind == "irt1d_full" else (3000, 40, 2, 100)
        y = rng.randint(0, 2, size=(N, J)).astype(np.uint8)
        mk = lambda: IrtEngine(torch.from_numpy(y).to(_dev()), model="irt_3pl", D=1, seed=11)
    elif kind == "amortized_d100_rows":
        N, J, D, H, S, B = 3000, 500, 100, 64, 2, 100
        y, _, _ = _random_problem(N, J, D, H, "irt_2pl", 0.1, seed=N + 5)
        mk = lambda: IrtEngine(torch.from_numpy(y).to(_dev()), model="irt_2pl", D=D, amortized=True, H=H, seed=11)
    else:
        N, J, K, S, B = (2000, 30, 6, 2, None) if kind == "hodina_full" else (1000, 30, 5, 4, 20)   # test.py:598-607: 20 rows, 10 particles
        q = (rng.rand(K, J) < 0.4).astype(np.float32)
        q[rng.randint(0, K, size=J), np.arange(J)] = 1.0
        y = rng.randint(0, 2, size=(N, J)).astype(np.uint8)
        mk = lambda: HoDinaEngine(torch.from_numpy(y).to(_dev()), q, amortized=kind != "hodina_full", H=64, seed=11)
    rows_all = [None if B is None else [torch.from_numpy(rng.choice(N, size=B, replace=False).astype(np.int64)) for _ in range(S)]
                for _ in range(n_steps)]
    out = []
    for graph in (True, False):
        eng = mk()
        eng.use_graph = graph
        lrs = LrSpec(lr_fn, milestones=(4,), gamma=0.5)
        losses = []
        for t in range(n_steps):
            losses.append(eng.step(lrs, rows=rows_all[t], b_global=B, num_particles=S))
            lrs.scheduler_step()
        torch.cuda.synchronize()
        assert eng.t == n_steps
        st = getattr(eng, "_graph", None) or {}
        assert (st.get("graph") is not None) == graph
        out.append((torch.stack(losses).cpu().numpy(), eng.P.cpu().numpy().copy(),
                    eng.PP.cpu().numpy().copy() if eng.per_person else None))
    assert np.isfinite(out[0][0]).all() and len(set(out[0][0].tolist())) == n_steps
    for u, v in zip(out[0], out[1]):
        assert (u is None and v is None) or np.array_equal(u, v)


@pytest.mark.parametrize("K,amort", [(8, False), (6, True), (3, False)])
def test_captured_hodina_step_equals_eager_step(K, amort):
    """The enumerated HO-DINA step (k_hodina_m for 5 <= K <= 8, k_hodina otherwise; either guide) replayed from its HIP graph
    -- the Philox step and Adam's t from the device counter that the loss sum advances -- against the same steps launched
    kernel by kernel: same bits, across a scheduler milestone, also four steps a replay."""
    from vipsy_amd.engine import HoDinaEngine, LrSpec
    rng = np.random.RandomState(17)
    N, J = 3000, 30
    q = (rng.rand(K, J) < 0.4).astype(np.float32)
    q[rng.randint(0, K, size=J), np.arange(J)] = 1.0
    y = rng.randint(0, 2, size=(N, J)).astype(np.uint8)
    y[rng.rand(N, J) < 0.1] = 255
    out = []
    for mode in ("graph", "eager", "steps"):
        eng = HoDinaEngine(torch.from_numpy(y).to(_dev()), q, amortized=amort, H=64, seed=11)
        eng.use_graph = mode != "eager"
        lrs = LrSpec(lambda m, p: {"lr": 1e-2 if p in ("g", "s") else 1e-3}, milestones=(5,), gamma=0.5)
        if mode == "steps":
            losses = eng.steps(lrs, [None] * 11, scheduler=True)
        else:
            losses = []
            for _ in range(11):
                losses.append(eng.step(lrs))
                lrs.scheduler_step()
        torch.cuda.synchronize()
        assert eng.t == 11
        assert ((getattr(eng, "_graph", None) or {}).get("graph") is not None) == (mode != "eager")
        out.append((torch.stack(losses).cpu().numpy(), eng.P.cpu().numpy().copy(),
                    eng.PP.cpu().numpy().copy() if eng.per_person else None))
    assert np.isfinite(out[0][0]).all() and len(set(out[0][0].tolist())) == 11
    for other in out[1:]:
        for u, v in zip(out[0], other):
            assert (u is None and v is None) or np.array_equal(u, v)


@pytest.mark.parametrize("amort,baseline,B", [(False, "avg", 100), (True, "avg", 100), (False, "none", None), (True, "none", None)])
def test_captured_cdm_sf_step_equals_eager_step(amort, baseline, B):
    """VCDM / VaeCDM (the score-function CDMs, vi.py:733-816; test.py:522-549 draws 100 or 1000 rows a step) replayed from HIP
    graphs -- the Bernoulli draws' Philox step and Adam's t from the device counter, the per-person guide's gather / scatter
    inside the capture, the decaying-average baseline updated by the kernel -- against the same steps launched kernel by
    kernel: same bits, across a scheduler milestone, also four steps a replay."""
    from vipsy_amd.engine import CdmSfEngine, LrSpec
    rng = np.random.RandomState(29)
    N, J, K = 3000, 30, 5
    q = (rng.rand(K, J) < 0.4).astype(np.float32)
    q[rng.randint(0, K, size=J), np.arange(J)] = 1.0
    y = rng.randint(0, 2, size=(N, J)).astype(np.uint8)
    draws = np.random.RandomState(5)
    n_steps = 19
    rows_all = [None if B is None else torch.from_numpy(np.sort(draws.choice(N, size=B, replace=False)).astype(np.int64))
                for _ in range(n_steps)]
    out = []
    for mode in ("graph", "eager", "steps"):
        eng = CdmSfEngine(torch.from_numpy(y).to(_dev()), q, amortized=amort, H=64, seed=11, baseline=baseline)
        eng.use_graph = mode != "eager"
        lrs = LrSpec(lambda m, p: {"lr": 1e-2}, milestones=(5,), gamma=0.5)
        if mode == "steps":
            losses = eng.steps(lrs, rows_all, b_global=B, scheduler=True)[-8:]
        else:
            losses = []
            for t in range(n_steps):
                losses.append(eng.step(lrs, rows=rows_all[t], b_global=B).clone())
                lrs.scheduler_step()
            losses = losses[-8:]
        torch.cuda.synchronize()
        assert eng.t == n_steps
        assert ((getattr(eng, "_graph", None) or {}).get("graph") is not None) == (mode != "eager")
        out.append((torch.stack(losses).cpu().numpy(), eng.P.cpu().numpy().copy(),
                    eng.PP.cpu().numpy().copy() if eng.per_person else None,
                    eng.base.cpu().numpy().copy() if eng.base is not None else None))
    assert np.isfinite(out[0][0]).all()
    for other in out[1:]:
        for u, v in zip(out[0], other):
            assert (u is None and v is None) or np.array_equal(u, v)


@pytest.mark.parametrize("cdm,B,vae", [("dina", None, False), ("dino", 100, False), ("dina", 100, True), ("dino", None, True)])
def test_captured_ccdm_step_equals_eager_step(cdm, B, vae):
    """VCCDM's step (the pattern-enumerated DINA / DINO with an empty guide, vi.py:819-865; test.py:560,585,624 draws 100-1500
    rows a step) replayed from its HIP graph -- Adam's t from the device counter the loss sum advances, host-drawn rows staged
    per replay -- against the same steps launched kernel by kernel: same bits, across a scheduler milestone, also four steps a
    replay."""
    from vipsy_amd.engine import CcdmEngine, VaeCcdmEngine, LrSpec
    rng = np.random.RandomState(23)
    N, J, K = 3000, 30, 5
    q = (rng.rand(K, J) < 0.4).astype(np.float32)
    q[rng.randint(0, K, size=J), np.arange(J)] = 1.0
    y = rng.randint(0, 2, size=(N, J)).astype(np.uint8)
    y[rng.rand(N, J) < 0.1] = 255
    draws = np.random.RandomState(5)
    n_steps = 19
    rows_all = [None if B is None else torch.from_numpy(draws.choice(N, size=B, replace=False).astype(np.int64)) for _ in range(n_steps)]
    out = []
    for mode in ("graph", "eager", "steps"):
        eng = (VaeCcdmEngine(torch.from_numpy(y).to(_dev()), q, cdm=cdm, H=64, seed=11) if vae     # (VaeCCDM: the same, with its
               else CcdmEngine(torch.from_numpy(y).to(_dev()), q, cdm=cdm, seed=11))                 # SoftmaxEncoder prior)
        eng.use_graph = mode != "eager"
        lrs = LrSpec(lambda m, p: {"lr": 1e-2}, milestones=(5,), gamma=0.5)
        if mode == "steps":
            losses = eng.steps(lrs, rows_all, b_global=B, scheduler=True)
        else:
            losses = []
            for t in range(n_steps):
                losses.append(eng.step(lrs, rows=rows_all[t], b_global=B))
                lrs.scheduler_step()
        torch.cuda.synchronize()
        assert eng.t == n_steps
        assert ((getattr(eng, "_graph", None) or {}).get("graph") is not None) == (mode != "eager")
        out.append((torch.stack([l.clone() for l in losses]).cpu().numpy() if mode != "steps" else torch.stack(losses[-8:]).cpu().numpy(),
                    eng.P.cpu().numpy().copy()))
    assert np.isfinite(out[0][0]).all()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    assert np.array_equal(out[2][0], out[1][0][-8:]) and np.array_equal(out[2][1], out[1][1])


@pytest.mark.parametrize("N,B", [
    (2048, None),        # the headline's model, small-batch kernels (SPLIT forward / hidden gradient), full batch
    (33024, None),       # ... the large-batch kernels of the judged step, second-stream tails included in the capture
    (5000, 100),         # the reference's own usage: subsample_size = 100 (test.py:338), host-drawn rows staged per replay
    (5000, 50),          # ... a subsample that is no multiple of 4: drawn two phantom rows longer (IrtEngine._pad_batch)
])
def test_captured_amortized_step_equals_eager_step(N, B):
    """The amortized D = 100 step (VaeIRT, vi.py:673-693) replayed from its HIP graph -- Philox step and Adam's t read from
    the device counter that the loss reduction advances, a subsample's rows copied into the fixed buffer the captured
    kernels read -- against the same steps launched kernel by kernel: same bits in every loss and every parameter, across a
    scheduler milestone (which captures again)."""
    from vipsy_amd.engine import IrtEngine, LrSpec
    J, D, H = 500, 100, 64
    y, _, _ = _random_problem(N, J, D, H, "irt_2pl", 0.1, seed=N + 1)
    draws = np.random.RandomState(5)
    n_steps = 7 if B is None else 19                           # (a subsample: more steps than the pinned row ring has slots)
    rows_all = [None if B is None else torch.from_numpy(draws.choice(N, size=B, replace=False).astype(np.int64)) for _ in range(n_steps)]
    out = []
    for graph in (True, False):
        eng = IrtEngine(torch.from_numpy(y).to(_dev()), model="irt_2pl", D=D, amortized=True, H=H, seed=11)
        eng.use_graph = graph
        lrs = LrSpec(lambda m, p: {"lr": 1e-2 if p in ("a", "b") else 1e-3}, milestones=(4,), gamma=0.5)
        losses = []
        for t in range(n_steps):
            losses.append(eng.step(lrs, rows=rows_all[t], b_global=B))       # host indices, as the fit loop hands them over
            lrs.scheduler_step()
        torch.cuda.synchronize()
        assert eng.t == n_steps
        st = getattr(eng, "_graph", None) or {}
        assert (st.get("graph") is not None) == graph
        if graph and B is not None:
            # the replay fetches its draw from the pinned host ring (vx_irt_cfg.rows_ring): no copy in front of it
            assert st.get("ring") is not None and st["ring"].is_pinned() and eng.rows_ring_slots < n_steps
        out.append((torch.stack(losses).cpu().numpy(), eng.P.cpu().numpy().copy()))
    assert np.isfinite(out[0][0]).all() and len(set(out[0][0].tolist())) == n_steps
    for u, v in zip(out[0], out[1]):
        assert np.array_equal(u, v)


@pytest.mark.parametrize("miss,model,graph", [(0.0, "irt_4pl", True), (0.0, "irt_2pl", False), (0.9, "irt_2pl", True), (0.9, "irt_3pl", False)])
def test_optimiser_in_the_steps_last_launch_equals_separate_launches(miss, model, graph):
    """One rank, D = 1 per-person guide: the slab sum and Adam in ONE launch (vx_irt1d_grad_adam / vx_irt1d_sparse_grad_adam,
    k_reduce_adam) against k_reduce_wide + k_adam2 (IrtEngine.fuse_tail = False): same losses, same replicated and
    per-person parameters and Adam moments, bit for bit -- eager and replayed, dense and on the observed-cell lists, across a
    scheduler milestone."""
    from vipsy_amd.engine import IrtEngine, LrSpec
    rng = np.random.RandomState(5)
    N, J = 70000 if miss == 0.0 and graph else 3000, 40       # (70 000 persons: more than 512 slabs -> the eight-column form)
    y = rng.randint(0, 2, size=(N, J)).astype(np.uint8)
    y[rng.rand(N, J) < miss] = 255
    out = []
    for fuse in (True, False):
        eng = IrtEngine(torch.from_numpy(y).to(_dev()), model=model, D=1, seed=11)
        eng.use_graph, eng.fuse_tail = graph, fuse
        lrs = LrSpec(lambda m, p: {"lr": 1e-2 if p in ("a", "b") else 1e-3}, milestones=(3,), gamma=0.5)
        losses = []
        for _ in range(7):
            losses.append(eng.step(lrs))
            lrs.scheduler_step()
        torch.cuda.synchronize()
        assert eng.t == 7
        out.append([torch.stack(losses).cpu().numpy()] + [t.cpu().numpy().copy() for t in (eng.P, eng.PP, eng.M, eng.V, eng.MP, eng.VP, eng.G[:eng.n_params])])
    assert np.isfinite(out[0][0]).all() and len(set(out[0][0].tolist())) == 7
    for u, v in zip(out[0], out[1]):
        assert np.array_equal(u, v)


@pytest.mark.parametrize("kind", ["irt1d_full", "amortized_rows", "amortized_full"])
def test_steps_replayed_four_at_a_time_equal_single_steps(kind):
    """IrtEngine.steps (what the fit loop calls): graph_steps consecutive steps replayed from ONE graph -- the Philox step and
    Adam's t from the device counter, a subsample's rows from the pinned ring -- against step() called in a loop: same bits in
    every loss and every parameter, across a scheduler milestone (the steps around it fall back to single replays) and with a
    remainder that is not a multiple of graph_steps."""
    from vipsy_amd.engine import IrtEngine, LrSpec
    rng = np.random.RandomState(11)
    n_steps = 27
    if kind == "irt1d_full":
        N, J = 3000, 40
        y = rng.randint(0, 2, size=(N, J)).astype(np.uint8)
        y[rng.rand(N, J) < 0.1] = 255
        mk = lambda: IrtEngine(torch.from_numpy(y).to(_dev()), model="irt_4pl", D=1, seed=11)
        rows_all, B = [None] * n_steps, None
    else:
        N, J, D, H = 5000 if kind == "amortized_rows" else 2048, 500, 100, 64
        y, _, _ = _random_problem(N, J, D, H, "irt_2pl", 0.1, seed=N + 3)
        mk = lambda: IrtEngine(torch.from_numpy(y).to(_dev()), model="irt_2pl", D=D, amortized=True, H=H, seed=11)
        B = 100 if kind == "amortized_rows" else None
        rows_all = [None if B is None else torch.from_numpy(rng.choice(N, size=B, replace=False).astype(np.int64)) for _ in range(n_steps)]
    out = []
    for multi in (True, False):
        eng = mk()
        lrs = LrSpec(lambda m, p: {"lr": 1e-2 if p in ("a", "b") else 1e-3}, milestones=(13,), gamma=0.5)
        if multi:
            assert eng.graph_steps == 4
            losses = eng.steps(lrs, rows_all, b_global=B, scheduler=True)
            assert (eng._graph or {}).get("multi") is not None          # the K-step graph was captured and replayed
        else:
            losses = []
            for t in range(n_steps):
                losses.append(eng.step(lrs, rows=rows_all[t], b_global=B))
                lrs.scheduler_step()
        torch.cuda.synchronize()
        assert eng.t == n_steps and lrs.epoch == n_steps
        out.append((torch.stack(losses).cpu().numpy(), eng.P.cpu().numpy().copy(), eng.PP.cpu().numpy().copy() if eng.per_person else None))
    assert np.isfinite(out[0][0]).all() and len(set(out[0][0].tolist())) == n_steps
    for u, v in zip(out[0], out[1]):
        assert (u is None and v is None) or np.array_equal(u, v)


def test_failed_capture_of_a_sharded_step_degrades_to_the_eager_step(monkeypatch):
    """A sharded step is captured WITHOUT its collective (two replays around an eager all-reduce).  If that capture fails
    beside a live communicator, the rank must warn and carry on kernel by kernel with the same results -- not end the job
    (VX_GRAPH_STRICT=1 keeps it an error).  The failure is injected: the process group is a one-rank gloo group, the graph
    object raises when the capture begins."""
    import torch.distributed as dist
    from vipsy_amd.engine import IrtEngine, LrSpec
    if not dist.is_initialized():
        dist.init_process_group(backend="gloo", init_method="tcp://127.0.0.1:29731", rank=0, world_size=1)
    N, J, D, H = 2048, 500, 100, 64
    y, _, _ = _random_problem(N, J, D, H, "irt_2pl", 0.1, seed=77)
    lrs = LrSpec(lambda m, p: {"lr": 1e-2 if p in ("a", "b") else 1e-3})
    ref = IrtEngine(torch.from_numpy(y).to(_dev()), model="irt_2pl", D=D, amortized=True, H=H, seed=11)
    ref.use_graph = False
    want = [float(ref.step(lrs)) for _ in range(4)]

    class Broken(torch.cuda.CUDAGraph):
        def capture_begin(self, *a, **k):
            raise RuntimeError("injected: capture refused")
    eng = IrtEngine(torch.from_numpy(y).to(_dev()), model="irt_2pl", D=D, amortized=True, H=H, seed=11, group=dist.group.WORLD)
    monkeypatch.setattr(torch.cuda, "CUDAGraph", Broken)
    with pytest.warns(UserWarning, match="graph capture of the sharded step failed"):
        got = [float(eng.step(lrs)) for _ in range(4)]
    monkeypatch.undo()
    torch.cuda.synchronize()
    assert eng.graph_fallback and eng.use_graph is False
    assert got == want
    np.testing.assert_array_equal(eng.P.cpu().numpy(), ref.P.cpu().numpy())
    # strict mode: the same failure is an error
    eng2 = IrtEngine(torch.from_numpy(y).to(_dev()), model="irt_2pl", D=D, amortized=True, H=H, seed=11, group=dist.group.WORLD)
    monkeypatch.setattr(torch.cuda, "CUDAGraph", Broken)
    monkeypatch.setenv("VX_GRAPH_STRICT", "1")
    eng2.step(lrs)                                             # the first step of a form is eager
    with pytest.raises(RuntimeError, match="injected"):
        eng2.step(lrs)
    monkeypatch.undo()
    dist.destroy_process_group()


def test_failed_capture_of_four_steps_degrades_to_single_steps(monkeypatch):
    """`steps()` records graph_steps steps as ONE graph once the single-step graph of the form exists.  If that
    second capture fails (a private pool that does not fit, a driver refusing the K-fold graph) nothing is half-done -- no
    collective was being recorded -- so the engine warns, keeps its single-step graph and takes the steps one by one with the
    same results; VX_GRAPH_STRICT=1 keeps it an error.  The failure is injected into the capture of the K-step graph only."""
    from vipsy_amd.engine import IrtEngine, LrSpec
    rng = np.random.RandomState(5)
    y = rng.randint(0, 2, size=(3000, 40)).astype(np.uint8)
    lrs = LrSpec(1e-2)
    ref = IrtEngine(torch.from_numpy(y).to(_dev()), model="irt_2pl", D=1, seed=3)
    want = [float(v) for v in ref.steps(lrs, [None] * 12)]
    eng = IrtEngine(torch.from_numpy(y).to(_dev()), model="irt_2pl", D=1, seed=3)
    got = [float(v) for v in eng.steps(lrs, [None] * 2)]              # eager, then the single-step capture
    real = torch.cuda.CUDAGraph

    class Broken(real):
        def capture_begin(self, *a, **k):
            raise RuntimeError("injected: K-step capture refused")
    monkeypatch.setattr(torch.cuda, "CUDAGraph", Broken)
    with pytest.warns(UserWarning, match="as one graph failed"):
        got += [float(v) for v in eng.steps(lrs, [None] * 6)]
    monkeypatch.undo()
    got += [float(v) for v in eng.steps(lrs, [None] * 4)]             # and stays on single steps afterwards
    torch.cuda.synchronize()
    assert eng.graph_steps == 1 and got == want
    np.testing.assert_array_equal(eng.P.cpu().numpy(), ref.P.cpu().numpy())
    eng2 = IrtEngine(torch.from_numpy(y).to(_dev()), model="irt_2pl", D=1, seed=3)
    eng2.steps(lrs, [None] * 2)
    monkeypatch.setattr(torch.cuda, "CUDAGraph", Broken)
    monkeypatch.setenv("VX_GRAPH_STRICT", "1")
    with pytest.raises(RuntimeError, match="injected"):
        eng2.steps(lrs, [None] * 4)
    monkeypatch.undo()


@pytest.mark.parametrize("H", [96, 128])
def test_irt1d_amortized_wide_hidden_layer(H):
    """NormEncoder with hidden_dim > 64 (vi.py:417-435 takes any width)."""
    test_irt1d_amortized_step_vs_oracle(300, 100, "irt_2pl", 0.2, None, H=H)


def test_small_batch_forward_is_bit_reproducible():
    """Small batches share the head tiles of a person tile among the four waves of a workgroup (k_mvn_enc_fwd_b<SPLIT>);
    a k that straddles two ranges gets two partial sums by LDS float adds -- two addends commute, so the step must be
    bit-identical from run to run."""
    from vipsy_amd.engine import IrtEngine
    rng = np.random.RandomState(8)
    N, J, D = 333, 500, 100
    y = rng.randint(0, 2, size=(N, J)).astype(np.uint8)
    y[rng.rand(N, J) < 0.1] = 255
    outs = []
    for _ in range(3):
        eng = IrtEngine(torch.from_numpy(y).to(_dev()), model="irt_2pl", D=D, amortized=True, H=64, seed=3)
        eng.loss_and_grads()
        torch.cuda.synchronize()
        outs.append((eng.G.cpu().numpy().copy(), eng.last["fw"]["x"][:N * D].cpu().numpy().copy()))
    for g, x in outs[1:]:
        assert np.array_equal(g, outs[0][0]) and np.array_equal(x, outs[0][1])


@pytest.mark.parametrize("amort,model,B,baseline", [(False, "irt_2pl", None, "none"), (False, "irt_4pl", 100, "avg"),
                                                     (True, "irt_2pl", None, "avg"), (True, "irt_3pl", 77, "none")])
def test_irt_score_function_step_vs_oracle(amort, model, B, baseline):
    """estimator='score' (north_star's REINFORCE mode for the IRT guide): the HIP step against the
    oracle's score mode on the same Philox draws -- per-person (BBVI) and amortized Normal guide, full batch and subsample,
    with the per-person decaying-average baseline carried over two steps."""
    from vipsy_amd.engine import IrtEngine
    N, J = 640, 60
    rng = np.random.RandomState(N + J + (1 if amort else 0))
    y = rng.randint(0, 2, size=(N, J)).astype(np.uint8)
    y[rng.rand(N, J) < 0.15] = 255
    eng = IrtEngine(torch.from_numpy(y).to(_dev()), model=model, D=1, amortized=amort, H=64, seed=5, estimator="score",
                    baseline=baseline, baseline_beta=0.8)
    eng.unconstrained("b").copy_(torch.from_numpy(0.7 * rng.randn(1, J)).float())
    eng.unconstrained("a").copy_(torch.from_numpy(0.5 + rng.rand(1, J)).float())
    if not amort:
        eng.PP.copy_(torch.from_numpy(np.concatenate([0.5 * rng.randn(N), -0.3 + 0.2 * rng.randn(N)])).float())
    spec = {"family": "irt", "model": model, "D": 1, "Dc": 1.0, "N": N, "amortized": amort, "share_cov": False, "a_free": None,
            "estimator": "score"}
    base = np.zeros(N)
    for t in range(2):
        idx = np.arange(N) if B is None else np.sort(rng.permutation(N)[:B])
        rows = None if B is None else torch.from_numpy(idx).to(_dev())
        eng.t = t                                                    # the Philox step of this pass
        eps = vo.philox_normals(5, t, 0, idx, 1)
        eng.loss_and_grads(rows, len(idx))
        torch.cuda.synchronize()
        names = eng.names() + ([] if amort else ["x_local", "x_scale"])
        params = {n: eng.unconstrained(n).cpu().numpy().astype(np.float64) for n in names}
        bl = base[idx] if baseline == "avg" else None
        loss_o, g_o, log_r = vo.irt_particle(spec, params, y, idx, eps, baseline=bl, want_log_r=True)
        if baseline == "avg":
            base[idx] = 0.8 * base[idx] + 0.2 * log_r
            np.testing.assert_allclose(eng.base.cpu().numpy(), base, rtol=2e-5, atol=2e-4)
        assert float(eng.G[eng.n_params].item()) == pytest.approx(loss_o, rel=3e-5)
        for name, go in g_o.items():
            gh = eng.unconstrained(name, eng.GP if (eng.per_person and name in eng.pp_off) else eng.G).cpu().numpy()
            sc = max(1e-6, float(np.abs(go).max()))
            assert np.abs(gh - go).max() / sc < GRAD_TOL, (name, t, np.abs(gh - go).max() / sc)


def test_latent_outside_the_f16_image_range_through_the_step():
    """A guide whose location head puts one latent dimension at 700 (> 511.75, the range of the fp16 x image the forward writes
    for k_irt_lik_h) with items that discriminate weakly on it (slope 0.01: the true z moves by 7, a saturated image would
    make it 5.1): the forward raises the image's overflow word, k_irt_lik_h stands down and the bf16x3 kernel does the step --
    loss and every gradient as the oracle's."""
    from vipsy_amd.engine import IrtEngine
    N, J, D, H = 320, 500, 100, 64
    y, enc, rng = _random_problem(N, J, D, H, "irt_2pl", 0.1, seed=77)
    enc["fc21.bias"][3] = 700.0
    eng = IrtEngine(torch.from_numpy(y).to(_dev()), model="irt_2pl", D=D, amortized=True, H=H,
                    encoder_init={k: v.astype(np.float32) for k, v in enc.items()}, seed=11)
    a0 = 0.05 * eng.unconstrained("a") * torch.from_numpy(1 + 0.3 * rng.randn(D, J)).float().to(_dev())
    a0[3, :] = 0.01
    eng.unconstrained("a").copy_(a0 * eng.unconstrained("a", eng.free))
    eng.unconstrained("b").copy_(torch.from_numpy(0.5 * rng.randn(1, J)).float())
    eng.loss_and_grads(None, N)
    torch.cuda.synchronize()
    fw = eng.last["fw"]
    assert float(fw["x"][:N * D].reshape(N, D)[:, 3].min()) > 600.0
    eps = fw["eps"][:N * D].reshape(N, D).cpu().numpy()
    spec = {"family": "irt", "model": "irt_2pl", "D": D, "Dc": 1.0, "N": N, "amortized": True, "share_cov": False,
            "a_free": vo.default_a_free(D, J)}
    params = {n: eng.unconstrained(n).cpu().numpy().astype(np.float64) for n in eng.names()}
    loss_o, g_o = vo.loss_and_grads(spec, params, y, [np.arange(N)], [eps])
    assert float(eng.G[eng.n_params].item()) == pytest.approx(loss_o, rel=3e-5)
    for name, go in g_o.items():
        gh = eng.unconstrained(name, eng.G).cpu().numpy() * eng.unconstrained(name, eng.free).cpu().numpy()
        sc = max(1e-6, float(np.abs(go).max()))
        assert np.abs(gh - go).max() / sc < GRAD_TOL, (name, np.abs(gh - go).max() / sc)


def test_f16x2_likelihood_entry_with_extreme_latents():
    """vx_irt_lik_grad called directly, without the forward's operand image, on latents that leave the fixed range of
    k_irt_lik_h's x operand (|x| 2^7 beyond the largest fp16): the image made inside the call raises its overflow word and
    the bf16x3 kernel, launched behind the fp16 one, does the work -- every output as the oracle's."""
    from vipsy_amd.engine import HipBackend
    N, J, D, scale = 192, 260, 100, 3.0
    rng = np.random.RandomState(5)
    y = rng.randint(0, 2, size=(N, J)).astype(np.uint8)
    y[rng.rand(N, J) < 0.1] = 255
    x = rng.randn(N, D).astype(np.float32)
    x[5, 3], x[17, 40], x[100, 99] = 900.0, -2000.0, 511.9
    a = (0.05 * (1 + 0.3 * rng.randn(D, J))).astype(np.float32)
    b = (0.5 * rng.randn(1, J)).astype(np.float32)
    be, dev = HipBackend(), _dev()
    cfg = be.cfg("irt_2pl", D, J, 64, 1.0, scale, 1, 0, 0)
    t = lambda v: torch.from_numpy(np.ascontiguousarray(v)).to(dev)
    stride = (N + 63) // 64 * 64
    yT = torch.full((J + 1, stride), 254, dtype=torch.uint8, device=dev)
    yT[:J, :N] = t(y).t()
    gxT = torch.empty(D * N, dtype=torch.float32, device=dev)
    ll = torch.empty(N, dtype=torch.float32, device=dev)
    gitem = torch.zeros(D * J + 3 * J, dtype=torch.float32, device=dev)
    ws = torch.empty(be.lik_workspace(cfg, N), dtype=torch.float32, device=dev)
    be.lik_grad(cfg, t(y), None, N, t(x), t(a), t(b), None, None, None, ll, gitem, ws, gxT=gxT, yT=yT.contiguous())
    torch.cuda.synchronize()
    x64 = x.astype(np.float64)
    ll_o, g = vo.irt_loglik("irt_2pl", x64, a.astype(np.float64), b.astype(np.float64), None, None, 1.0, y)
    for name, got, want in (("ll", ll.cpu().numpy(), ll_o - 0.5 * (x64 ** 2).sum(1)),
                            ("gx", gxT.cpu().numpy().reshape(D, N).T, scale * (g["x"] - x64)),
                            ("ga", gitem[:D * J].cpu().numpy().reshape(D, J), -scale * g["a"]),
                            ("gb", gitem[D * J:D * J + J].cpu().numpy().reshape(1, J), -scale * g["b"])):
        assert np.isfinite(got).all(), name
        err = np.abs(got - want)
        w = np.unravel_index(err.argmax(), err.shape)
        assert err.max() / np.abs(want).max() < GRAD_TOL, (name, err.max() / np.abs(want).max(), w, got[w], want[w])


def test_f16x2_likelihood_entry_with_a_nan_latent():
    """A NaN (or infinite) latent must raise the f16 image's overflow word like any other value outside its range (ADVICE
    round 3: fmaxf dropped the NaN and the person was clamped to +-511.75, contaminating the item gradients with finite
    numbers): the bf16x3 kernel then runs and the NaN reaches that person's outputs and the item gradients, as the
    reference's float chain would have it (vi.py:41 on a NaN x)."""
    from vipsy_amd.engine import HipBackend
    N, J, D, scale = 128, 132, 100, 1.0
    rng = np.random.RandomState(6)
    y = rng.randint(0, 2, size=(N, J)).astype(np.uint8)
    x = rng.randn(N, D).astype(np.float32)
    x[7, 11] = np.nan
    a = (0.05 * (1 + 0.3 * rng.randn(D, J))).astype(np.float32)
    b = (0.5 * rng.randn(1, J)).astype(np.float32)
    be, dev = HipBackend(), _dev()
    cfg = be.cfg("irt_2pl", D, J, 64, 1.0, scale, 1, 0, 0)
    t = lambda v: torch.from_numpy(np.ascontiguousarray(v)).to(dev)
    stride = (N + 63) // 64 * 64
    yT = torch.full((J + 1, stride), 254, dtype=torch.uint8, device=dev)
    yT[:J, :N] = t(y).t()
    gxT = torch.empty(D * N, dtype=torch.float32, device=dev)
    ll = torch.empty(N, dtype=torch.float32, device=dev)
    gitem = torch.zeros(D * J + 3 * J, dtype=torch.float32, device=dev)
    ws = torch.empty(be.lik_workspace(cfg, N), dtype=torch.float32, device=dev)
    be.lik_grad(cfg, t(y), None, N, t(x), t(a), t(b), None, None, None, ll, gitem, ws, gxT=gxT, yT=yT.contiguous())
    torch.cuda.synchronize()
    ll_h = ll.cpu().numpy()
    assert np.isnan(ll_h[7]), "the NaN latent was clamped into a finite log-likelihood"
    others = np.ones(N, dtype=bool)
    others[7] = False
    assert np.isfinite(ll_h[others]).all()
    assert np.isnan(gitem[:D * J].cpu().numpy().reshape(D, J)[11]).any(), "item gradients silently finite under a NaN latent"


@pytest.mark.parametrize("guide,N,J,D,model,B,baseline", [
    ("amortized", 264, 36, 8, "irt_2pl", None, "none"),              # encoder heads: dimension-major backward kernels
    ("amortized", 640, 40, 12, "irt_3pl", 128, "avg"),               # ... subsample, decaying-average baseline by row
    ("amortized", 320, 500, 100, "irt_2pl", None, "avg"),            # the headline's guide shape
    ("shared", 300, 30, 3, "irt_2pl", None, "none"),                 # VIRT share_cov (vi.py:706-715)
    ("person", 257, 33, 4, "irt_4pl", 100, "avg"),                   # VIRT per-person Cholesky rows (vi.py:716-723)
])
def test_mvn_score_function_step_vs_oracle(guide, N, J, D, model, B, baseline):
    """estimator='score' for the multivariate Normal guides (x_feature > 1): w_i L_i^-T eps_i and the DIAG-row operand from
    k_mvn_score.hip through the unchanged guide-backward kernels, against the oracle's score mode (mvn_score_terms) on the
    same Philox draws; two steps so that the decaying-average baseline is carried over."""
    from vipsy_amd.engine import IrtEngine
    amort = guide == "amortized"
    rng = np.random.RandomState(N + J + D)
    if amort:
        y, enc, rng = _random_problem(N, J, D, 64, model, 0.1, seed=N + J + D)
        kw = {"encoder_init": {k: v.astype(np.float32) for k, v in enc.items()}}
    else:
        y = rng.randint(0, 2, size=(N, J)).astype(np.uint8)
        y[rng.rand(N, J) < 0.1] = 255
        kw = {"share_cov": guide == "shared"}
    eng = IrtEngine(torch.from_numpy(y).to(_dev()), model=model, D=D, amortized=amort, H=64, seed=5, estimator="score",
                    baseline=baseline, baseline_beta=0.8, **kw)
    eng.unconstrained("b").copy_(torch.from_numpy(0.5 * rng.randn(1, J)).float())
    a0 = eng.unconstrained("a") * torch.from_numpy(1 + 0.3 * rng.randn(D, J)).float().to(_dev())
    eng.unconstrained("a").copy_((0.05 if D >= 64 else 1.0) * a0 * eng.unconstrained("a", eng.free))
    if not amort:
        eng.unconstrained("x_local").copy_(torch.from_numpy(0.5 * rng.randn(N, D)).float())
        Ms = 0.3 * rng.randn(*((D, D) if guide == "shared" else (N, D, D)))
        eng.unconstrained("x_scale").copy_(torch.from_numpy(Ms).float())
    spec = {"family": "irt", "model": model, "D": D, "Dc": 1.0, "N": N, "amortized": amort, "share_cov": guide == "shared",
            "a_free": vo.default_a_free(D, J), "estimator": "score"}
    base = np.zeros(N)
    for t in range(2):
        idx = np.arange(N) if B is None else np.sort(rng.permutation(N)[:B])
        rows = None if B is None else torch.from_numpy(idx).to(_dev())
        eng.t = t
        eps = vo.philox_normals(5, t, 0, idx, D)
        eng.loss_and_grads(rows, len(idx))
        torch.cuda.synchronize()
        names = eng.names() + ([] if amort else ["x_local"] + ([] if guide == "shared" else ["x_scale"]))
        params = {n: eng.unconstrained(n).cpu().numpy().astype(np.float64) for n in dict.fromkeys(names)}
        bl = base[idx] if baseline == "avg" else None
        loss_o, g_o, log_r = vo.irt_particle(spec, params, y, idx, eps, baseline=bl, want_log_r=True)
        if baseline == "avg":
            base[idx] = 0.8 * base[idx] + 0.2 * log_r
            np.testing.assert_allclose(eng.base.cpu().numpy(), base, rtol=3e-5, atol=1e-3 * max(1.0, np.abs(log_r).max() * 1e-2))
        assert float(eng.G[eng.n_params].item()) == pytest.approx(loss_o, rel=3e-5)
        for name, go in g_o.items():
            src = eng.GP if (eng.per_person and name in eng.pp_off) else eng.G
            gh = eng.unconstrained(name, src).cpu().numpy()
            if name == "a":
                gh = gh * eng.unconstrained("a", eng.free).cpu().numpy()
            sc = max(1e-6, float(np.abs(go).max()))
            assert np.abs(gh.reshape(go.shape) - go).max() / sc < GRAD_TOL, (name, t, np.abs(gh.reshape(go.shape) - go).max() / sc)


@pytest.mark.parametrize("N,D,J", [(33024, 100, 500), (4104, 8, 36), (1000, 100, 40), (8200, 32, 40)])
def test_mvn_score_operands_mfma_kernel_vs_scalar_kernel_and_oracle(N, D, J):
    """k_mvn_score_b (u = L^-T eps by column tiles of the head GEMM on the fp16 MFMA, k_mvn_score_b.hip) against (1) the scalar
    kernel it replaces for these shapes (k_mvn_score_operands<0>, same engine state, the seam `score_mfma`) and (2) float64:
    L rebuilt from the heads (vi.py:448-455), u by a triangular solve, for a sample of the persons -- at the headline's shape
    with every CU busy (33 024 persons), at the smallest dimension the kernel takes, at a large one with a ragged last wave
    (1 000 = 31 waves + 8 persons) and at D = 32 (columns of one tile each)."""
    from vipsy_amd.engine import IrtEngine
    y, enc, rng = _random_problem(N, J, D, 64, "irt_2pl", 0.1, seed=N + D)
    eng = IrtEngine(torch.from_numpy(y).to(_dev()), model="irt_2pl", D=D, amortized=True, H=64, seed=7, estimator="score",
                    baseline="avg", baseline_beta=0.8, encoder_init={k: v.astype(np.float32) for k, v in enc.items()})
    a0 = eng.unconstrained("a") * torch.from_numpy(1 + 0.3 * rng.randn(D, J)).float().to(_dev())
    eng.unconstrained("a").copy_((0.05 if D >= 64 else 1.0) * a0 * eng.unconstrained("a", eng.free))
    out = {}
    for mfma in (True, False):
        eng.score_mfma = mfma
        eng.base.zero_()
        eng.loss_and_grads()
        torch.cuda.synchronize()
        gd_off = eng.be.mvn_enc_bwd_gd_offset(eng.be.cfg(eng.model, D, J, 64, 1.0, 1.0, 7, 0, 0), N)
        assert gd_off >= 0
        out[mfma] = (eng.last["gxT"][:N * D].reshape(D, N).cpu().numpy().copy(),
                     eng._ws["encb_ws"][gd_off:gd_off + N * D].reshape(D, N).cpu().numpy().copy(),
                     eng.unconstrained("encoder$$$fc22.weight", eng.G).cpu().numpy().copy(), eng.last_log_r.cpu().numpy().copy())
    for q, name in enumerate(("gxT", "gdT", "G_W22", "log_r")):
        sc = np.abs(out[False][q]).max()
        err = np.abs(out[True][q] - out[False][q]).max() / sc
        print("score operands, N = %d D = %d: %s MFMA against scalar kernel %.2e of the max" % (N, D, name, err))
        assert err < GRAD_TOL, (name, err)
    # float64 for a sample of the persons (the first, the last, random ones)
    idx = np.unique(np.concatenate([np.arange(40), np.arange(N - 40, N), rng.choice(N, 200, replace=False)]))
    fw = eng.last["fw"]
    params = {n: eng.unconstrained(n).cpu().numpy().astype(np.float64) for n in eng.names()}
    W = {k: params["encoder$$$" + k] for k in vo.ENC_KEYS}
    loc, raw, _ = vo.enc_forward(W, vo.enc_input(y[idx], np.float64))
    eps = fw["eps"][:N * D].reshape(N, D).cpu().numpy()[idx].astype(np.float64)
    r_, c_ = vo.tril_rows_cols(D)
    w = out[True][3][idx].astype(np.float64)                  # log_r - baseline (the baseline starts at zero)
    gx_o = np.empty((len(idx), D))
    gd_o = np.empty((len(idx), D))
    for n in range(len(idx)):
        M = np.zeros((D, D))
        M[r_, c_] = raw[n]
        L = np.tril(M, -1) + np.diag(np.exp(np.diag(M)))
        u = np.linalg.solve(L.T, eps[n])
        gx_o[n] = w[n] * u
        gd_o[n] = w[n] * (u * eps[n] * np.diag(L) - 1.0)
    for got, want, name in ((out[True][0][:, idx].T, gx_o, "gxT"), (out[True][1][:, idx].T, gd_o, "gdT")):
        err = np.abs(got - want).max() / np.abs(want).max()
        assert err < GRAD_TOL, (name, err)


def test_mvn_score_function_loo_baseline_through_step():
    """baseline='loo' for a multivariate guide (per-person Cholesky rows, D = 3) through IrtEngine.step: three particles share
    the batch, each with the leave-one-out mean of the others' log_r as its control variate (lr = 0 keeps the parameters)."""
    from vipsy_amd.engine import IrtEngine, LrSpec
    N, J, D, S = 256, 30, 3, 3
    rng = np.random.RandomState(21)
    y = rng.randint(0, 2, size=(N, J)).astype(np.uint8)
    eng = IrtEngine(torch.from_numpy(y).to(_dev()), model="irt_2pl", D=D, seed=9, estimator="score", baseline="loo")
    eng.unconstrained("x_local").copy_(torch.from_numpy(0.5 * rng.randn(N, D)).float())
    eng.unconstrained("x_scale").copy_(torch.from_numpy(0.3 * rng.randn(N, D, D)).float())
    spec = {"family": "irt", "model": "irt_2pl", "D": D, "Dc": 1.0, "N": N, "amortized": False, "share_cov": False,
            "a_free": vo.default_a_free(D, J), "estimator": "score"}
    params = {n: eng.unconstrained(n).cpu().numpy().astype(np.float64) for n in eng.names() + ["x_local", "x_scale"]}
    idx = np.arange(N)
    eps = [vo.philox_normals(9, 0, s, idx, D) for s in range(S)]        # the particle index is the Philox stream
    lrs = [vo.irt_particle(spec, params, y, idx, e, want_log_r=True)[2] for e in eps]
    outs = [vo.irt_particle(spec, params, y, idx, e, baseline=(sum(lrs) - lrs[s]) / (S - 1)) for s, e in enumerate(eps)]
    loss_h = float(eng.step(LrSpec(0.0), num_particles=S).item())
    torch.cuda.synchronize()
    assert loss_h == pytest.approx(np.mean([o[0] for o in outs]), rel=3e-5)
    for name in ("a", "b", "x_local", "x_scale"):
        go = np.mean([o[1][name] for o in outs], axis=0)
        gh = eng.unconstrained(name, eng.GP if name in eng.pp_off else eng.G).cpu().numpy().reshape(go.shape)
        if name == "a":
            gh = gh * eng.unconstrained("a", eng.free).cpu().numpy()
        sc = max(1e-6, float(np.abs(go).max()))
        assert np.abs(gh - go).max() / sc < GRAD_TOL, (name, np.abs(gh - go).max() / sc)


def test_irt_score_function_loo_baseline_through_step():
    """baseline='loo' through IrtEngine.step: three particles share the batch, each with the leave-one-out mean of the others'
    log_r as its control variate (lr = 0 keeps the parameters, so the averaged gradient can be checked)."""
    from vipsy_amd.engine import IrtEngine, LrSpec
    N, J, S = 512, 40, 3
    rng = np.random.RandomState(12)
    y = rng.randint(0, 2, size=(N, J)).astype(np.uint8)
    eng = IrtEngine(torch.from_numpy(y).to(_dev()), model="irt_2pl", D=1, seed=9, estimator="score", baseline="loo")
    eng.PP.copy_(torch.from_numpy(np.concatenate([0.5 * rng.randn(N), -0.3 + 0.2 * rng.randn(N)])).float())
    spec = {"family": "irt", "model": "irt_2pl", "D": 1, "Dc": 1.0, "N": N, "amortized": False, "share_cov": False, "a_free": None,
            "estimator": "score"}
    params = {n: eng.unconstrained(n).cpu().numpy().astype(np.float64) for n in eng.names() + ["x_local", "x_scale"]}
    idx = np.arange(N)
    eps = [vo.philox_normals(9, 0, s, idx, 1) for s in range(S)]        # the particle index is the Philox stream
    lrs = [vo.irt_particle(spec, params, y, idx, e, want_log_r=True)[2] for e in eps]
    outs = [vo.irt_particle(spec, params, y, idx, e, baseline=(sum(lrs) - lrs[s]) / (S - 1)) for s, e in enumerate(eps)]
    loss_h = float(eng.step(LrSpec(0.0), num_particles=S).item())
    torch.cuda.synchronize()
    assert loss_h == pytest.approx(np.mean([o[0] for o in outs]), rel=3e-5)
    for name in ("a", "b", "x_local", "x_scale"):
        go = np.mean([o[1][name] for o in outs], axis=0)
        gh = eng.unconstrained(name, eng.GP if name in eng.pp_off else eng.G).cpu().numpy()
        sc = max(1e-6, float(np.abs(go).max()))
        assert np.abs(gh - go).max() / sc < GRAD_TOL, (name, np.abs(gh - go).max() / sc)
