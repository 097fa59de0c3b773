"""Size-independent properties at the BASELINE.json full sizes (the oracle cannot run there):

* sharding additivity -- the gradient buffer + loss of one engine over N persons equals the sum over two engines that
  own the two halves (same parameters, seed, global N): exercises the global-person-id keyed RNG, the plate scale, the
  tail handling of every kernel and every fixed-order reduction at 1M persons;
* determinism -- a repeated step is bit-identical;
* the dimension-major copies written by the producers are exact transposes of the person-major ones.

All through the C ABI (engine -> ctypes).  Sizes: cfg3 (1M x 500 x 100 amortized 2PL), cfg4 (1M x 500 2PL, 90 %
missing), cfg5 (HO-DINA 1M x 30 x 8); ~2 GB of HBM, a few seconds."""
import numpy as np
import pytest
import torch

from oracle import vi_oracle as vo          # the checker (tests only)

pytestmark = pytest.mark.gpu

N_SAMPLE = 4096


def _sample_rows(N, seed):
    """4 096 persons of the 1M: random ones, the first 64, and the LAST 256 (the end of every buffer; 64-bit offsets of the
    5 050-row x 1M-person arrays only go wrong out there)."""
    rng = np.random.RandomState(seed)
    idx = np.concatenate([rng.choice(N - 320, N_SAMPLE - 320, replace=False) + 64, np.arange(64), np.arange(N - 256, N)])
    return np.sort(idx)


def _dev():
    return torch.device("cuda:0")


def _flat(eng):
    return eng.G[:eng.n_params + 1].double().cpu().numpy()


def _copy_params(dst, src):
    dst.P.copy_(src.P)


def test_cfg3_headline_sharding_additivity_and_determinism():
    from vipsy_amd import synth
    from vipsy_amd.engine import IrtEngine
    N, J, D, H = 1000000, 500, 100, 64
    a, b = synth.mirt_item_params(J, D, seed=20243)
    y = synth.simulate_responses(N, 0, {"a": a, "b": b}, "irt_2pl", _dev(), seed=20240)
    full = IrtEngine(y, model="irt_2pl", D=D, n_global=N, gid0=0, amortized=True, H=H, seed=1234)
    rng = np.random.RandomState(7)
    full.unconstrained("b").copy_(torch.from_numpy(0.3 * rng.randn(1, J)).float())
    full.loss_and_grads()
    torch.cuda.synchronize()
    g_full = _flat(full)
    assert np.isfinite(g_full).all()
    # dimension-major copies == transposes (hT / epsT from the forward, gxT from the likelihood)
    fw = full.last["fw"]
    if "epsT" in fw:
        eps = fw["eps"][:N * D].reshape(N, D)
        assert torch.equal(fw["epsT"][:N * D].reshape(D, N).t(), eps)
        assert torch.equal(fw["hT"][:N * H].reshape(H, N).t(), fw["h"][:N * H].reshape(N, H))
    # determinism
    full.loss_and_grads()
    torch.cuda.synchronize()
    assert np.array_equal(_flat(full), g_full)
    _check_cfg3_sample_against_oracle(full, y, N, J, D, H)
    # two shards of 500k persons
    acc = np.zeros_like(g_full)
    for s in range(2):
        lo, hi = s * (N // 2), (s + 1) * (N // 2)
        sh = IrtEngine(y[lo:hi], model="irt_2pl", D=D, n_global=N, gid0=lo, amortized=True, H=H, seed=1234)
        _copy_params(sh, full)
        sh.loss_and_grads()
        torch.cuda.synchronize()
        acc += _flat(sh)
        del sh
    scale = np.abs(g_full).max()
    assert np.abs(acc[:-1] - g_full[:-1]).max() <= 2e-4 * scale
    assert acc[-1] == pytest.approx(g_full[-1], rel=2e-5)


def test_cfg3_half_shard_replayed_steps_equal_eager_steps():
    """A 500 000-person shard of the headline (what each of two GPUs holds): seven steps replayed from the HIP graph against
    the same steps launched kernel by kernel -- same bits in every loss and every parameter, across a scheduler milestone.
    (The second-stream branches of the large-batch forward and backward are inside the capture; shards of this size replay
    since round 6, tools/step_events_cost.py.)"""
    from vipsy_amd import synth
    from vipsy_amd.engine import IrtEngine, LrSpec
    N, J, D, H = 500000, 500, 100, 64
    a, b = synth.mirt_item_params(J, D, seed=20243)
    y = synth.simulate_responses(N, 0, {"a": a, "b": b}, "irt_2pl", _dev(), seed=20241)
    out = []
    for graph in (True, False):
        eng = IrtEngine(y, model="irt_2pl", D=D, amortized=True, H=H, seed=11)
        eng.use_graph = graph
        assert eng._graphable() == graph
        lrs = LrSpec(lambda m, p: {"lr": 1e-2 if p in ("a", "b") else 1e-3}, milestones=(4,), gamma=0.5)
        losses = []
        for t in range(7):
            losses.append(eng.step(lrs))
            lrs.scheduler_step()
        torch.cuda.synchronize()
        st = getattr(eng, "_graph", None) or {}
        assert (st.get("graph") is not None) == graph
        out.append((torch.stack(losses).cpu().numpy(), eng.P.cpu().numpy().copy()))
        del eng
    assert np.isfinite(out[0][0]).all() and len(set(out[0][0].tolist())) == 7
    for u, v in zip(out[0], out[1]):
        assert np.array_equal(u, v)


def _check_cfg3_sample_against_oracle(eng, y, N, J, D, H):
    """The judged size against the oracle, on a sample: everything the step computes PER PERSON -- h, x, ent of the guide
    (vi.py:448-455, 686-693), the person's log-likelihood + prior term and d ELBO / d x (vi.py:32-41, 596-625) -- depends on
    that person's responses and draws alone, so 4 096 of the 1M persons cost a 4 096-person oracle call (float64, the kernel's
    own Philox draws, which are themselves compared with the oracle's generator).  Tolerances: those of
    tests/test_gpu_parity.py::test_headline_large_batch_kernels_vs_oracle (2e-5 of the range; 3e-5 of the tensor's max for
    the gradient rows)."""
    idx = _sample_rows(N, 11)
    it = torch.from_numpy(idx).to(y.device)
    fw = eng.last["fw"]
    ys = y[it].cpu().numpy()
    eps = fw["eps"][:N * D].reshape(N, D)[it].cpu().numpy()
    np.testing.assert_allclose(eps, vo.philox_normals(1234, 0, 0, idx, D), atol=2e-5)
    params = {n: eng.unconstrained(n).cpu().numpy().astype(np.float64) for n in eng.names()}
    W = {k: params["encoder$$$" + k] for k in vo.ENC_KEYS}
    loc, raw, cache = vo.enc_forward(W, vo.enc_input(ys, np.float64))
    ec = eps.astype(np.float64)
    x_o, col0, msum = loc.copy(), 0, np.zeros(len(idx))
    for k in range(D):                                  # row k of L (vi.py:452-454): raw[(k, 0..k-1)], exp on the diagonal
        x_o[:, k] += (raw[:, col0:col0 + k] * ec[:, :k]).sum(1) + np.exp(raw[:, col0 + k]) * ec[:, k]
        msum += raw[:, col0 + k]
        col0 += k + 1
    ent_o = 0.5 * (ec ** 2).sum(1) + msum
    ll_o, g = vo.irt_loglik("irt_2pl", x_o, params["a"], params["b"], None, None, 1.0, ys)
    ll_o = ll_o - 0.5 * (x_o ** 2).sum(1)               # + the N(0, I) prior on x without its constant (vi.py:607-613)
    gx_o = g["x"] - x_o                                 # plate scale 1 (full batch)
    x_h = fw["x"][:N * D].reshape(N, D)[it].cpu().numpy()
    h_h = fw["h"][:N * H].reshape(N, H)[it].cpu().numpy()
    np.testing.assert_allclose(h_h, cache[2], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(x_h, x_o, atol=2e-5 * max(1.0, np.abs(x_o).max()), rtol=1e-5)
    np.testing.assert_allclose(fw["ent"][:N][it].cpu().numpy(), ent_o, atol=2e-5 * max(1.0, np.abs(ent_o).max()), rtol=1e-5)
    # cells whose float32 logit and float64 logit lie on different sides of the Bernoulli clamp move a person's ll by up to
    # 1.2e-7 and a gx row by |a|: leave out the persons with such a cell (a dozen of 2 M cells at slopes of 1)
    z = x_o @ params["a"] + params["b"]
    zc = float(np.log((1.0 - vo.EPS32) / vo.EPS32))
    clear = ~((np.abs(np.abs(z) - zc) < 1e-3 * np.abs(z)) & (ys != 255)).any(1)
    assert clear.mean() > 0.9
    ll_h = eng.last["ll"][:N][it].cpu().numpy()
    np.testing.assert_allclose(ll_h[clear], ll_o[clear], rtol=3e-5, atol=3e-5 * np.abs(ll_o).max())
    gx_h = eng.last["gxT"][:N * D].reshape(D, N)[:, it].t().cpu().numpy()
    sc = np.abs(gx_o[clear]).max()
    err = np.abs(gx_h[clear] - gx_o[clear]).max() / sc
    print("cfg3 at 1M persons, %d sampled persons (%d clear of the clamp): gx rows within %.2e of the tensor's max" %
          (len(idx), int(clear.sum()), err))
    assert err < 3e-5, err


def test_cfg4_bbvi_missing90_sharding_additivity():
    from vipsy_amd import synth
    from vipsy_amd.engine import IrtEngine
    N, J = 1000000, 500
    items = synth.irt_item_params(J, "irt_2pl", seed=20242)
    y = synth.simulate_responses(N, 0, items, "irt_2pl", _dev(), seed=20240, missing=0.9)
    frac = float((y == 255).float().mean())
    assert 0.89 < frac < 0.91
    full = IrtEngine(y, model="irt_2pl", D=1, n_global=N, gid0=0, seed=1234)
    full.loss_and_grads()
    torch.cuda.synchronize()
    g_full = _flat(full)
    gp_full = full.GP.double().cpu().numpy()
    full.loss_and_grads()                                   # determinism: a repeated step is bit-identical
    torch.cuda.synchronize()
    assert np.array_equal(_flat(full), g_full) and np.array_equal(full.GP.double().cpu().numpy(), gp_full)
    # the judged size against the oracle, on a sample: the per-person gradient rows (d loss / d x_local, d x_scale;
    # vi.py:698-705) of 4 096 persons from a 4 096-person oracle call with the same parameters and the same Philox draws
    idx = _sample_rows(N, 12)
    it = torch.from_numpy(idx).to(_dev())
    spec = {"family": "irt", "model": "irt_2pl", "D": 1, "Dc": 1.0, "N": len(idx), "amortized": False, "share_cov": False,
            "a_free": None}
    params = {n: full.unconstrained(n).cpu().numpy().astype(np.float64) for n in full.names()}
    pp = full.PP.double().cpu().numpy()
    params["x_local"], params["x_scale"] = pp[:N][idx].reshape(-1, 1), pp[N:][idx].reshape(-1, 1)
    _, g_o = vo.loss_and_grads(spec, params, y[it].cpu().numpy(), [np.arange(len(idx))], [vo.philox_normals(1234, 0, 0, idx, 1)])
    for name, rows in (("x_local", gp_full[:N][idx]), ("x_scale", gp_full[N:][idx])):
        go = g_o[name].reshape(-1)
        err = np.abs(rows - go).max() / max(1e-6, np.abs(go).max())
        assert err < 3e-5, (name, err)
    acc = np.zeros_like(g_full)
    for s in range(2):
        lo, hi = s * (N // 2), (s + 1) * (N // 2)
        sh = IrtEngine(y[lo:hi], model="irt_2pl", D=1, n_global=N, gid0=lo, seed=1234)
        _copy_params(sh, full)
        sh.loss_and_grads()
        torch.cuda.synchronize()
        acc += _flat(sh)
        n = hi - lo                                         # per-person rows are owned by the shard: exact match
        gp = sh.GP.double().cpu().numpy()
        assert np.array_equal(gp[:n], gp_full[lo:hi]) and np.array_equal(gp[n:], gp_full[N + lo:N + hi])
        del sh
    assert np.abs(acc[:-1] - g_full[:-1]).max() <= 2e-4 * np.abs(g_full[:-1]).max()
    assert acc[-1] == pytest.approx(g_full[-1], rel=2e-5)


def test_cfg5_hodina_sharding_additivity():
    from vipsy_amd import synth
    from vipsy_amd.engine import HoDinaEngine
    N, J, K = 1000000, 30, 8
    prm = synth.hodina_params(J, K, seed=20245)
    y = synth.simulate_hodina(N, 0, prm, _dev(), seed=20240)
    full = HoDinaEngine(y, prm["q"], n_global=N, gid0=0, seed=1234)
    full.loss_and_grads()
    torch.cuda.synchronize()
    g_full = _flat(full)
    assert np.isfinite(g_full).all()
    full.loss_and_grads()                                   # determinism (fixed-point pattern table, per-wave reduce slots)
    torch.cuda.synchronize()
    assert np.array_equal(_flat(full), g_full)
    # the judged size against the oracle, on a sample: the per-person gradient rows (d loss / d theta_local, d theta_scale;
    # vi.py:925-934 through the enumerated model, vi.py:897-923) of 4 096 of the 1M persons
    idx = _sample_rows(N, 13)
    it = torch.from_numpy(idx).to(_dev())
    spec = {"family": "hodina", "K": K, "N": len(idx), "amortized": False, "q": prm["q"].cpu().numpy() if hasattr(prm["q"], "cpu")
            else np.asarray(prm["q"])}
    params = {n: full.unconstrained(n).cpu().numpy().astype(np.float64) for n in full.all_names() if n not in full.pp_off}
    pp = full.PP.double().cpu().numpy()
    gp = full.GP.double().cpu().numpy()
    params["theta_local"], params["theta_scale"] = pp[:N][idx].reshape(-1, 1), pp[N:][idx].reshape(-1, 1)
    _, g_o = vo.loss_and_grads(spec, params, y[it].cpu().numpy(), [np.arange(len(idx))], [vo.philox_normals(1234, 0, 0, idx, 1)])
    for name, rows in (("theta_local", gp[:N][idx]), ("theta_scale", gp[N:][idx])):
        go = g_o[name].reshape(-1)
        err = np.abs(rows - go).max() / max(1e-6, np.abs(go).max())
        assert err < 3e-5, (name, err)
    acc = np.zeros_like(g_full)
    for s in range(2):
        lo, hi = s * (N // 2), (s + 1) * (N // 2)
        sh = HoDinaEngine(y[lo:hi], prm["q"], n_global=N, gid0=lo, seed=1234)
        _copy_params(sh, full)
        sh.loss_and_grads()
        torch.cuda.synchronize()
        acc += _flat(sh)
        del sh
    assert np.abs(acc[:-1] - g_full[:-1]).max() <= 2e-4 * np.abs(g_full[:-1]).max()
    assert acc[-1] == pytest.approx(g_full[-1], rel=2e-5)


@pytest.mark.parametrize("case", ["cfg2_4pl_dense", "cfg4_dense_kernel", "virt_d8_share_cov", "generic_mvn"])
def test_repeated_step_is_bit_identical(case):
    """Every block reduction is a fixed-order tree or an integer (fixed-point) sum: no float atomics anywhere, so a repeated
    step reproduces every gradient bit for bit -- also on the dense D = 1 kernel (BASELINE config 2), the shared-covariance
    BBVI guide and the shape-generic amortized kernels (odd D, H = 40)."""
    from vipsy_amd import synth
    from vipsy_amd.engine import IrtEngine
    dev = _dev()
    if case == "cfg2_4pl_dense":
        items = synth.irt_item_params(100, "irt_4pl", seed=20242)
        y = synth.simulate_responses(100000, 0, items, "irt_4pl", dev, seed=20240)
        eng = IrtEngine(y, model="irt_4pl", D=1, seed=1234)
    elif case == "cfg4_dense_kernel":
        items = synth.irt_item_params(500, "irt_2pl", seed=20242)
        y = synth.simulate_responses(200000, 0, items, "irt_2pl", dev, seed=20240, missing=0.3)    # < 50 % missing: dense kernel
        eng = IrtEngine(y, model="irt_2pl", D=1, seed=1234)
    elif case == "virt_d8_share_cov":
        items = synth.irt_item_params(60, "irt_2pl", seed=20242, D=8)
        y = synth.simulate_responses(20000, 0, items, "irt_2pl", dev, seed=20240)
        eng = IrtEngine(y, model="irt_2pl", D=8, share_cov=True, seed=1234)
    else:
        a, b = synth.mirt_item_params(70, 7, seed=20243)
        y = synth.simulate_responses(5000, 0, {"a": a, "b": b}, "irt_3pl" if False else "irt_2pl", dev, seed=20240)
        eng = IrtEngine(y, model="irt_2pl", D=7, amortized=True, H=40, seed=1234)
    outs = []
    for _ in range(2):
        eng.loss_and_grads()
        torch.cuda.synchronize()
        outs.append((_flat(eng), eng.GP.double().cpu().numpy() if eng.per_person else None))
    assert np.isfinite(outs[0][0]).all()
    assert np.array_equal(outs[0][0], outs[1][0])
    if outs[0][1] is not None:
        assert np.array_equal(outs[0][1], outs[1][1])
