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

pytestmark = pytest.mark.gpu


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
