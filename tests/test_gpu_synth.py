"""On-device synthesis (vipsy_amd/csrc/k_synth.hip), the Random* generator classes (vi.py:120-412) and the generate ->
dump -> fit replication path (test.py:144-236)."""
import os

import numpy as np
import pytest
import torch

from oracle import vi_oracle as vo

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


@pytest.mark.parametrize("cls_name,kw", [("RandomIrt2PL", {}), ("RandomIrt4PL", {"x_feature": 3}),
                                         ("RandomMilIrt2PL", {"x_feature": 4}), ("RandomIrt1PL", {})])
def test_random_irt_classes_match_the_restated_draw_rule(cls_name, kw):
    from vipsy_amd import random_data as rd
    torch.manual_seed(5)
    ri = getattr(rd, cls_name)(sample_size=300, item_size=37, device=_dev(), seed=77, gid0=1 << 34, **kw)
    y = ri.y.cpu().numpy()
    assert y.dtype == np.uint8 and y.shape == (300, 37) and set(np.unique(y)) <= {0, 1}
    get = lambda k: getattr(ri, k).numpy().astype(np.float64) if hasattr(ri, k) else None        # noqa: E731
    gids = (1 << 34) + np.arange(300)
    yo, xo, P, u = vo.synth_irt(ri.name, 77, gids, get("a"), get("b"), get("c"), get("d"))
    np.testing.assert_allclose(ri.x.cpu().numpy(), xo, atol=2e-5)
    clear = np.abs(u - P) > 1e-5                       # a uniform that sits on the probability may round either way
    assert clear.mean() > 0.999 and np.array_equal(y[clear], yo[clear])
    if hasattr(ri, "a") and ri.x_feature > 1:
        for i in range(ri.x_feature):
            assert float(ri.a[i, 37 - i:].abs().sum()) == 0            # identification zeros (vi.py:257-258, 378-379)


def test_random_irt_1pl_with_a_latent_location_and_scale():
    """vi.py:202-233 with x_local / x_scale: the 1PL link has no slope to carry the scale, so the class draws through the 2PL
    link with a = x_scale; x = x_local + x_scale z, responses ~ Bernoulli(sigmoid(x + b))."""
    from vipsy_amd import random_data as rd
    torch.manual_seed(9)
    ri = rd.RandomIrt1PL(sample_size=400, item_size=21, device=_dev(), seed=78, x_local=0.3, x_scale=1.7)
    y = ri.y.cpu().numpy()
    gids = np.arange(400)
    b = ri.b.numpy().astype(np.float64)
    a_eff = 1.7 * np.ones((1, 21))
    yo, zo, P, u = vo.synth_irt("irt_2pl", 78, gids, a_eff, b + 0.3, None, None)
    np.testing.assert_allclose(ri.x.cpu().numpy(), 0.3 + 1.7 * zo, atol=5e-5)
    clear = np.abs(u - P) > 1e-5
    assert clear.mean() > 0.999 and np.array_equal(y[clear], yo[clear])


def test_random_cdm_classes_and_sharding_independence():
    from vipsy_amd import random_data as rd
    torch.manual_seed(6)
    r = rd.RandomHoDina(sample_size=5000, item_size=30, q_size=4, device=_dev(), seed=3)
    y, attr, th = r.y.cpu().numpy(), r.attr.cpu().numpy(), r.theta.cpu().numpy()
    assert y.shape == (5000, 30) and attr.shape == (5000, 4) and abs(th.mean()) < 0.06 and abs(th.std() - 1) < 0.05
    q, need = r.q.numpy(), (r.q.numpy() ** 2).sum(0)
    eta = (attr.astype(np.float64) @ q == need)
    P = np.where(eta, 1 - r.s.numpy(), r.g.numpy())
    assert abs(y.mean() - P.mean()) < 0.02
    # the same persons drawn as two shards: identical bytes (every draw is keyed by the global person id)
    r2 = rd.RandomHoDina(sample_size=2000, item_size=30, q_size=4, device=_dev(), seed=3, gid0=3000)
    for k in ("q", "g", "s", "lam0", "lam1"):
        setattr(r2, k, getattr(r, k))
    assert np.array_equal(r2.y.cpu().numpy(), y[3000:])
    rdn = rd.RandomDino(sample_size=2000, item_size=12, q_size=3, device=_dev(), seed=4)
    assert set(np.unique(rdn.y.cpu().numpy())) <= {0, 1}


def test_article_test_util_generates_dumps_fits(tmp_path):
    """test.py:144-201: generate, dump the text files the R script reads, fit, error metric; then two replications."""
    from vipsy_amd import harness, vi, random_data as rd
    out = harness.article_test_util(sample_size=1500, item_size=12, vi_class=vi.VIRT, random_class=rd.RandomIrt2PL,
                                    vi_class_kwargs={"subsample_size": 1500}, folder=str(tmp_path), seed=8,
                                    vi_fit_kwargs={"optim": vi.Adam({"lr": 5e-2}), "max_iter": 300})
    assert set(out) == {"a", "b"} and out["b"] < 0.25 and out["a"] < 0.45
    pre = os.path.join(str(tmp_path), "irt_2pl_sample_1500_item_12_dim_1")
    for suffix in ("_0.txt", "_a_0.txt", "_b_0.txt"):
        assert os.path.exists(pre + suffix)
    y, r = harness.load_case(str(tmp_path), "2pl", 1500, 12, 1, 0)          # the loader of test.py:18-67 reads them back
    assert y.shape == (1500, 12) and r.a.shape == (1, 12)
    s = harness.multiprocess_article_test_util(sample_size=800, item_size=10, vi_class=vi.VaeIRT,
                                               random_class=rd.RandomIrt2PL, try_count=2, folder=str(tmp_path), seed=20,
                                               vi_fit_kwargs={"max_iter": 50})
    assert set(s) == {"a", "b"} and all(len(v) == 2 for v in s.values())
