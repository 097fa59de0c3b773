"""Host-side logic of the engine that needs no GPU: the observed-cell lists of the D = 1 list kernel (their layout is an ABI
contract, include/vipsy_amd.h) and the O(B) subsample of small minibatches."""
import numpy as np
import torch

from tests.oracle_backend import OracleBackend
from vipsy_amd.engine import IrtEngine


def _decode_lists(sp, n, J):
    """Rebuild the response matrix from pidx / pent / glen exactly as k_irt1d_sp walks them."""
    pent = sp["pent"].cpu().numpy().view(np.uint16)            # [ng][Lq][64][4]
    pidx = sp["pidx"].cpu().numpy()
    glen = sp["glen"].cpu().numpy()
    ng, Lq = pent.shape[0], pent.shape[1]
    assert pent.shape == (ng, Lq, 64, 4) and pidx.shape == (ng * 64,) and glen.shape == (ng,)
    y = np.full((n, J), 255, dtype=np.uint8)
    seen = np.zeros(n, dtype=bool)
    for g in range(ng):
        for lane in range(64):
            i = pidx[g * 64 + lane]
            codes = pent[g, :, lane, :].reshape(-1)
            if i < 0:
                assert (codes == 0xFFFF).all()                  # an empty slot has an empty list
                continue
            assert not seen[i]
            seen[i] = True
            live = codes != 0xFFFF
            k = int(live.sum())
            assert live[:k].all() and k <= 4 * glen[g]          # the list is a prefix; the group's quad count covers it
            items = (codes[:k] & 0x7FFF).astype(int)
            assert (np.diff(items) > 0).all()                   # ascending items, each once
            y[i, items] = (codes[:k] >> 15).astype(np.uint8)
    assert seen.all()
    return y


def test_observed_cell_lists_encode_the_responses():
    rng = np.random.RandomState(0)
    for n, J, miss in ((300, 37, 0.9), (5000, 12, 0.6), (64, 500, 0.95), (65, 8, 0.999)):
        y = rng.randint(0, 2, size=(n, J)).astype(np.uint8)
        y[rng.rand(n, J) < miss] = 255
        eng = IrtEngine(torch.from_numpy(y), model="irt_2pl", D=1, backend=OracleBackend(), observed_lists=True)
        sp = eng._sparse_lists(None)
        assert sp is not None and sp["n_groups"] == (n + 63) // 64
        assert np.array_equal(_decode_lists(sp, n, J), y)
        # slots are sorted by list length inside windows of 4096: the 64 lists of a group end (almost) together
        cnt = (y != 255).sum(1)
        pidx = sp["pidx"].cpu().numpy()
        for w0 in range(0, len(pidx), 4096):
            c = np.array([cnt[i] if i >= 0 else -1 for i in pidx[w0:w0 + 4096]])
            assert (np.diff(c) <= 0).all()
    # mostly observed responses, or a minibatch: no lists
    y = rng.randint(0, 2, size=(100, 10)).astype(np.uint8)
    eng = IrtEngine(torch.from_numpy(y), model="irt_2pl", D=1, backend=OracleBackend(), observed_lists=True)
    assert eng._sparse_lists(None) is None
    assert eng._sparse_lists(torch.arange(10)) is None


def test_small_subsamples_are_distinct_rows_drawn_on_the_host():
    from vipsy_amd import vi
    y = torch.from_numpy(np.random.RandomState(1).randint(0, 2, size=(4000, 6)).astype(np.uint8))
    m = vi.VIRT(data=y, model="irt_2pl", subsample_size=100, backend=OracleBackend(), seed=5)
    draws = [m._subsample() for _ in range(50)]
    for idx, bg in draws:
        a = idx.cpu().numpy()
        assert bg == 100 and a.shape == (100,) and len(set(a.tolist())) == 100 and a.min() >= 0 and a.max() < 4000
    assert len({tuple(d[0].cpu().numpy().tolist()) for d in draws}) == 50            # fresh rows every step
    m2 = vi.VIRT(data=y, model="irt_2pl", subsample_size=100, backend=OracleBackend(), seed=5)
    assert np.array_equal(m2._subsample()[0].cpu().numpy(), draws[0][0].cpu().numpy())   # reproducible from the seed
    hits = np.bincount(np.concatenate([d[0].cpu().numpy() for d in draws]), minlength=4000)
    assert hits.max() <= 8                                     # 5000 draws over 4000 rows: no row is favoured


def test_captured_row_forms_are_bounded():
    """Every distinct (nb, b_global) subsample form would keep a captured graph for ever: beyond
    graph_max_row_forms the least recently used forms are dropped, the most recently used stay."""
    y = torch.from_numpy(np.random.RandomState(2).randint(0, 2, size=(64, 6)).astype(np.uint8))
    eng = IrtEngine(y, model="irt_2pl", D=1, backend=OracleBackend())
    eng._graphs, eng._graph = {}, None
    full = ("full", 64, 64)
    eng._graphs[full] = {"graph": object()}
    for nb in range(1, 21):
        eng._graphs[("rows", nb, 100)] = {"graph": object(), "rows": object()}
        eng._evict_graph_forms()
        if nb == 5:                                            # a form that is used again moves to the recent end
            eng._graphs[("rows", 1, 100)] = eng._graphs.pop(("rows", 1, 100))
    rows_forms = [m for m in eng._graphs if m[0] == "rows"]
    assert len(rows_forms) == eng.graph_max_row_forms == 8
    assert rows_forms == [("rows", nb, 100) for nb in range(13, 21)]
    assert full in eng._graphs                                 # the full-batch form is not a candidate


def test_loss_reaches_the_ring_without_an_optimiser_launch():
    """A step whose optimiser has nothing to launch still files its loss (the ring slot used to be written only
    as a side effect of the first Adam launch)."""
    from vipsy_amd.engine import LrSpec, LOSS_RING
    y = torch.from_numpy(np.random.RandomState(3).randint(0, 2, size=(32, 5)).astype(np.uint8))
    eng = IrtEngine(y, model="irt_2pl", D=1, backend=OracleBackend())
    eng.loss_and_grads(None, None, None, 0)
    want = float(eng.G[eng.n_params])
    eng.names = lambda: []                                     # no trainable replicated segment ...
    eng.per_person = False                                     # ... and no per-person one: apply_optim launches nothing
    eng.apply_optim(LrSpec(1e-2))
    assert float(eng.step_loss()) == want and float(eng.loss_ring[eng.t % LOSS_RING]) == want
