"""The oracle against the golden vectors produced by the reference's own model()/guide()/SVI.step
code (tests/golden/make_golden.py): loss, every gradient, every parameter after every step."""
import numpy as np
import pytest

from oracle import vi_oracle as vo
from tests import golden_util as gu


@pytest.mark.parametrize("tag", gu.elbo_cases())
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_oracle_matches_reference_steps(tag, dtype):
    spec, params, opt, y, steps, B = gu.build(tag, dtype)
    adam = vo.Adam(opt["lr"], milestones=opt["milestones"], gamma=opt["gamma"])
    rtol = 2e-4 if dtype == np.float32 else 5e-5      # golden itself is float32 torch arithmetic
    for t, rec in enumerate(steps):
        assert all(len(i) == B for i in rec["idx"])
        loss, grads = vo.loss_and_grads(spec, params, y, rec["idx"], rec["eps"])
        assert loss == pytest.approx(rec["loss"], rel=rtol), (tag, t)
        assert set(grads) == set(rec["grad"]), (tag, sorted(grads), sorted(rec["grad"]))
        for k, g in rec["grad"].items():
            scale = gu.grad_scale(rec, k)
            np.testing.assert_allclose(grads[k] / scale, g / scale, atol=5 * rtol, err_msg="%s step %d grad %s" % (tag, t, k))
        adam.step(params, grads)
        adam.scheduler_step()
        for k, p in rec["param"].items():
            ok = gu.adam_conditioned(steps, t, k) if k in rec["grad"] else np.ones(p.shape, bool)
            np.testing.assert_allclose(params[k][ok], p[ok], atol=2e-5 if dtype == np.float64 else 1e-4, rtol=1e-4,
                                       err_msg="%s step %d param %s" % (tag, t, k))
            params[k][~ok] = p[~ok].astype(params[k].dtype)      # follow the reference where its own step is rounding noise
