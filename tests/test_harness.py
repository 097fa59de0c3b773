"""Text interchange format and error summaries of the reference's replication study (test.py:18-141)."""
import os

import numpy as np
import pytest
import torch


def test_response_text_round_trip(tmp_path):
    from vipsy_amd import harness
    rng = np.random.RandomState(0)
    y = rng.randint(0, 2, size=(13, 7)).astype(np.uint8)
    y[rng.rand(13, 7) < 0.2] = 255
    p = os.path.join(str(tmp_path), "y.txt")
    harness.save_responses(p, y)
    assert "nan" in open(p).read()
    assert np.array_equal(harness.load_responses(p), y)
    with open(p, "w") as f:                                  # lsat.dat style: tab separated ints, no missing
        f.write("1\t0\t1\n0\t0\t1\n")
    assert harness.load_responses(p).tolist() == [[1, 0, 1], [0, 0, 1]]
    with open(p, "w") as f:
        f.write("0.5 1\n")
    with pytest.raises(ValueError):
        harness.load_responses(p)


def test_lsat_fixture_matches_text_loader(tmp_path):
    from vipsy_amd import harness
    here = os.path.dirname(os.path.abspath(__file__))
    y = np.load(os.path.join(here, "golden", "lsat6.npz"))["y"]
    p = os.path.join(str(tmp_path), "lsat.dat")
    np.savetxt(p, y, fmt="%d", delimiter="\t")
    assert np.array_equal(harness.load_responses(p), y.astype(np.uint8))


def test_case_files_and_summary(tmp_path):
    from vipsy_amd import harness
    rng = np.random.RandomState(1)
    y = rng.randint(0, 2, size=(20, 6)).astype(np.uint8)
    items = {"a": torch.rand(2, 6), "b": torch.randn(1, 6)}
    harness.save_case(str(tmp_path), "2pl", y, items, 2, file_postfix=3)
    y2, r = harness.load_case(str(tmp_path), "2pl", 20, 6, 2, file_postfix=3)
    assert np.array_equal(y2, y) and r.a.shape == (2, 6) and r.b.shape == (1, 6)
    assert torch.allclose(r.a, items["a"], atol=1e-6) and torch.allclose(r.b, items["b"], atol=1e-6)
    s = harness.summarize_rmse([{"a": 1.0, "b": 2.0}, {"a": 3.0, "b": 2.0}])
    assert s["a"] == (2.0, 1.0) and s["b"] == (2.0, 0.0)


@pytest.mark.gpu
def test_replications_from_files(tmp_path):
    """multiprocess_article_test_load_data_util call pattern (test.py:94-127), two tiny replications."""
    from vipsy_amd import harness, synth, vi
    dev = torch.device("cuda:0")
    for k in range(2):
        items = synth.irt_item_params(12, "irt_2pl", seed=30 + k)
        y = synth.simulate_responses(1500, 0, items, "irt_2pl", dev, seed=40 + k)
        harness.save_case(str(tmp_path), "2pl", y.cpu().numpy(), {"a": items["a"], "b": items["b"]}, 1, file_postfix=k)
    out = harness.multiprocess_article_test_load_data_util(
        "2pl", 1500, 12, 1, vi_class=vi.VIRT, try_count=2, folder=str(tmp_path),
        vi_fit_kwargs={"optim": vi.Adam({"lr": 5e-2}), "max_iter": 300})
    assert set(out) == {"a", "b"} and out["b"][0] < 0.2 and out["a"][0] < 0.4
