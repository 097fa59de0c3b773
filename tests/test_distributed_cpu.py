"""world_size-2 and world_size-8 gloo runs of the engine's host logic (oracle-backed compute, CPU): the sharded run must
reproduce the single-process run -- same losses, same replicated parameters, same per-person rows.  World 8 is the
driver's multi-GPU shape: eight contiguous person ranges, the last one shorter when N % 8 != 0."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "_dist_worker.py")


def _run(case, world, tmp_path, port):
    out = str(tmp_path / ("%s_w%d" % (case, world)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    if world == 1:
        cmd = [sys.executable, WORKER, case, out]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(port), WORKER, case, out]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    return [json.load(open(out + ".%d" % r)) for r in range(world)]


@pytest.mark.parametrize("case,port", [("irt1d", 29611), ("mvn", 29613)])
def test_two_ranks_match_one_rank(case, port, tmp_path):
    one = _run(case, 1, tmp_path, port)[0]
    two = _run(case, 2, tmp_path, port + 1)
    for r in two:
        np.testing.assert_allclose(r["loss"], one["loss"], rtol=1e-5)
        np.testing.assert_allclose(r["P"], one["P"], rtol=1e-4, atol=1e-6)      # replicated params identical
    np.testing.assert_allclose(two[0]["P"], two[1]["P"], rtol=0, atol=0)         # bit-identical across ranks
    if "PP" in one:
        n = len(one["PP"]) // 2
        loc = np.concatenate([np.array(r["PP"])[:len(r["PP"]) // 2] for r in two])
        raw = np.concatenate([np.array(r["PP"])[len(r["PP"]) // 2:] for r in two])
        np.testing.assert_allclose(loc, np.array(one["PP"])[:n], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(raw, np.array(one["PP"])[n:], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("case,port", [("irt1d8", 29641), ("mvn8", 29643)])
def test_eight_ranks_match_one_rank(case, port, tmp_path):
    """The 8-way partition the driver's scaling run uses: contiguous person ranges with a ragged
    last shard (N % 8 != 0), ranks whose share of a global subsample is small or EMPTY, epsilon keyed by the global person id,
    one all-reduce a step, replicated Adam -- world 8 against world 1, every rank bit-identical to rank 0."""
    one = _run(case, 1, tmp_path, port)[0]
    eight = _run(case, 8, tmp_path, port + 1)
    assert len(eight) == 8
    for r in eight:
        np.testing.assert_allclose(r["loss"], one["loss"], rtol=1e-5)
        np.testing.assert_allclose(r["P"], one["P"], rtol=1e-4, atol=1e-6)
        assert r["P"] == eight[0]["P"]                                            # replicated state: bit-identical across ranks
    if "PP" in one:
        n = len(one["PP"]) // 2
        assert [r["lo"] for r in eight] == sorted(r["lo"] for r in eight) and eight[-1]["hi"] == n
        assert eight[-1]["hi"] - eight[-1]["lo"] < eight[0]["hi"] - eight[0]["lo"]     # the ragged last shard
        loc = np.concatenate([np.array(r["PP"])[:len(r["PP"]) // 2] for r in eight])
        raw = np.concatenate([np.array(r["PP"])[len(r["PP"]) // 2:] for r in eight])
        np.testing.assert_allclose(loc, np.array(one["PP"])[:n], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(raw, np.array(one["PP"])[n:], rtol=1e-4, atol=1e-6)


def test_fit_loop_draws_a_stratified_subsample_on_eight_ranks():
    """vipsy_amd/vi.py::_subsample with world = 8: every rank draws B / 8 distinct rows of ITS shard and reports the global
    batch 8 * (B / 8) (the plate scale N / B the engine applies uses the global figures)."""
    import torch
    from tests.oracle_backend import OracleBackend
    from vipsy_amd import vi
    y = torch.from_numpy(np.random.RandomState(4).randint(0, 2, size=(125, 6)).astype(np.uint8))
    m = vi.VIRT(data=y, model="irt_2pl", subsample_size=100, backend=OracleBackend(), seed=5)
    m.world, m.sample_size = 8, 1000                          # (what BasePsy derives from a process group of eight)
    idx, bg = m._subsample()
    a = idx.cpu().numpy()
    assert bg == 96 and a.shape == (12,) and len(set(a.tolist())) == 12 and a.min() >= 0 and a.max() < 125


def test_harness_replications_over_two_ranks(tmp_path):
    """Replications spread over ranks (test.py:94-127 farms them to a process pool): every replication owns its problem
    -- no all-reduce between different data sets -- and an uneven try_count (3 over 2 ranks) neither hangs nor changes
    any error: the summary equals the single-process one bit for bit."""
    import torch
    from vipsy_amd import harness
    worker = os.path.join(ROOT, "tests", "_harness_worker.py")
    rng = np.random.RandomState(5)
    for k in range(3):
        y = rng.randint(0, 2, size=(40, 6)).astype(np.uint8)
        y[rng.rand(40, 6) < 0.1] = 255
        harness.save_case(str(tmp_path), "2pl", y, {"a": torch.rand(1, 6) + 0.5, "b": torch.randn(1, 6)}, 1, file_postfix=k)
    res = {}
    for world, port in ((1, 29631), (2, 29633)):
        out = str(tmp_path / ("h_w%d" % world))
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        if world == 1:
            cmd = [sys.executable, worker, str(tmp_path), out, "3"]
        else:
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                   "--master-addr", "127.0.0.1", "--master-port", str(port), worker, str(tmp_path), out, "3"]
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        res[world] = [json.load(open(out + ".%d" % r)) for r in range(world)]
    assert res[2][0] == res[2][1] == res[1][0]
    assert set(res[1][0]) == {"a", "b"}
