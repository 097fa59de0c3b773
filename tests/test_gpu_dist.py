"""The sharded step on the real HIP path: world 2 and world 4 against world 1 (tests/test_distributed_cpu.py covers the same
host logic with the oracle backend on CPU, up to world 8; a GPU box admits at most six processes on its card, so four ranks is
the widest rehearsal that may touch it).  With two devices visible the ranks use RCCL (backend nccl); on a one-GPU box
both ranks share cuda:0 and exchange through gloo -- everything but the RCCL transport is what the 8-GPU run executes."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "_dist_gpu_worker.py")


def _run(case, world, tmp_path, port):
    out = str(tmp_path / ("%s_w%d" % (case, world)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    if world == 1:
        cmd = [sys.executable, WORKER, case, out]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(port), WORKER, case, out]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    return [json.load(open(out + ".%d" % r)) for r in range(world)]


@pytest.mark.parametrize("case,port", [("mvn", 29711), ("irt1d", 29713), ("hodina", 29715), ("vaeccdm", 29717),
                                       ("cdmsf", 29719)])
def test_hip_two_ranks_match_one_rank(case, port, tmp_path):
    import torch
    one = _run(case, 1, tmp_path, port)[0]
    two = _run(case, 2, tmp_path, port + 1)
    if torch.cuda.device_count() >= 2:
        # a multi-GPU box must not rehearse on gloo: the transport under test is RCCL
        assert all(r["backend"] == "nccl" for r in two), [r["backend"] for r in two]
    if case in ("irt1d", "mvn"):
        # replayed from HIP graphs: one for a single rank, two around the eager all-reduce when the persons are sharded
        # (engine.py::_step_graph; the gloo rehearsal replays the same two graphs around its host reduction)
        assert one["graphed"] and all(r["graphed"] for r in two)
    for r in two:
        np.testing.assert_allclose(r["loss"], one["loss"], rtol=2e-5)
        np.testing.assert_allclose(r["P"], one["P"], rtol=2e-4, atol=2e-6)    # fp32 summation order differs by shard
    np.testing.assert_allclose(two[0]["P"], two[1]["P"], rtol=0, atol=0)       # replicated state: bit-identical
    if "PP" in one:
        n_one = len(one["PP"]) // 2
        loc = np.concatenate([np.array(r["PP"])[:len(r["PP"]) // 2] for r in two])
        raw = np.concatenate([np.array(r["PP"])[len(r["PP"]) // 2:] for r in two])
        np.testing.assert_allclose(loc, np.array(one["PP"])[:n_one], rtol=2e-4, atol=2e-6)
        np.testing.assert_allclose(raw, np.array(one["PP"])[n_one:], rtol=2e-4, atol=2e-6)


def test_hip_four_ranks_match_one_rank(tmp_path):
    """Four shards of the amortized multivariate step (264 persons each; rank 3's share of the subsampled step is its own
    count): losses and parameters against the single-rank run, every rank bit-identical, every rank replaying its graphs."""
    import torch
    one = _run("mvn", 1, tmp_path, 29731)[0]
    four = _run("mvn", 4, tmp_path, 29732)
    if torch.cuda.device_count() >= 4:
        assert all(r["backend"] == "nccl" for r in four), [r["backend"] for r in four]
    assert one["graphed"] and all(r["graphed"] for r in four)
    for r in four:
        np.testing.assert_allclose(r["loss"], one["loss"], rtol=2e-5)
        np.testing.assert_allclose(r["P"], one["P"], rtol=2e-4, atol=2e-6)
        assert r["P"] == four[0]["P"]


def test_bench_self_launch_four_ranks_of_125k():
    """`python bench.py --gpus 4` on shards of 125 000 persons -- the shard one of EIGHT GPUs holds in the driver's scaling run
    (four ranks is what a one-GPU box admits): the ranks start, shard, replay their two graphs
    around the all-reduce and print one line."""
    import torch
    backend = "nccl" if torch.cuda.device_count() >= 4 else "gloo"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3", "--warmup", "3",
                        "--persons", "500000", "--no-cpu-baseline", "--dist-backend", backend],
                       capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert p.returncode == 0, p.stderr[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1
    d = json.loads(line[0])
    assert d["n_gpus"] == 4 and d["value"] > 0 and d["config"]["persons_per_rank"] == 125000
    assert d["config"]["launch"].startswith("whole step replayed from HIP graphs"), d["config"]["launch"]
    # the exchange of a step, as the line reports it: one all-reduce of the flat buffer [417 314 gradients, with the unused
    # c / d segments and the alignment pad of the layout | loss] over four ranks, timed
    c = d["collective"]
    assert c["ranks"] == 4 and c["backend"] == backend and c["calls_per_step"] == 1
    assert 417315 <= c["floats"] <= 417315 + 1100 and c["bytes"] == 4 * c["floats"]
    assert c["allreduce_ms"] is not None and c["allreduce_ms"] > 0
    assert (backend == "nccl") != bool(d.get("rehearsal")), d["metric"]     # a gloo run says that it is a rehearsal
    assert np.isfinite(d["loss_first"]) and np.isfinite(d["loss_last"]) and d["loss_last"] < d["loss_first"]


def test_bench_self_launch_two_ranks():
    """`python bench.py --gpus 2` must start its own ranks (the driver's command) and print one JSON line with
    n_gpus == 2; RCCL when two devices are visible, the gloo rehearsal on a one-GPU box."""
    import torch
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--persons", "65536", "--no-cpu-baseline", "--dist-backend", backend],
                       capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert p.returncode == 0, p.stderr[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1
    d = json.loads(line[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["persons_per_rank"] == 32768


def test_rccl_one_rank_group():
    """RCCL itself on the box this suite runs on: a ONE-rank process group with backend nccl (a GPU box of the builder's pool
    has one card, and RCCL refuses two ranks on one device).  The all-reduce is then the identity, so the sharded code path --
    kernel by kernel, two graph replays around the eager all-reduce (a capture beside a live communicator), and the collective
    captured INTO the step's graph (VX_GRAPH_COLLECTIVE=1) -- must reproduce the group-less run bit for bit.  What it cannot
    show is the transport between GPUs.  Skips when the communicator cannot be made."""
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_SOCKET_IFNAME="lo")
    try:
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_rccl_one_rank.py"), "29761"], env=env, capture_output=True,
                           text=True, timeout=600)
    except subprocess.TimeoutExpired:
        pytest.skip("the one-rank RCCL communicator did not come up within ten minutes on this box")
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if p.returncode != 0 and not lines:
        pytest.skip("the one-rank RCCL worker ended with code %d: %s" % (p.returncode, p.stderr[-500:]))
    out = json.loads(lines[-1])
    if "skip" in out:
        pytest.skip("no RCCL communicator on this box: " + out["skip"])
    assert out["backend"] == "nccl"
    for name in ("eager", "two_replays", "captured_collective"):
        m = out["modes"][name]
        assert "error" not in m, (name, m)
        assert m["same_losses"] and m["same_params"], (name, m)
    assert out["modes"]["two_replays"]["fallback"] is None     # the capture beside the live communicator did not fall back
