"""Worker for the world_size-2 gloo tests (launched by tests/test_distributed_cpu.py)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.oracle_backend import OracleBackend          # noqa: E402
from vipsy_amd.engine import IrtEngine, LrSpec          # noqa: E402


def main():
    case, out_path = sys.argv[1], sys.argv[2]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        torch.distributed.init_process_group(backend="gloo", rank=rank, world_size=world)
    rng = np.random.RandomState(123)
    if case == "irt1d":
        N, J, D, model, amort = 240, 17, 1, "irt_4pl", False
    elif case == "irt1d8":                                  # eight ranks: shards of 30 and a ragged last one of 27
        N, J, D, model, amort = 237, 17, 1, "irt_4pl", False
    elif case == "mvn8":                                    # eight ranks: shards of 12 and a last one of 7
        N, J, D, model, amort = 91, 21, 3, "irt_2pl", True
    else:
        N, J, D, model, amort = 96, 21, 3, "irt_2pl", True
    y = rng.randint(0, 2, size=(N, J)).astype(np.uint8)
    y[rng.rand(N, J) < 0.2] = 255
    enc = None
    if amort:
        H = 8
        enc = {"fc1.weight": rng.randn(H, J) / 4, "fc1.bias": 0.1 * rng.randn(H), "fc21.weight": rng.randn(D, H) / 3,
               "fc21.bias": 0.1 * rng.randn(D), "fc22.weight": 0.2 * rng.randn(D * (D + 1) // 2, H),
               "fc22.bias": 0.05 * rng.randn(D * (D + 1) // 2)}
    per = (N + world - 1) // world
    lo, hi = rank * per, min(N, rank * per + per)
    group = torch.distributed.group.WORLD if world > 1 else None       # sharding is explicit, never inferred
    eng = IrtEngine(torch.from_numpy(y[lo:hi]), model=model, D=D, n_global=N, gid0=lo, amortized=amort, H=8,
                    encoder_init=enc, seed=77, backend=OracleBackend(), group=group, observed_lists=False)
    lrs = LrSpec(lambda m, n: {"lr": 1e-2 if n in ("a", "b") else 3e-3})
    losses = []
    # global subsample drawn identically on every rank, then intersected with the local shard
    for t in range(3):
        g = np.random.RandomState(1000 + t)
        if t == 1:
            rows, bg = None, N                                    # one full-batch step
        else:
            # (t == 2 of the eight-rank cases: a subsample so small that some ranks hold none of it)
            idx = np.sort(g.permutation(N)[:(5 if (t == 2 and case.endswith("8")) else N // 2)])
            mine = idx[(idx >= lo) & (idx < hi)] - lo
            rows, bg = torch.from_numpy(mine.astype(np.int64)), len(idx)
        losses.append(float(eng.step(lrs, rows=rows, b_global=bg)))
    res = {"loss": losses, "P": eng.P.double().numpy().tolist()}
    if not amort:
        res["PP"] = eng.PP.double().numpy().tolist()
        res["lo"], res["hi"] = lo, hi
    with open(out_path + ".%d" % rank, "w") as f:
        json.dump(res, f)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
