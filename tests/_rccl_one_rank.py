"""Worker of tests/test_gpu_dist.py::test_rccl_one_rank_group: the sharded step's code path with backend nccl (= RCCL) on a
ONE-rank group -- the all-reduce is the identity, so every mode must reproduce the group-less run bit for bit: kernel by kernel,
two replays around the eager all-reduce, and (VX_GRAPH_COLLECTIVE=1) the collective captured into the step's one graph.
Prints one JSON line; {"skip": reason} when the communicator cannot be made on this box."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    port = sys.argv[1]
    import torch.distributed as dist
    try:
        torch.cuda.set_device(0)
        dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%s" % port, rank=0, world_size=1)
        probe = torch.ones(8, device="cuda")
        dist.all_reduce(probe)
        torch.cuda.synchronize()
    except Exception as e:                                   # no RCCL transport on this box
        print(json.dumps({"skip": repr(e)[:300]}))
        return
    from vipsy_amd.engine import IrtEngine, LrSpec
    rng = np.random.RandomState(3)
    N, J, D, H = 2048, 500, 100, 64
    y = rng.randint(0, 2, size=(N, J)).astype(np.uint8)
    y[rng.rand(N, J) < 0.1] = 255
    yt = torch.from_numpy(y).cuda()

    def run(group, graph):
        eng = IrtEngine(yt, model="irt_2pl", D=D, amortized=True, H=H, seed=11, group=group)
        eng.use_graph = graph
        lrs = LrSpec(lambda m, p: {"lr": 1e-2 if p in ("a", "b") else 1e-3})
        losses = [eng.step(lrs) for _ in range(6)]
        torch.cuda.synchronize()
        return torch.stack(losses).cpu().numpy(), eng.P.cpu().numpy().copy(), getattr(eng, "graph_fallback", None)
    ref = run(None, True)
    out = {"backend": dist.get_backend(), "modes": {}}
    for name, graph, env in (("eager", False, "0"), ("two_replays", True, "0"), ("captured_collective", True, "1")):
        os.environ["VX_GRAPH_COLLECTIVE"] = env
        try:
            got = run(dist.group.WORLD, graph)
            out["modes"][name] = {"same_losses": bool(np.array_equal(got[0], ref[0])), "same_params": bool(np.array_equal(got[1], ref[1])),
                                  "fallback": got[2]}
        except Exception as e:
            out["modes"][name] = {"error": repr(e)[:300]}
    os.environ["VX_GRAPH_COLLECTIVE"] = "0"
    print(json.dumps(out))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
