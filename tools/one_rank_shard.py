"""The sharded step's host path at a shard's size on ONE GPU: a one-rank RCCL group (the all-reduce moves nothing), the step as
two graph replays around the eager all-reduce, against the group-less single replay and the eager step.  What a rank of an
N-GPU run does per step except the transport.   usage (GPU box): python tools/one_rank_shard.py [persons ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch.distributed as dist
    from vipsy_amd import synth
    from vipsy_amd.engine import IrtEngine, LrSpec
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:29771", rank=0, world_size=1)
    J, D, H = 500, 100, 64
    a, b = synth.mirt_item_params(J, D, seed=20243)
    for N in [int(v) for v in sys.argv[1:]] or [125000, 500000]:
        y = synth.simulate_responses(N, 0, {"a": a, "b": b}, "irt_2pl", torch.device("cuda:0"), seed=20240)
        for name, group, graph in (("no group, eager", None, False), ("no group, one replay", None, True),
                                   ("one-rank RCCL group, eager", dist.group.WORLD, False),
                                   ("one-rank RCCL group, two replays around the all-reduce", dist.group.WORLD, True)):
            eng = IrtEngine(y, model="irt_2pl", D=D, amortized=True, H=H, seed=11, group=group)
            eng.use_graph = graph
            lrs = LrSpec(lambda m, p: {"lr": 1e-3})
            eng.steps(lrs, [None] * 12)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 40
            losses = eng.steps(lrs, [None] * n)
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / n
            print("N = %7d  %-58s %.3f ms/step   loss %.6e   fallback %s" % (N, name, ms, float(losses[-1]), getattr(eng, "graph_fallback", None)),
                  flush=True)
            del eng
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
