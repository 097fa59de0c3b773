#!/usr/bin/env python3
"""The reference's CFA demo as a step (test.py:420-430: VIRT, x_feature = 2, subsample_size = 100, Trace_ELBO(num_particles = 20)):
host loop over the particles against the one-graph replay.    python tools/particles_probe.py [--S 20] [--steps 300]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vipsy_amd.engine import IrtEngine, LrSpec                # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--S", type=int, default=20)
    ap.add_argument("--N", type=int, default=1000)
    ap.add_argument("--B", type=int, default=100)
    ap.add_argument("--steps", type=int, default=300)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    rng = np.random.RandomState(1)
    J = 6
    y = torch.from_numpy(rng.randint(0, 2, size=(args.N, J)).astype(np.uint8)).to(dev)
    a_free = torch.tensor([[1, 1, 1, 0, 0, 0], [0, 0, 0, 1, 1, 1]], dtype=torch.float32)
    lrs = LrSpec({"lr": 1e-2})
    rg = np.random.Generator(np.random.PCG64(7))

    def draw():
        return [torch.from_numpy(rg.choice(args.N, size=args.B, replace=False).astype(np.int64)) for _ in range(args.S)]
    for graph in (False, True):
        eng = IrtEngine(y, model="irt_2pl", D=2, a_free=a_free, a0=a_free, seed=3)
        eng.use_graph = graph
        for _ in range(5):
            eng.step(lrs, rows=draw(), b_global=args.B, num_particles=args.S)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eng.step(lrs, rows=draw(), b_global=args.B, num_particles=args.S)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%-6s S=%d B=%d: %.1f us/step (%.0f steps/s)" % ("graph" if graph else "eager", args.S, args.B, 1e6 * dt / args.steps, args.steps / dt))


if __name__ == "__main__":
    main()
