#!/usr/bin/env python3
"""Where a B = 100 step of the headline model goes (the reference's own usage, test.py:338): eager against replayed, with the
host-side cost of a step (rows drawn and staged, launches) separated from the device-side cost.
    python tools/minibatch_probe.py [--persons N] [--B 100] [--steps 300]"""
import argparse
import gc
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vipsy_amd import synth                                   # noqa: E402
from vipsy_amd.engine import IrtEngine, LrSpec                # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--persons", type=int, default=200000)
    ap.add_argument("--B", type=int, default=100)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--modes", default="eager,graph,graph4")
    ap.add_argument("--k", type=int, default=0, help="steps a replay for mode graph4 (0: the engine's graph_steps)")
    ap.add_argument("--time-sync", action="store_true", help="time the host's waits for the GPU (Event.synchronize)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    J, D, H = 500, 100, 64
    a, b = synth.mirt_item_params(J, D, seed=20243)
    y = synth.simulate_responses(args.persons, 0, {"a": a, "b": b}, "irt_2pl", dev, seed=20240, missing=0.0)
    lrs = LrSpec(lambda m, p: {"lr": 1e-2 if p in ("a", "b") else 1e-3})
    rg = np.random.Generator(np.random.PCG64(7))

    def draw():
        return torch.from_numpy(rg.choice(args.persons, size=args.B, replace=False).astype(np.int64))
    for mode in args.modes.split(","):
        eng = IrtEngine(y, model="irt_2pl", D=D, amortized=True, H=H, seed=1234)
        eng.use_graph = mode in ("graph", "graph4")
        if args.k:
            eng.graph_steps = args.k
        if mode == "graph4":                                   # what the fit loop does: four steps a replay (IrtEngine.steps)
            def run(n, fixed=None):
                i = 0
                while i < n:
                    k = min(eng.graph_steps, n - i)
                    eng.steps(lrs, [fixed if fixed is not None else draw() for _ in range(k)], b_global=args.B)
                    i += k
        else:
            def run(n, fixed=None):
                for _ in range(n):
                    eng.step(lrs, rows=(fixed.to(dev) if mode == "eager" else fixed) if fixed is not None else draw(), b_global=args.B)
        run(12)                                                # (eager step, the captures, a few replays)
        torch.cuda.synchronize()
        gc.collect()
        gc.freeze()                                            # (a full collection of the interpreter's heap is a 40 ms stall of the
        blocked = [0.0]                                        #  host, once, somewhere in the first few hundred steps: not the step's cost)
        ev_sync = torch.cuda.Event.synchronize

        def timed_sync(ev):                                    # time the host spends waiting for the GPU (ring slots / pinned buffers)
            tb = time.perf_counter()
            ev_sync(ev)
            blocked[0] += time.perf_counter() - tb
        if args.time_sync:
            torch.cuda.Event.synchronize = timed_sync
        t0 = time.perf_counter()
        run(args.steps)
        t_host = time.perf_counter() - t0                      # the host has enqueued everything
        torch.cuda.Event.synchronize = ev_sync
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        # the same replays without a new draw (device-side cost of the step alone)
        r = draw()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run(args.steps, fixed=r)
        torch.cuda.synchronize()
        t_fixed = time.perf_counter() - t1
        t2 = time.perf_counter()
        for _ in range(args.steps):
            draw()
        t_draw = time.perf_counter() - t2
        print("%-6s B=%d: %.1f us/step (%.0f steps/s); host enqueue %.1f us/step of which %.1f waiting for the GPU; same rows "
              "re-used %.1f us/step; draw alone %.1f us"
              % (mode, args.B, 1e6 * t_all / args.steps, args.steps / t_all, 1e6 * t_host / args.steps,
                 1e6 * blocked[0] / args.steps, 1e6 * t_fixed / args.steps, 1e6 * t_draw / args.steps))


if __name__ == "__main__":
    main()
