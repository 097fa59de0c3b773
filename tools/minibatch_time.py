# step rate of the headline model at the reference's own minibatch sizes (test.py:338 uses subsample_size=100)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vipsy_amd import synth
from vipsy_amd.engine import IrtEngine, LrSpec
dev = torch.device("cuda", 0)
N, J, D, H = 1000000, 500, 100, 64
a, b = synth.mirt_item_params(J, D, seed=20243)
y = synth.simulate_responses(N, 0, {"a": a, "b": b}, "irt_2pl", dev, seed=20240, missing=0.0)
eng = IrtEngine(y, model="irt_2pl", D=D, amortized=True, H=H, seed=1234)
lrs = LrSpec(lambda m, p: {"lr": 1e-2 if p in ("a", "b") else 1e-3})
for B in (100, 1000, 10000, 100000):
    g = np.random.Generator(np.random.PCG64(1))
    draw = lambda: torch.from_numpy(g.choice(N, size=B, replace=False, shuffle=True).astype(np.int64)).to(dev, non_blocking=True)
    for it in range(5):
        eng.step(lrs, rows=draw(), b_global=B)
    torch.cuda.synchronize()
    n = 100 if B <= 10000 else 20
    t0 = time.perf_counter()
    for it in range(n):
        eng.step(lrs, rows=draw(), b_global=B)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("B = %6d : %.3f ms/step, %.0f steps/s, %.2f M person-rows/s" % (B, dt * 1e3, 1 / dt, B / dt / 1e6))
