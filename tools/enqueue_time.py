# CPU-side enqueue time of a step vs GPU step time
import sys, time, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vipsy_amd import synth
from vipsy_amd.engine import IrtEngine, LrSpec
for wl in ["irt4pl_1d_bbvi_100kx100", "irt2pl_1d_bbvi_missing90_1Mx500"]:
    model, N, J, D, H, am, miss = bench.WORKLOADS[wl]
    dev = torch.device("cuda", 0)
    items = synth.irt_item_params(J, model, seed=20242)
    y = synth.simulate_responses(N, 0, items, model, dev, seed=20240, missing=miss)
    eng = IrtEngine(y, model=model, D=D, seed=1)
    lrs = LrSpec(1e-2)
    for _ in range(5): eng.step(lrs); 
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): eng.step(lrs)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(wl, "enqueue ms/step", (t1 - t0) / 200 * 1e3, "total ms/step", (t2 - t0) / 200 * 1e3)
