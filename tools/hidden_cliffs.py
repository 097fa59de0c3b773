"""Step time of the amortized guides around hidden_dim 64 (the MFMA kernels' width): what a narrower encoder costs.
(hidden_dim 128 with x_feature 100 is refused by the generic forward: VX_EINVAL.)   usage (GPU box): python tools/hidden_cliffs.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vipsy_amd.engine import IrtEngine, LrSpec
dev = torch.device("cuda:0"); N = 200000
for J, D, H in ((500, 100, 64), (500, 100, 32), (500, 100, 48), (40, 8, 32), (40, 8, 64), (500, 1, 32), (500, 1, 64)):
    g = torch.Generator(device=dev); g.manual_seed(J * 1000 + D)
    y = (torch.rand(N, J, device=dev, generator=g) < 0.5).to(torch.uint8)
    eng = IrtEngine(y, model="irt_2pl", D=D, amortized=True, H=H, seed=3)
    lrs = LrSpec(lambda m, p: {"lr": 1e-3})
    eng.steps(lrs, [None] * 4); torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 8
    eng.steps(lrs, [None] * n); torch.cuda.synchronize()
    print("J = %3d  D = %3d  H = %3d : %8.3f ms/step" % (J, D, H, 1e3 * (time.perf_counter() - t0) / n), flush=True)
    del eng, y
