"""Step time of the amortized HO-DINA guide (VaeCHoDina: NormEncoder over the responses) around J % 4 and hidden_dim 64.
usage (GPU box): python tools/hodina_cliffs.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vipsy_amd.engine import HoDinaEngine, LrSpec
dev = torch.device("cuda:0"); N, K = 200000, 8
for J, H, amort in ((30, 64, False), (30, 64, True), (32, 64, True), (30, 32, True)):
    rng = np.random.RandomState(J)
    q = (rng.rand(K, J) < 0.3).astype(np.float32); q[0, q.sum(0) == 0] = 1
    g = torch.Generator(device=dev); g.manual_seed(J)
    y = (torch.rand(N, J, device=dev, generator=g) < 0.5).to(torch.uint8)
    eng = HoDinaEngine(y, torch.from_numpy(q).to(dev), amortized=amort, H=H, seed=3)
    lrs = LrSpec(lambda m, p: {"lr": 1e-3})
    eng.steps(lrs, [None] * 6); torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 20
    eng.steps(lrs, [None] * n); torch.cuda.synchronize()
    print("HO-DINA N = %d K = %d J = %d H = %d amortized = %s : %.3f ms/step" % (N, K, J, H, amort, 1e3 * (time.perf_counter() - t0) / n), flush=True)
# the reference's own usage (test.py:598-607): 20 rows a particle, 10 particles
for J in (30, 32):
    rng = np.random.RandomState(J)
    q = (rng.rand(K, J) < 0.3).astype(np.float32); q[0, q.sum(0) == 0] = 1
    g = torch.Generator(device=dev); g.manual_seed(J)
    y = (torch.rand(N, J, device=dev, generator=g) < 0.5).to(torch.uint8)
    eng = HoDinaEngine(y, torch.from_numpy(q).to(dev), amortized=True, H=64, seed=3)
    lrs = LrSpec(lambda m, p: {"lr": 1e-3})
    draws = np.random.RandomState(1)
    def rows():
        return [torch.from_numpy(np.unique(draws.randint(0, N, 40))[:20].astype(np.int64)) for _ in range(10)]   # (cheap distinct rows)
    for _ in range(12):
        eng.step(lrs, rows=rows(), b_global=20, num_particles=10)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 200
    for _ in range(n):
        eng.step(lrs, rows=rows(), b_global=20, num_particles=10)
    torch.cuda.synchronize()
    print("HO-DINA amortized, 20 rows x 10 particles, J = %d : %.1f us/step" % (J, 1e6 * (time.perf_counter() - t0) / n), flush=True)
