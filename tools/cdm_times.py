"""Step times of the cognitive-diagnosis engines at the reference's shapes (J = 30 items, K = 3 / 5 attributes), full batch and
the reference's subsamples.   usage (GPU box): python tools/cdm_times.py [persons]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vipsy_amd.engine import CcdmEngine, VaeCcdmEngine, CdmSfEngine, LrSpec
dev = torch.device("cuda:0"); N = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
J, K = int(os.environ.get("CDM_J", "30")), 5
rng = np.random.RandomState(1)
q = (rng.rand(K, J) < 0.4).astype(np.float32); q[0, q.sum(0) == 0] = 1
g = torch.Generator(device=dev); g.manual_seed(1)
y = (torch.rand(N, J, device=dev, generator=g) < 0.5).to(torch.uint8)
qt = torch.from_numpy(q)
for name, mk in (("VCCDM (enumerated)", lambda: CcdmEngine(y, qt)), ("VaeCCDM (SoftmaxEncoder)", lambda: VaeCcdmEngine(y, qt, H=64)),
                 ("VCDM (score function)", lambda: CdmSfEngine(y, qt, amortized=False)),
                 ("VaeCDM (BinEncoder, score function)", lambda: CdmSfEngine(y, qt, amortized=True, H=64))):
    eng = mk()
    lrs = LrSpec(lambda m, p: {"lr": 1e-3})
    for B in (None, 100):
        draws = np.random.RandomState(2)
        def rows():
            return None if B is None else torch.from_numpy(np.unique(draws.randint(0, N, 3 * B))[:B].astype(np.int64))
        n = 48
        pre = [rows() for _ in range(n)]                       # (drawn ahead: the fit loop's draw is not what is timed)
        eng.steps(lrs, pre[:24], b_global=B)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.steps(lrs, pre, b_global=B)
        torch.cuda.synchronize()
        print("%-38s N = %d  J = %d  K = %d  batch %-6s: %8.3f ms/step" % (name, N, J, K, "full" if B is None else B, 1e3 * (time.perf_counter() - t0) / n), flush=True)
    del eng
