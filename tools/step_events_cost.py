"""What the in-region measurement costs the judged step: the 1M x 500 x 100 step eager with the HIP events bench.py records
(library brackets around the large kernels + the engine's phase events), eager without any, and replayed from its HIP graphs
(VX_GRAPH_MAX_PERSONS lifted).  usage (GPU box): python tools/step_events_cost.py [persons]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
os.environ.setdefault("VX_GRAPH_MAX_PERSONS", "2000000")
import torch
from vipsy_amd import synth, _hip
from vipsy_amd.engine import IrtEngine, LrSpec

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
J, D, H = 500, 100, 64
dev = torch.device("cuda:0")
a, b = synth.mirt_item_params(J, D, seed=20243)
y = synth.simulate_responses(N, 0, {"a": a, "b": b}, "irt_2pl", dev, seed=20240)
lrs = LrSpec(lambda m, p: {"lr": 1e-2 if p in ("a", "b") else 1e-3})
eng = IrtEngine(y, model="irt_2pl", D=D, n_global=N, gid0=0, amortized=True, H=H, seed=1234)


def timed(fn, n=30, reps=3):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fn(n)
        torch.cuda.synchronize(); out.append(1e3 * (time.perf_counter() - t0) / n)
    return out


eng.use_graph = False
for _ in range(5):
    eng.step(lrs)
print("eager, no events      :", ["%.3f" % v for v in timed(lambda n: [eng.step(lrs) for _ in range(n)])])
ev = []
eng.events = ev
_hip.lib().vx_prof_enable(1)
print("eager, all events     :", ["%.3f" % v for v in timed(lambda n: [eng.step(lrs) for _ in range(n)])])
_hip.lib().vx_prof_enable(0)
eng.events = None
print("eager, no events again:", ["%.3f" % v for v in timed(lambda n: [eng.step(lrs) for _ in range(n)])])
eng.use_graph = True
eng.steps(lrs, [None] * 8)
print("replayed (4 a replay) :", ["%.3f" % v for v in timed(lambda n: eng.steps(lrs, [None] * (4 * (n // 4))), n=32)])
