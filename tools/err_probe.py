import json, os, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", ".")
code = r'''
import json, os, sys, numpy as np, torch
sys.path.insert(0, %r)
from vipsy_amd.engine import IrtEngine
rng = np.random.RandomState(5)
N, J, D, H = 512, 500, 100, 64
y = rng.randint(0, 2, size=(N, J)).astype(np.uint8); y[rng.rand(N, J) < 0.2] = 255
eng = IrtEngine(torch.from_numpy(y).cuda(), model="irt_2pl", D=D, amortized=True, H=H, seed=21)
eng.unconstrained("b").copy_(torch.from_numpy(0.5 * rng.randn(1, J)).float())
eng.loss_and_grads()
torch.cuda.synchronize()
out = {n: eng.unconstrained(n, eng.G).double().cpu().numpy().reshape(-1).tolist() for n in eng.names()}
print("RESULT" + json.dumps(out))
''' % ROOT
res = {}
for mode, extra in {"fast": {}, "generic": {"VX_FORCE_GENERIC": "1"}}.items():
    env = dict(os.environ); env.update(extra)
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")][0]
    res[mode] = json.loads(line[6:])
import numpy as np
for n in res["fast"]:
    a, b = np.array(res["fast"][n]), np.array(res["generic"][n])
    print("%-22s max|diff| / max|g| = %.2e" % (n, np.abs(a - b).max() / max(1e-30, np.abs(b).max())))
