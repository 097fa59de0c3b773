import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vipsy_amd.engine import IrtEngine
N, J, D, H = [int(v) for v in sys.argv[1:5]]
rng = np.random.RandomState(N + J + D)
y = rng.randint(0, 2, size=(N, J)).astype(np.uint8); y[rng.rand(N, J) < 0.1] = 255
eng = IrtEngine(torch.from_numpy(y).cuda(), model="irt_2pl", D=D, amortized=True, H=H, seed=11)
eng.loss_and_grads(); torch.cuda.synchronize()
for name in eng.names():
    g = eng.unconstrained(name, eng.G).cpu().numpy()
    bad = ~np.isfinite(g)
    print(name, g.shape, "non-finite:", int(bad.sum()), "first:", np.argwhere(bad)[:6].tolist(), "absmax", float(np.nanmax(np.abs(g))))
ws = eng._ws["encb_ws"]
print("maxw words:", ws[-8:-4].cpu().numpy())
print("scales:", eng._ws["packws"][-16:].cpu().numpy())
