import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from oracle import vi_oracle as vo
from vipsy_amd.engine import HoDinaEngine
for (N,J,K,miss,B) in [(333,30,int(os.environ.get("KK","5")),0.2,77)]:
    rng = np.random.RandomState(N + J + K)
    q = (rng.rand(K, J) < 0.4).astype(np.float32)
    q[rng.randint(0, K, size=J), np.arange(J)] = 1.0
    y = rng.randint(0, 2, size=(N, J)).astype(np.uint8)
    y[rng.rand(N, J) < miss] = 255
    eng = HoDinaEngine(torch.from_numpy(y).cuda(), q, amortized=False, H=64, seed=9)
    eng.unconstrained("lam0").copy_(torch.from_numpy(0.5 * rng.randn(1, K)).float())
    eng.unconstrained("lam1").copy_(torch.from_numpy(0.4 * rng.randn(1, K)).float())
    eng.unconstrained("g").add_(torch.from_numpy(0.5 * rng.randn(1, J)).float().cuda())
    eng.unconstrained("s").add_(torch.from_numpy(0.5 * rng.randn(1, J)).float().cuda())
    eng.PP.copy_(torch.from_numpy(np.concatenate([rng.randn(N), 0.3 * rng.randn(N)])).float())
    idx = np.arange(N) if B is None else np.sort(rng.permutation(N)[:B])
    rows = None if B is None else torch.from_numpy(idx).cuda()
    eps = vo.philox_normals(9, 0, 0, idx, 1)
    eng.loss_and_grads(rows, len(idx)); torch.cuda.synchronize()
    spec = {"family": "hodina", "K": K, "N": N, "amortized": False, "q": q}
    params = {n: eng.unconstrained(n).cpu().numpy().astype(np.float64) for n in eng.all_names()}
    loss_o, g_o = vo.loss_and_grads(spec, params, y, [idx], [eps])
    print((N,J,K,B), "loss", float(eng.G[eng.n_params].item()), loss_o)
    for name, go in g_o.items():
        gh = eng.unconstrained(name, eng.GP if (eng.per_person and name in eng.pp_off) else eng.G).cpu().numpy()
        d = np.abs(gh - go).reshape(-1)
        print("   ", name, "maxerr", d.max(), "scale", np.abs(go).max(), "argmax", d.argmax(), gh.reshape(-1)[:6], go.reshape(-1)[:6])
