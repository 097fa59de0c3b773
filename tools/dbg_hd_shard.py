"""Per-person gradients of the HO-DINA step from one engine over all persons against two engines over the two halves
(same global person ids): the kernel's per-person arithmetic must not depend on which persons share a wave."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vipsy_amd.engine import HoDinaEngine

dev = torch.device("cuda", 0)
rng = np.random.RandomState(321)
N, J, K = 900, 30, 6
y = rng.randint(0, 2, size=(N, J)).astype(np.uint8)
y[rng.rand(N, J) < 0.15] = 255
q = (rng.rand(K, J) < 0.5).astype(np.float32)
q[0, q.sum(0) == 0] = 1
full = HoDinaEngine(torch.from_numpy(y).to(dev), q, n_global=N, gid0=0, seed=77)
full.loss_and_grads()
torch.cuda.synchronize()
gp = full.GP.cpu().numpy().copy()
per = 450
parts = []
for r in range(2):
    e = HoDinaEngine(torch.from_numpy(y[r * per:(r + 1) * per]).to(dev), q, n_global=N, gid0=r * per, seed=77)
    e.loss_and_grads()
    torch.cuda.synchronize()
    parts.append(e.GP.cpu().numpy().copy())
loc2 = np.concatenate([p[:per] for p in parts]); raw2 = np.concatenate([p[per:] for p in parts])
d1 = np.abs(loc2 - gp[:N]); d2 = np.abs(raw2 - gp[N:])
print("gloc: max abs diff %.3g (max |g| %.3g), worst persons %s" % (d1.max(), np.abs(gp[:N]).max(), np.argsort(-d1)[:8]))
print("graw: max abs diff %.3g (max |g| %.3g), worst persons %s" % (d2.max(), np.abs(gp[N:]).max(), np.argsort(-d2)[:8]))
print("small |gloc| persons:", int((np.abs(gp[:N]) < 1e-5).sum()), " values", np.sort(np.abs(gp[:N]))[:6])
from oracle import vi_oracle as vo
spec = {"family": "hodina", "K": K, "N": N, "amortized": False, "q": q}
params = {n: full.unconstrained(n).cpu().numpy().astype(np.float64) for n in full.all_names()}
eps = vo.philox_normals(77, 0, 0, np.arange(N), 1)
loss_o, g_o = vo.loss_and_grads(spec, params, y, [np.arange(N)], [eps])
print("oracle grads keys", list(g_o.keys()))
for nm in g_o:
    if nm in full.pp_off:
        gh = full.unconstrained(nm, full.GP).cpu().numpy().reshape(-1)
        go = np.asarray(g_o[nm]).reshape(-1)
        d = np.abs(gh - go)
        w = np.argsort(-d)[:6]
        print(nm, "max err %.3g of max %.3g; worst persons %s hip %s oracle %s" % (d.max(), np.abs(go).max(), w, gh[w], go[w]))
print("loss hip %.6f oracle %.6f" % (float(full.G[full.n_params]), loss_o))
gh = full.unconstrained("theta_local", full.GP).cpu().numpy().reshape(-1)
go = np.asarray(g_o["theta_local"]).reshape(-1)
bad = np.flatnonzero(np.abs(gh - go) > 1e-3)
print("bad persons", len(bad), "groups fully bad:", [g for g in range((N+31)//32) if all((p in set(bad.tolist())) for p in range(32*g, min(N, 32*g+32)))], "groups fully good:", [g for g in range((N+31)//32) if not any((p in set(bad.tolist())) for p in range(32*g, min(N,32*g+32)))])
print("slow-path groups (a person with |theta| > 2.45):", sorted(set((np.flatnonzero(np.abs(eps.reshape(-1)) > 2.45) // 32).tolist())))
print("their theta", np.round(eps.reshape(-1)[bad], 2).tolist())
print("persons with |theta| > 2.3:", np.flatnonzero(np.abs(eps.reshape(-1)) > 2.3).tolist())
