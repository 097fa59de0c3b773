"""Where do the item-gradient errors of k_irt_lik_b sit?  Engine step at the headline shape, then GA / G_b / gx recomputed in
float64 from the x the forward kernel produced."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vipsy_amd.engine import IrtEngine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 33024
J, D, H = 500, 100, 64
rng = np.random.RandomState(N)
y = rng.randint(0, 2, size=(N, J)).astype(np.uint8); y[rng.rand(N, J) < 0.1] = 255
eng = IrtEngine(torch.from_numpy(y).cuda(), model="irt_2pl", D=D, amortized=True, H=H, seed=11)
a0 = eng.unconstrained("a") * torch.from_numpy(1 + 0.3 * rng.randn(D, J)).float().cuda()
eng.unconstrained("a").copy_(a0 * eng.unconstrained("a", eng.free))
eng.unconstrained("b").copy_(torch.from_numpy(0.5 * rng.randn(1, J)).float())
eng.loss_and_grads(); torch.cuda.synchronize()
x = eng.last["fw"]["x"][:N * D].reshape(N, D).double().cpu().numpy()
a = eng.unconstrained("a").double().cpu().numpy(); b = eng.unconstrained("b").double().cpu().numpy()
from oracle import vi_oracle as vo
z = x @ a + b
m = (y != 255)
R = np.empty((N, J))
for lo in range(0, N, 4096):                       # the reference's semantics: zero gradient where P hits the clamp (App. A.1)
    sg = vo.sigmoid(z[lo:lo + 4096])
    lp, dP = vo.bernoulli_logprob_probs(sg, y[lo:lo + 4096])
    R[lo:lo + 4096] = dP * sg * (1 - sg)
GA = -(x.T @ R); Gb = -R.sum(0)
gA = eng.unconstrained("a", eng.G).double().cpu().numpy(); gb = eng.unconstrained("b", eng.G).double().cpu().numpy()[0]
free = eng.unconstrained("a", eng.free).cpu().numpy()
eA = (gA - GA) * free; eb = gb - Gb
print("max|GA|", np.abs(GA * free).max(), "max|Gb|", np.abs(Gb).max(), "sum|R| per item", np.abs(R).sum(0).mean())
print("err A max", np.abs(eA).max(), "rms", np.sqrt((eA ** 2).mean()), " err b max", np.abs(eb).max(), "rms", np.sqrt((eb**2).mean()), "mean", eb.mean())
print("err A by d-block of 8 (rms):", [float("%.3g" % np.sqrt((eA[8*i:8*i+8] ** 2).mean())) for i in range(13)])
print("err A by item chunk of 128 (rms):", [float("%.3g" % np.sqrt((eA[:, 128*i:128*i+128] ** 2).mean())) for i in range(4)])
print("corr(eb, Gb)", np.corrcoef(eb, Gb)[0, 1], "corr(eb, sum|R|)", np.corrcoef(eb, np.abs(R).sum(0))[0,1])
sat = (np.abs(z) > 17) & m
print("frac saturated cells", sat.mean(), "frac |R|==1-ish", (np.abs(np.abs(R) - 1) < 1e-7).mean())
gxT = eng.last["gxT"][:N * D].reshape(D, N).double().cpu().numpy()
gx = (R @ a.T - x)
print("gx err max", np.abs(gxT.T - gx).max(), "max|gx|", np.abs(gx).max())
# error if R were rounded to bf16 two-term
ll, g = vo.irt_loglik("irt_2pl", x[:2048], a, b, None, None, 1.0, y[:2048])
print("oracle gx vs mine", np.abs(g["x"] - (R[:2048] @ a.T)).max(), " oracle gx-x vs gxT", np.abs(g["x"] - x[:2048] - gxT.T[:2048]).max())
print("gxT[:3,:3]", gxT[:3, :3], "mine", gx[:3, :3].T)
print("ll err", np.abs(eng.last["ll"][:2048].double().cpu().numpy() - ll).max())

# what a two-term bf16 R (RTN each) would give for Gb, and a one-sided (truncated) second term
import struct
def bf16_rtn(v):
    u = v.astype(np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32).astype(np.float64)
R32 = R.astype(np.float32).astype(np.float64)
h1 = bf16_rtn(R32); h2 = bf16_rtn(R32 - h1)
print("Gb err of 2-term RTN R:", np.abs((h1 + h2).sum(0) - R.sum(0)).max(), " fp32 R:", np.abs(R32.sum(0) - R.sum(0)).max())
print("eb first 8", eb[:8], "Gb first 8", Gb[:8])
