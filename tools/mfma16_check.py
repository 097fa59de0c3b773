"""A/B of the 16-bit-MFMA kernels (f16x2 / bf16 terms: the default) against the fp32-MFMA ones (VX_MFMA16=0) and the shape-generic
kernels: each mode in a child process (the switches are read once per process).  Run on a GPU box."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import json, sys, numpy as np, torch
sys.path.insert(0, %r)
from vipsy_amd.engine import IrtEngine
N = int(sys.argv[1])
rng = np.random.RandomState(5)
J, D, H = 500, 100, 64
y = rng.randint(0, 2, size=(N, J)).astype(np.uint8); y[rng.rand(N, J) < 0.2] = 255
eng = IrtEngine(torch.from_numpy(y).cuda(), model="irt_2pl", D=D, amortized=True, H=H, seed=21)
eng.unconstrained("b").copy_(torch.from_numpy(0.5 * rng.randn(1, J)).float())
eng.loss_and_grads()
torch.cuda.synchronize()
np.save(sys.argv[2], eng.G[:eng.n_params + 1].double().cpu().numpy())
''' % ROOT
for N in (512, 520, 1000, 4096):
    g = {}
    for mode, extra in {"fp32": {"VX_MFMA16": "0"}, "mfma16": {"VX_MFMA16": "1"}, "generic": {"VX_FORCE_GENERIC": "1"}}.items():
        env = dict(os.environ, VX_FORCE_GENERIC="0"); env.update(extra)
        out = "/tmp/bfx_%s.npy" % mode
        p = subprocess.run([sys.executable, "-c", CODE, str(N), out], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        import numpy as np
        g[mode] = np.load(out)
    s = abs(g["generic"]).max()
    print("N=%d  |fp32-generic|=%.3g  |mfma16-generic|=%.3g  |mfma16-fp32|=%.3g  (scale %.3g)" % (
        N, abs(g["fp32"] - g["generic"]).max() / s, abs(g["mfma16"] - g["generic"]).max() / s,
        abs(g["mfma16"] - g["fp32"]).max() / s, s), flush=True)
