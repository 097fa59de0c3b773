"""Step time of the amortized multivariate guide around the shape conditions of the f16x2 kernels (J % 4, D % 4, N % 8):
what a user with 'odd' sizes pays.   usage (GPU box): python tools/shape_cliffs.py [persons]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from vipsy_amd.engine import IrtEngine, LrSpec
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    dev = torch.device("cuda:0")
    for J, D in ((500, 100), (499, 100), (501, 100), (502, 100), (500, 99), (500, 98), (500, 96), (500, 64), (500, 32), (37, 100), (40, 8), (37, 7)):
        g = torch.Generator(device=dev); g.manual_seed(J * 1000 + D)
        y = (torch.rand(N, J, device=dev, generator=g) < 0.5).to(torch.uint8)
        eng = IrtEngine(y, model="irt_2pl", D=D, amortized=True, H=64, seed=3)
        lrs = LrSpec(lambda m, p: {"lr": 1e-3})
        eng.steps(lrs, [None] * 6)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 12
        eng.steps(lrs, [None] * n)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / n
        print("N = %d  J = %3d  D = %3d : %8.3f ms/step" % (N, J, D, ms), flush=True)
        del eng, y
    # the other guides: amortized 1-D (VaeIRT, x_feature 1) and per-person (VIRT) 1-D / multivariate
    for label, kw, shapes in (("amortized 1-D", dict(D=1, amortized=True, H=64), ((500, 1), (499, 1), (37, 1))),
                              ("per-person 1-D", dict(D=1), ((500, 1), (499, 1))),
                              ("per-person multivariate", dict(), ((40, 8), (37, 8), (40, 7)))):
        for J, D in shapes:
            g = torch.Generator(device=dev); g.manual_seed(J * 1000 + D)
            y = (torch.rand(N, J, device=dev, generator=g) < 0.5).to(torch.uint8)
            eng = IrtEngine(y, model="irt_2pl", seed=3, **dict(kw, D=D))
            lrs = LrSpec(lambda m, p: {"lr": 1e-3})
            eng.steps(lrs, [None] * 6)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 12
            eng.steps(lrs, [None] * n)
            torch.cuda.synchronize()
            print("%-24s N = %d  J = %3d  D = %3d : %8.3f ms/step" % (label, N, J, D, 1e3 * (time.perf_counter() - t0) / n), flush=True)
            del eng, y


if __name__ == "__main__":
    main()
