"""TEST / BENCH INFRASTRUCTURE, not the product path.  A plain PyTorch float32 restatement of one full-batch step of the
reference's amortized multidimensional model (vi.py:448-455 MvnEncoder, vi.py:596-625 model, vi.py:686-693 guide,
vi.py:505-516 step) with autograd and torch.optim.Adam, for bench.py's `cpu_baseline` leg (a
pure-PyTorch CPU restatement under torch.set_num_threads).  It materialises the (B, D, D) scale matrix as the reference
does.  Only bench.py imports it."""
import math
import time

import torch


def make_step(J, D, H, n, seed=0, lr=1e-3):
    g = torch.Generator().manual_seed(seed)
    T = D * (D + 1) // 2
    y = torch.randint(0, 2, (n, J), generator=g).float()
    P = {"W1": torch.randn(H, J, generator=g) / math.sqrt(J), "b1": torch.zeros(H),
         "W21": torch.randn(D, H, generator=g) / 8, "b21": torch.zeros(D),
         "W22": 0.1 * torch.randn(T, H, generator=g) / 8, "b22": torch.zeros(T),
         "a": torch.ones(D, J), "b": torch.zeros(1, J)}
    free = torch.ones(D, J)
    for i in range(D):
        free[i, J - i:] = 0                                   # vi.py:570-572
    P["a"] = P["a"] * free
    for v in P.values():
        v.requires_grad_(True)
    opt = torch.optim.Adam(list(P.values()), lr=lr)
    r, c = torch.tril_indices(D, D)
    eps32 = torch.finfo(torch.float32).eps

    def step():
        opt.zero_grad(set_to_none=True)
        eps = torch.randn(n, D, generator=g)
        h = torch.nn.functional.softplus(y @ P["W1"].t() + P["b1"])                     # vi.py:449
        loc = h @ P["W21"].t() + P["b21"]
        raw = h @ P["W22"].t() + P["b22"]
        M = torch.zeros(n, D, D)
        M[:, r, c] = raw                                                                 # vi.py:452-454
        diag = torch.diagonal(M, dim1=1, dim2=2)
        L = torch.tril(M, -1) + torch.diag_embed(torch.exp(diag))
        x = loc + (L @ eps.unsqueeze(-1)).squeeze(-1)
        logq = -0.5 * (eps ** 2).sum(1) - diag.sum(1)
        logp = -0.5 * (x ** 2).sum(1)
        Pz = torch.sigmoid(x @ P["a"] + P["b"]).clamp(eps32, 1 - eps32)                  # vi.py:41, clamp_probs
        ll = (y * torch.log(Pz) + (1 - y) * torch.log1p(-Pz)).sum(1)
        loss = -(ll + logp - logq).sum()
        loss.backward()
        P["a"].grad.mul_(free)                                                           # vi.py:511-512
        opt.step()
        return float(loss.detach())
    return step


def time_step(J, D, H, n, threads, min_seconds=6.0):
    torch.set_num_threads(int(threads))
    step = make_step(J, D, H, n)
    step()                                                    # warm-up (allocator, thread pool)
    reps, t0 = 0, time.perf_counter()
    while True:
        step()
        reps += 1
        el = time.perf_counter() - t0
        if el >= min_seconds:
            return el / reps
