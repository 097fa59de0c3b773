"""CPU oracle: a numpy restatement of the reference's ELBO-gradient step.  TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product path (vipsy_amd/) never does and fails loudly when the HIP library is missing.

What it restates (all citations into /root/reference):
  * response functions            vi.py:22-66 (irt_1pl..4pl), vi.py:69-83 (dina)
  * missing-data mask             vi.py:617-625
  * IRT model log-density         vi.py:574-615
  * guides (BBVI / amortized)     vi.py:673-723, encoders vi.py:417-455
  * HO-DINA model / guide         vi.py:897-934, 968-981; pattern table vi.py:825-837
  * SVI.step + free mask          vi.py:503-516, identification constraints vi.py:566-572
  * the estimator and optimiser that live in the un-vendored dependency pyro-ppl==1.4.0
    (requirements.txt:1): Trace_ELBO (pathwise for Normal/MVN guides), TraceEnum_ELBO (exact
    enumeration), pyro.optim.Adam = one torch.optim.Adam per tensor, MultiStepLR stepped every
    iteration -- restated from the published algorithm.

PARITY STATUS: pinned against tests/golden/*.npz, which were produced by executing the reference's
own model()/guide()/SVI.step code (tests/golden/make_golden.py) -- but under a minimal stand-in for
pyro (tests/golden/_gen/pyro_shim), because pyro-ppl cannot be installed here and the reference's
tests hold no golden values.  So: pinned for everything vi.py itself defines,
"parity unpinned" for pyro-ppl 1.4.0 internals.  Independent pins: analytic-vs-autograd gradients,
ELBO <= Gauss-Hermite log-marginal, and the LSAT-6 known answer (tests/test_oracle_*.py).

All gradients are of the LOSS (= -ELBO) with respect to the UNCONSTRAINED leaves, exactly what
vi.py:508-514 hands to the optimiser.
"""
import math

import numpy as np

EPS32 = float(np.finfo(np.float32).eps)          # torch clamp_probs epsilon for float32
LOG_2PI = math.log(2.0 * math.pi)

# ------------------------------------------------------------------------------------------------
# Counter-based RNG shared with the HIP kernels (vipsy_amd/csrc/philox.h): Philox4x32-10 + Box-Muller
# key = (seed_lo, seed_hi); counter = (gid_lo, gid_hi, step, (stream << 16) | block)
# block b yields normals for dims 4b..4b+3 of person `gid`.
# ------------------------------------------------------------------------------------------------
_PH_M0 = np.uint64(0xD2511F53)
_PH_M1 = np.uint64(0xCD9E8D57)
_PH_W0 = np.uint32(0x9E3779B9)
_PH_W1 = np.uint32(0xBB67AE85)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10.  Inputs are uint32 arrays (broadcastable); returns 4 uint32 arrays."""
    c0 = np.asarray(c0, dtype=np.uint32)
    c1 = np.asarray(c1, dtype=np.uint32)
    c2 = np.asarray(c2, dtype=np.uint32)
    c3 = np.asarray(c3, dtype=np.uint32)
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = c0.astype(np.uint64) * _PH_M0
            p1 = c2.astype(np.uint64) * _PH_M1
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = p0.astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = p1.astype(np.uint32)
            n0 = hi1 ^ c1 ^ k0
            n1 = lo1
            n2 = hi0 ^ c3 ^ k1
            n3 = lo0
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0 = np.uint32((int(k0) + int(_PH_W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(_PH_W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def _u01(x):
    """uint32 -> float32 in (0,1): ((x >> 8) + 0.5) * 2^-24 (exact in float32)."""
    return ((x >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -24)


def philox_normals(seed, step, stream, gids, D):
    """eps[i, d] for global person ids `gids` (int64) -- float32 [len(gids), D].

    Box-Muller on the four Philox words of block d // 4: (w0,w1) -> dims 4b, 4b+1; (w2,w3) -> 4b+2,
    4b+3.  Independent of how persons are sharded over GPUs (keyed by GLOBAL id)."""
    gids = np.asarray(gids, dtype=np.int64)
    nb = (D + 3) // 4
    blocks = np.arange(nb, dtype=np.uint32)[None, :]
    g_lo = (gids & 0xFFFFFFFF).astype(np.uint32)[:, None]
    g_hi = ((gids >> 32) & 0xFFFFFFFF).astype(np.uint32)[:, None]
    c3 = (np.uint32(stream) << np.uint32(16)) | blocks
    w0, w1, w2, w3 = philox4x32_10(g_lo, g_hi, np.uint32(step), c3,
                                   np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF))
    out = np.empty((len(gids), nb * 4), dtype=np.float32)
    two_pi = np.float32(2.0 * math.pi)
    for k, (ua, ub) in enumerate(((w0, w1), (w2, w3))):
        r = np.sqrt(np.float32(-2.0) * np.log(_u01(ua)))
        th = two_pi * _u01(ub)
        out[:, 2 * k::4] = r * np.cos(th)
        out[:, 2 * k + 1::4] = r * np.sin(th)
    return out[:, :D]


def philox_uniforms(seed, step, stream, gids, K):
    """u[i, k] in (0, 1) for global person ids `gids`: word k % 4 of Philox block k // 4 (the draw rule of the
    Bernoulli-guide kernel, vipsy_amd/csrc/k_cdm_sf.hip: attr_ik = u_ik < p_ik) -- float32 [len(gids), K]."""
    gids = np.asarray(gids, dtype=np.int64)
    nb = (K + 3) // 4
    blocks = np.arange(nb, dtype=np.uint32)[None, :]
    g_lo = (gids & 0xFFFFFFFF).astype(np.uint32)[:, None]
    g_hi = ((gids >> 32) & 0xFFFFFFFF).astype(np.uint32)[:, None]
    c3 = (np.uint32(stream) << np.uint32(16)) | blocks
    w = philox4x32_10(g_lo, g_hi, np.uint32(step), c3, np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF))
    out = np.empty((len(gids), nb * 4), dtype=np.float32)
    for k in range(4):
        out[:, k::4] = _u01(w[k])
    return out[:, :K]


def _philox_word_uniform(seed, stream, gids, idx):
    """u01 of word idx % 4 of Philox block idx // 4 (step 0, given stream) for every (person, idx): [len(gids), len(idx)]."""
    gids = np.asarray(gids, dtype=np.int64)
    idx = np.asarray(idx, dtype=np.int64)
    g_lo = (gids & 0xFFFFFFFF).astype(np.uint32)[:, None]
    g_hi = ((gids >> 32) & 0xFFFFFFFF).astype(np.uint32)[:, None]
    c3 = (np.uint32(stream) << np.uint32(16)) | (idx >> 2).astype(np.uint32)[None, :]
    w = philox4x32_10(g_lo, g_hi, np.uint32(0), c3, np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF))
    sel = (idx & 3)[None, :]
    word = np.where(sel == 0, w[0], np.where(sel == 1, w[1], np.where(sel == 2, w[2], w[3])))
    return _u01(word)


def synth_irt(model, seed, gids, a, b, c, d, Dc=1.0, missing=0.0, x=None):
    """Restatement of vipsy_amd/csrc/k_synth.hip::k_synth_irt (streams 0xE0 latent, 0xD1 response, 0xD2 missing)."""
    J = b.shape[-1]
    D = 1 if a is None else a.shape[0]
    if x is None:
        x = philox_normals(seed, 0, 0xE0, gids, D).astype(np.float64)
    z = x @ a if a is not None else np.repeat(x[:, :1], J, axis=1)
    P = sigmoid(Dc * (z + b.reshape(1, J)))
    cc = 0.0 if c is None else c.reshape(1, J)
    dd = 1.0 if d is None else d.reshape(1, J)
    P = cc + (dd - cc) * P
    u = _philox_word_uniform(seed, 0xD1, gids, np.arange(J))
    y = (u < P).astype(np.uint8)
    if missing > 0:
        y[_philox_word_uniform(seed, 0xD2, gids, np.arange(J)) < missing] = 255
    return y, x, P, u


# ------------------------------------------------------------------------------------------------
# small math helpers
# ------------------------------------------------------------------------------------------------
def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def softplus(x):
    return np.maximum(x, 0) + np.log1p(np.exp(-np.abs(x)))


def logit(p):
    return np.log(p) - np.log1p(-p)


def bernoulli_logprob_probs(P, y_u8, eps=EPS32):
    """torch.distributions.Bernoulli(probs=P).log_prob(y) as pyro evaluates vi.py:595/615/923,
    with the missing-cell rule of vi.py:621-624 (y:=0, P:=0 where NaN; uint8 255 here).

    Returns (lp, dlp_dP): clamp_probs -> logits = log(P~) - log1p(-P~) -> -BCE_with_logits.
    dlp/dP is 0 where P is outside [eps, 1-eps] (clamp) and on missing cells."""
    miss = (y_u8 == 255)
    y = np.where(miss, 0, y_u8).astype(P.dtype)
    Pm = np.where(miss, 0.0, P)
    Pc = np.clip(Pm, eps, 1.0 - eps)
    lg = np.log(Pc) - np.log1p(-Pc)
    lp = y * lg - np.maximum(lg, 0) - np.log1p(np.exp(-np.abs(lg)))
    inside = (Pm >= eps) & (Pm <= 1.0 - eps) & (~miss)
    dlp = np.where(inside, (y - Pc) / (Pc * (1.0 - Pc)), 0.0)
    return lp, dlp


def all_attrs(K, dtype=np.float64):
    """vi.py:825-837: pattern c has attribute k iff bit k of c is set (LSB first)."""
    c = np.arange(2 ** K)[:, None]
    return ((c >> np.arange(K)[None, :]) & 1).astype(dtype)


def tril_rows_cols(D):
    """Row-major lower-triangle order (0,0),(1,0),(1,1),(2,0).. == torch.tril_indices (vi.py:453)."""
    r, c = np.tril_indices(D)
    return r, c


# ------------------------------------------------------------------------------------------------
# IRT likelihood block (vi.py:22-66, 617-625) with analytic derivatives
# ------------------------------------------------------------------------------------------------
def irt_loglik(model, x, a, b, c, d, Dc, y_u8):
    """Returns ll_i (B,), and d(sum ll)/d{x (B,D), a (D,J), b (1,J), c (1,J), d (1,J)} (constrained)."""
    if model == "irt_1pl":
        z = Dc * (x + b)                               # vi.py:29 (x is (B,1), broadcast over items)
    else:
        z = Dc * (x @ a + b)                           # vi.py:41
    sg = sigmoid(z)
    lo = c if model in ("irt_3pl", "irt_4pl") else 0.0
    hi = d if model == "irt_4pl" else 1.0
    P = lo + (hi - lo) * sg                            # vi.py:53, 66
    lp, dP = bernoulli_logprob_probs(P, y_u8)
    ll = lp.sum(axis=1)
    dz = dP * (hi - lo) * sg * (1.0 - sg)
    g = {"b": (Dc * dz).sum(axis=0, keepdims=True)}
    if model == "irt_1pl":
        g["x"] = (Dc * dz).sum(axis=1, keepdims=True)
    else:
        g["x"] = (Dc * dz) @ a.T
        g["a"] = x.T @ (Dc * dz)
    if model in ("irt_3pl", "irt_4pl"):
        g["c"] = (dP * (1.0 - sg)).sum(axis=0, keepdims=True)
    if model == "irt_4pl":
        g["d"] = (dP * sg).sum(axis=0, keepdims=True)
    return ll, g


# ------------------------------------------------------------------------------------------------
# Encoders (vi.py:417-455): forward + backward
# ------------------------------------------------------------------------------------------------
def enc_input(y_u8, dtype):
    """vi.py:680-682 / 689-691: NaN -> -1."""
    return np.where(y_u8 == 255, -1.0, y_u8.astype(dtype)).astype(dtype)


def enc_forward(W, yin):
    """W: dict with fc1.weight (H,J), fc1.bias, fc21.*, fc22.*.  Returns (loc, raw22, cache)."""
    pre = yin @ W["fc1.weight"].T + W["fc1.bias"]
    h = softplus(pre)
    loc = h @ W["fc21.weight"].T + W["fc21.bias"]
    raw = h @ W["fc22.weight"].T + W["fc22.bias"]
    return loc, raw, (yin, pre, h)


def enc_backward(W, cache, g_loc, g_raw):
    yin, pre, h = cache
    g = {"fc21.weight": g_loc.T @ h, "fc21.bias": g_loc.sum(0),
         "fc22.weight": g_raw.T @ h, "fc22.bias": g_raw.sum(0)}
    g_h = g_loc @ W["fc21.weight"] + g_raw @ W["fc22.weight"]
    g_pre = g_h * sigmoid(pre)
    g["fc1.weight"] = g_pre.T @ yin
    g["fc1.bias"] = g_pre.sum(0)
    return g


ENC_KEYS = ("fc1.weight", "fc1.bias", "fc21.weight", "fc21.bias", "fc22.weight", "fc22.bias")


# ------------------------------------------------------------------------------------------------
# One particle of the IRT ELBO -> loss and grads w.r.t. unconstrained leaves
# ------------------------------------------------------------------------------------------------
def mvn_score_terms(L, eps):
    """d log q(x) / d loc (B, D) and / d M (B, D, D; M the unconstrained matrix: strict lower part = L's, diagonal = log L_kk)
    of q = MultivariateNormal(loc, scale_tril = L) at the FIXED point x = loc + L eps:
    log q = -sum_k log L_kk - 0.5 |L^-1 (x - loc)|^2 + const  =>  u = L^-T eps, d/d loc = u, d/d L_kc = u_k eps_c (c <= k)
    - [k == c] / L_kk, and through L_kk = exp(M_kk): d/d M_kk = u_k eps_k L_kk - 1."""
    B, D = eps.shape
    u = np.stack([np.linalg.solve(L[i].T, eps[i]) for i in range(B)])
    sM = np.tril(np.einsum("bi,bj->bij", u, eps), -1)
    dg = u * eps * np.einsum("bii->bi", L) - 1.0
    sM = sM + np.einsum("bi,ij->bij", dg, np.eye(D, dtype=eps.dtype))
    return u, sM


def irt_particle(spec, params, y_u8_full, idx, eps, baseline=None, want_log_r=False):
    """spec: dict(model, D, Dc, N, amortized, share_cov, a_free (D,J bool or None)[, estimator]).
    params: unconstrained leaves keyed by the reference's param-store names.
    idx: (B,) int64 rows of the plate subsample; eps: (B,D) standard normal draws.
    spec["estimator"] == "score" (SURVEY.md App. A.5 -- NOT what the reference does for its Normal guides, vi.py:684,693,
    705,715,723 are reparameterised): the guide's gradient is the score-function one, (log_r_i - baseline_i) d log q / d phi
    with log_r_i = scale (ll_i + log p(x_i) - log q(x_i)) detached and the score term unscaled; the item gradients stay
    pathwise.  For D > 1, x = loc + L eps held fixed: d log q / d loc = u = L^-T eps, d log q / d L_kc = u_k eps_c (c < k),
    d log q / d M_kk = u_k eps_k L_kk - 1 (mvn_score_terms).  baseline: (B,) control variate or None.
    Returns (loss, grads) for ONE particle (not yet divided by num_particles); with want_log_r also log_r (B,)."""
    model, D, Dc, N = spec["model"], spec["D"], spec["Dc"], spec["N"]
    dt = params["b"].dtype
    B = len(idx)
    scale = dt.type(N) / dt.type(B)
    y = y_u8_full[idx]
    eps = eps.astype(dt)
    b = params["b"]
    a = params.get("a")
    c = sigmoid(params["c"]) if "c" in params else None
    d = sigmoid(params["d"]) if "d" in params else None
    grads = {}

    # ---- guide: x = loc + L eps ------------------------------------------------------------
    if spec["amortized"]:
        W = {k: params["encoder$$$" + k] for k in ENC_KEYS}
        loc, raw, cache = enc_forward(W, enc_input(y, dt))
    else:
        loc = params["x_local"][idx]
        if D == 1:
            raw = params["x_scale"][idx]                       # log sigma (constraints.positive)
        elif spec["share_cov"]:
            raw = params["x_scale"]                             # (D,D) unconstrained
        else:
            raw = params["x_scale"][idx]                        # (B,D,D)
    if D == 1:
        sig = np.exp(raw)
        x = loc + sig * eps
        logq = -0.5 * eps[:, 0] ** 2 - raw[:, 0] - 0.5 * LOG_2PI
    else:
        if spec["amortized"]:
            r_, c_ = tril_rows_cols(D)
            M = np.zeros((B, D, D), dtype=dt)
            M[:, r_, c_] = raw                                   # vi.py:452-454
        elif spec["share_cov"]:
            M = np.broadcast_to(raw, (B, D, D))
        else:
            M = raw
        diag = np.einsum("bii->bi", M)
        L = np.tril(M, -1) + np.einsum("bi,ij->bij", np.exp(diag), np.eye(D, dtype=dt))
        x = loc + np.einsum("bij,bj->bi", L, eps)
        logq = -0.5 * (eps ** 2).sum(1) - diag.sum(1) - 0.5 * D * LOG_2PI
    logp_x = -0.5 * (x ** 2).sum(1) - 0.5 * D * LOG_2PI

    # ---- model ----------------------------------------------------------------------------------
    ll, g = irt_loglik(model, x, a, b, c, d, Dc, y)
    elbo = scale * (ll + logp_x - logq).sum()
    loss = -elbo

    grads["b"] = -scale * g["b"]
    if "a" in g:
        ga = -scale * g["a"]
        if spec.get("a_free") is not None:
            ga = ga * spec["a_free"].astype(dt)                  # vi.py:511-512
        grads["a"] = ga
    if "c" in g:
        grads["c"] = -scale * g["c"] * c * (1 - c)
    if "d" in g:
        grads["d"] = -scale * g["d"] * d * (1 - d)

    gx = scale * (g["x"] - x)                                    # d ELBO / d x (likelihood + prior)
    g_loc = gx
    log_r = scale * (ll + logp_x - logq)
    if spec.get("estimator", "pathwise") == "score" and D > 1:
        f = (log_r - (0.0 if baseline is None else np.asarray(baseline, dt)))
        s_loc, s_M = mvn_score_terms(L, eps)
        g_loc = f[:, None] * s_loc
        g_raw = f[:, None, None] * s_M
    elif spec.get("estimator", "pathwise") == "score":
        f = (log_r - (0.0 if baseline is None else np.asarray(baseline, dt)))[:, None]
        g_loc = f * eps / sig                                    # d log q / d loc = (x - loc) / sigma^2 = eps / sigma
        g_raw = f * (eps ** 2 - 1.0)                             # d log q / d raw (sigma = exp(raw))
    elif D == 1:
        g_raw = gx * sig * eps + scale                           # + scale from -logq
    else:
        gM = np.einsum("bi,bj->bij", gx, eps)
        gM = np.tril(gM)
        dg = np.einsum("bii->bi", gM) * np.exp(diag) + scale
        gM = np.tril(gM, -1) + np.einsum("bi,ij->bij", dg, np.eye(D, dtype=dt))
        g_raw = gM
    if spec["amortized"]:
        if D > 1:
            g_raw = g_raw[:, r_, c_]
        ge = enc_backward(W, cache, -g_loc, -g_raw)
        for k in ENC_KEYS:
            grads["encoder$$$" + k] = ge[k]
    else:
        gl = np.zeros_like(params["x_local"])
        np.add.at(gl, idx, -g_loc)
        grads["x_local"] = gl
        if D > 1 and spec["share_cov"]:
            grads["x_scale"] = -g_raw.sum(0)
        else:
            gs = np.zeros_like(params["x_scale"])
            np.add.at(gs, idx, -g_raw)
            grads["x_scale"] = gs
    if want_log_r:
        return loss, grads, log_r
    return loss, grads


# ------------------------------------------------------------------------------------------------
# HO-DINA with exact enumeration (vi.py:897-934)
# ------------------------------------------------------------------------------------------------
def dina_eta(K, q):
    """eta[c, j] = 1 iff pattern c masters every attribute item j requires (vi.py:78-81)."""
    al = all_attrs(K, q.dtype)
    yita = al @ q
    aa = (q ** 2).sum(axis=0)
    return (yita == aa).astype(q.dtype), al


def dino_eta(K, q):
    """eta[c, j] of the reference's dino() (vi.py:86-101), INCLUDING its in-place sequencing: yita = (1 - attr) q;
    `yita[yita < aa] = 1` runs first, then `yita[yita == aa] = 0` -- which also clears the entries the first statement
    has just set to 1 whenever aa_j == 1.  Net effect: eta = 1 iff pattern c masters at least one attribute of item j
    AND item j requires at least two attributes; single-attribute items always get eta = 0."""
    al = all_attrs(K, q.dtype)
    aa = np.broadcast_to((q ** 2).sum(axis=0), (al.shape[0], q.shape[1]))
    yita = (1 - al) @ q
    yita[yita < aa] = 1
    yita[yita == aa] = 0
    return yita.astype(q.dtype), al


def ccdm_particle(spec, params, y_u8_full, idx, eps=None):
    """Pattern-enumerated DINA / DINO with the uniform pattern prior (VCCDM.model, vi.py:840-859; empty guide,
    TraceEnum_ELBO): ELBO = scale * sum_i log sum_c (1/C) prod_j Bern(y_ij | p_cj)."""
    K, N = spec["K"], spec["N"]
    dt = params["g"].dtype
    q = spec["q"].astype(dt)
    B = len(idx)
    scale = dt.type(N) / dt.type(B)
    y = y_u8_full[idx]
    eta, _ = (dino_eta if spec.get("cdm", "dina") == "dino" else dina_eta)(K, q)      # (C,J)
    C = eta.shape[0]
    g_ = sigmoid(params["g"])
    s_ = sigmoid(params["s"])
    pr = np.full((1, C), 1.0 / C, dt)                             # Categorical(probs = 1/C): renormalise, clamp, log
    pr = pr / pr.sum(axis=1, keepdims=True)
    lg = np.log(np.clip(pr, EPS32, 1 - EPS32))
    lp0, d0 = bernoulli_logprob_probs(np.broadcast_to(g_, y.shape).astype(dt), y)
    lp1, d1 = bernoulli_logprob_probs(np.broadcast_to(1 - s_, y.shape).astype(dt), y)
    Bc = lp0.sum(1, keepdims=True) + (lp1 - lp0) @ eta.T          # (B,C)
    f = lg + Bc
    fmax = f.max(axis=1, keepdims=True)
    lse = fmax[:, 0] + np.log(np.exp(f - fmax).sum(axis=1))
    elbo = scale * lse.sum()
    r = np.exp(f - lse[:, None])
    E = r @ eta
    grads = {
        "g": -scale * ((1 - E) * d0).sum(0, keepdims=True) * g_ * (1 - g_),
        "s": -scale * (-(E * d1)).sum(0, keepdims=True) * s_ * (1 - s_),
    }
    return -elbo, grads


# ------------------------------------------------------------------------------------------------
# Bernoulli-guide CDMs: the score-function (REINFORCE) estimator (VCDM / VaeCDM, vi.py:726-816; BinEncoder :458-470)
# ------------------------------------------------------------------------------------------------
TINY32 = float(np.finfo(np.float32).tiny)
CDM_REF_PRIOR = 1.5         # vi.py:753: Bernoulli(torch.ones(..) + 0.5), i.e. probs = 1.5 (validation is off in pyro 1.4)


def cdm_eta_rows(attr, q, cdm):
    """eta[i, j] for explicit attribute rows (vi.py:69-100), the DINO variant with the reference's in-place sequencing
    (see dino_eta)."""
    aa = np.broadcast_to((q ** 2).sum(axis=0), (attr.shape[0], q.shape[1]))
    if cdm == "dino":
        yita = (1 - attr) @ q
        yita[yita < aa] = 1
        yita[yita == aa] = 0
        return yita
    yita = attr @ q
    return (yita == aa).astype(q.dtype)


def bin_enc_forward(W, yin):
    """BinEncoder (vi.py:458-470): softplus(fc1) -> sigmoid(fc2); returns (logits of the attribute probabilities, cache)."""
    pre = yin @ W["fc1.weight"].T + W["fc1.bias"]
    h = softplus(pre)
    u = h @ W["fc2.weight"].T + W["fc2.bias"]
    return u, (yin, pre, h)


def bin_enc_backward(W, cache, g_u):
    yin, pre, h = cache
    g_h = g_u @ W["fc2.weight"]
    g_pre = g_h * sigmoid(pre)
    return {"fc2.weight": g_u.T @ h, "fc2.bias": g_u.sum(0), "fc1.weight": g_pre.T @ yin, "fc1.bias": g_pre.sum(0)}


def cdm_sf_particle(spec, params, y_u8_full, idx, attr, baseline=None):
    """One Trace_ELBO particle of VCDM / VaeCDM with the score-function estimator:

        log_r_i = scale [ log p(attr_i) + log p(y_i | attr_i) - log q(attr_i) ]          (kept per plate index, detached)
        loss    = - sum_i log_r_i
        d loss / d (item leaves)  : pathwise through scale log p(y_i | attr_i)
        d loss / d (guide logits) = - (log_r_i - baseline_i) d log q(attr_i) / d logits    (score term UNSCALED)

    attr: the (B, K) 0/1 draws of the guide (so that reference, oracle and HIP replay the same sample); baseline: (B,)
    control variate or None (pyro's Trace_ELBO has none).  spec["attr_prior"]: None = the reference's Bernoulli(1.5).
    Returns (loss, grads, log_r)."""
    K, N = spec["K"], spec["N"]
    dt = params["g"].dtype
    q = spec["q"].astype(dt)
    B = len(idx)
    scale = dt.type(N) / dt.type(B)
    y = y_u8_full[idx]
    if (y == 255).any():
        raise ValueError("VCDM / VaeCDM pass the responses unmasked (vi.py:756): a NaN cell makes the reference's loss NaN")
    attr = np.asarray(attr, dt).reshape(B, K)
    amort = spec["amortized"]
    if amort:
        W = {k.split("$$$")[1]: v for k, v in params.items() if k.startswith("encoder$$$")}
        u, cache = bin_enc_forward(W, y.astype(dt))                  # vi.py:800: data[idx] as it is (0 / 1)
        p = sigmoid(u)
        inside_t = np.ones_like(p, dtype=bool)
    else:
        u = params["attr_p"][idx]
        ps = sigmoid(u)                                            # constraints.unit_interval -> SigmoidTransform, clamped
        p = np.clip(ps, TINY32, 1.0 - EPS32)
        inside_t = (ps >= TINY32) & (ps <= 1.0 - EPS32)
    pc = np.clip(p, EPS32, 1.0 - EPS32)                            # Bernoulli(probs).log_prob: clamp_probs
    inside = inside_t & (p >= EPS32) & (p <= 1.0 - EPS32)
    lq = (attr * np.log(pc) + (1 - attr) * np.log1p(-pc)).sum(1)
    prior = CDM_REF_PRIOR if spec.get("attr_prior") is None else spec["attr_prior"]
    pp = np.clip(dt.type(prior), EPS32, 1.0 - EPS32)
    lpa = (attr * np.log(pp) + (1 - attr) * np.log1p(-pp)).sum(1)
    eta = cdm_eta_rows(attr, q, spec.get("cdm", "dina"))
    g_, s_ = sigmoid(params["g"]), sigmoid(params["s"])
    P = (1 - s_) ** eta * g_ ** (1 - eta)
    lp, dlp = bernoulli_logprob_probs(P.astype(dt), y)
    ll = lp.sum(1)
    log_r = scale * (lpa + ll - lq)
    loss = -log_r.sum()
    f = log_r - (0.0 if baseline is None else np.asarray(baseline, dt))
    g_u = -(f[:, None]) * np.where(inside, attr - pc, 0.0)           # d log q / d logit = attr - p inside the clamps
    grads = {"g": -scale * (dlp * (1 - eta)).sum(0, keepdims=True) * g_ * (1 - g_),
             "s": -scale * (-(dlp * eta)).sum(0, keepdims=True) * s_ * (1 - s_)}
    if amort:
        for k, v in bin_enc_backward(W, cache, g_u).items():
            grads["encoder$$$" + k] = v
    else:
        ga = np.zeros_like(params["attr_p"])
        np.add.at(ga, idx, g_u)
        grads["attr_p"] = ga
    return loss, grads, log_r


def init_cdm_sf_params(spec, J, dtype=np.float32, encoder=None):
    K, N = spec["K"], spec["N"]
    p = {"g": np.full((1, J), logit(np.asarray(0.1, dtype)), dtype), "s": np.full((1, J), logit(np.asarray(0.1, dtype)), dtype)}
    if spec["amortized"]:
        for k, v in encoder.items():
            p["encoder$$$" + k] = np.asarray(v, dtype)
    else:
        p["attr_p"] = np.zeros((N, K), dtype)                        # unit_interval^-1 (0.5) = 0 (vi.py:811)
    return p


def vaeccdm_particle(spec, params, y_u8_full, idx, eps=None):
    """VaeCCDM (vi.py:866-891, SoftmaxEncoder vi.py:473-485), TraceEnum_ELBO, as the reference computes it:
      * the pattern prior of person i is Categorical(attr_p[i]) with attr_p = softmax(fc2(relu(fc1(data_))), dim=0) --
        the softmax runs over the BATCH (vi.py:478,483), so every column of attr_p sums to one over the persons of the
        subsample; Categorical renormalises each row and clamps it to [eps, 1 - eps] before the log;
      * missing responses are replaced by -1 for the encoder AND stay -1 in `obs` (vi.py:882-891: no mask), so a missing cell
        contributes Bernoulli(p).log_prob(-1) = -logit(p) - softplus(logit(p)).
    ELBO = scale sum_i log sum_c pi_ic prod_j exp(lp_icj)."""
    K, N = spec["K"], spec["N"]
    dt = params["g"].dtype
    q = spec["q"].astype(dt)
    B = len(idx)
    scale = dt.type(N) / dt.type(B)
    y = y_u8_full[idx]
    v = np.where(y == 255, -1.0, y.astype(dt)).astype(dt)          # encoder input and observation alike
    W = {k.split("$$$")[1]: p for k, p in params.items() if k.startswith("encoder$$$")}
    pre = v @ W["fc1.weight"].T + W["fc1.bias"]
    h = np.maximum(pre, 0)
    z = h @ W["fc2.weight"].T + W["fc2.bias"]                       # (B, C)
    zm = z.max(axis=0, keepdims=True)
    ez = np.exp(z - zm)
    a = ez / ez.sum(axis=0, keepdims=True)                          # softmax over the batch
    R = a.sum(axis=1, keepdims=True)
    pi = a / R
    ins = (pi >= EPS32) & (pi <= 1 - EPS32)
    lg = np.log(np.clip(pi, EPS32, 1 - EPS32))
    eta, _ = (dino_eta if spec.get("cdm", "dina") == "dino" else dina_eta)(K, q)      # (C, J)
    g_, s_ = sigmoid(params["g"]), sigmoid(params["s"])

    def lp_const(P):
        Pc = np.clip(P, EPS32, 1 - EPS32)
        l = np.log(Pc) - np.log1p(-Pc)
        lp = v * l - softplus(l)
        inside = (P >= EPS32) & (P <= 1 - EPS32)
        return lp, np.where(inside, (v - Pc) / (Pc * (1 - Pc)), 0.0)
    lp0, d0 = lp_const(np.broadcast_to(g_, v.shape).astype(dt))
    lp1, d1 = lp_const(np.broadcast_to(1 - s_, v.shape).astype(dt))
    Bc = lp0.sum(1, keepdims=True) + (lp1 - lp0) @ eta.T
    f = lg + Bc
    fmax = f.max(axis=1, keepdims=True)
    lse = fmax[:, 0] + np.log(np.exp(f - fmax).sum(axis=1))
    elbo = scale * lse.sum()
    r = np.exp(f - lse[:, None])
    E = r @ eta
    grads = {"g": -scale * ((1 - E) * d0).sum(0, keepdims=True) * g_ * (1 - g_),
             "s": -scale * (-(E * d1)).sum(0, keepdims=True) * s_ * (1 - s_)}
    G = scale * np.where(ins, r, 0.0)                               # d ELBO / d log pi
    H = G - pi * G.sum(axis=1, keepdims=True)                       # d ELBO / d log a   (row normalisation)
    gz = H - a * H.sum(axis=0, keepdims=True)                       # d ELBO / d z       (softmax over the batch)
    g_h = gz @ W["fc2.weight"]
    g_pre = g_h * (pre > 0)
    grads["encoder$$$fc2.weight"] = -(gz.T @ h)
    grads["encoder$$$fc2.bias"] = -gz.sum(0)
    grads["encoder$$$fc1.weight"] = -(g_pre.T @ v)
    grads["encoder$$$fc1.bias"] = -g_pre.sum(0)
    return -elbo, grads


def hodina_particle(spec, params, y_u8_full, idx, eps):
    K, N = spec["K"], spec["N"]
    dt = params["g"].dtype
    q = spec["q"].astype(dt)
    B = len(idx)
    scale = dt.type(N) / dt.type(B)
    y = y_u8_full[idx]
    eps = eps.astype(dt).reshape(B, 1)
    eta, al = dina_eta(K, q)                                      # (C,J), (C,K)
    g_ = sigmoid(params["g"])
    s_ = sigmoid(params["s"])
    lam0 = params["lam0"]
    lam1 = np.exp(params["lam1"])
    if spec["amortized"]:
        W = {k: params["encoder$$$" + k] for k in ENC_KEYS}
        loc, raw, cache = enc_forward(W, enc_input(y, dt))
    else:
        loc = params["theta_local"][idx]
        raw = params["theta_scale"][idx]
    sig = np.exp(raw)
    th = loc + sig * eps                                          # (B,1)
    logq = -0.5 * eps[:, 0] ** 2 - raw[:, 0] - 0.5 * LOG_2PI
    logp_th = -0.5 * th[:, 0] ** 2 - 0.5 * LOG_2PI
    t = th @ lam1 + lam0                                          # (B,K) vi.py:911
    pi = sigmoid(t)
    A = np.log(pi) @ al.T + np.log(1 - pi) @ (1 - al.T)           # (B,C) vi.py:912 (log of it)
    pr = np.exp(A)
    pr = pr / pr.sum(axis=1, keepdims=True)                       # Categorical(probs) renormalises
    inside = (pr >= EPS32) & (pr <= 1 - EPS32)
    lg = np.log(np.clip(pr, EPS32, 1 - EPS32))                    # probs_to_logits clamp
    # item likelihood under each pattern: p_cj in {g_j, 1 - s_j}
    lp0, d0 = bernoulli_logprob_probs(np.broadcast_to(g_, y.shape).astype(dt), y)
    lp1, d1 = bernoulli_logprob_probs(np.broadcast_to(1 - s_, y.shape).astype(dt), y)
    Bc = lp0.sum(1, keepdims=True) + (lp1 - lp0) @ eta.T          # (B,C)
    f = lg + Bc
    fmax = f.max(axis=1, keepdims=True)
    lse = fmax[:, 0] + np.log(np.exp(f - fmax).sum(axis=1))
    elbo = scale * (lse + logp_th - logq).sum()
    r = np.exp(f - lse[:, None])                                  # responsibilities
    # d/dA through the normalised, clamped categorical
    ri = r * inside
    rho = ri - pr * ri.sum(axis=1, keepdims=True)
    tau = rho @ al - pi * rho.sum(axis=1, keepdims=True)          # (B,K)
    E = r @ eta                                                   # (B,J) expected eta
    grads = {
        "lam0": -scale * tau.sum(0, keepdims=True),
        "lam1": -scale * (tau * th).sum(0, keepdims=True) * lam1,
        "g": -scale * ((1 - E) * d0).sum(0, keepdims=True) * g_ * (1 - g_),
        "s": -scale * (-(E * d1)).sum(0, keepdims=True) * s_ * (1 - s_),
    }
    gth = scale * ((tau * lam1).sum(1, keepdims=True) - th)
    g_loc = gth
    g_raw = gth * sig * eps + scale
    if spec["amortized"]:
        ge = enc_backward(W, cache, -g_loc, -g_raw)
        for k in ENC_KEYS:
            grads["encoder$$$" + k] = ge[k]
    else:
        gl = np.zeros_like(params["theta_local"])
        np.add.at(gl, idx, -g_loc)
        gs = np.zeros_like(params["theta_scale"])
        np.add.at(gs, idx, -g_raw)
        grads["theta_local"], grads["theta_scale"] = gl, gs
    return -elbo, grads


# ------------------------------------------------------------------------------------------------
# loss_and_grads over particles + optimiser
# ------------------------------------------------------------------------------------------------
def loss_and_grads(spec, params, y_u8, idx_list, eps_list):
    if spec.get("family") == "cdm_sf":                      # eps_list carries the guide's attribute draws
        fn = lambda sp, pa, yy, ii, at: cdm_sf_particle(sp, pa, yy, ii, at)[:2]      # noqa: E731
    else:
        fn = {"hodina": hodina_particle, "ccdm": ccdm_particle, "vaeccdm": vaeccdm_particle}.get(spec.get("family"), irt_particle)
    S = len(idx_list)
    loss, grads = 0.0, None
    for idx, eps in zip(idx_list, eps_list):
        l, g = fn(spec, params, y_u8, idx, eps)
        loss += l / S
        if grads is None:
            grads = {k: v / S for k, v in g.items()}
        else:
            for k, v in g.items():
                grads[k] = grads[k] + v / S
    return loss, grads


class Adam(object):
    """torch.optim.Adam restated, one state per tensor; lr via dict or callable(module, name)
    (pyro.optim.Adam call sites vi.py:514,627; test.py:321-327,345-350); optional MultiStepLR
    (test.py:352-359) advanced once per iteration (vi.py:639-640)."""

    def __init__(self, lr, betas=(0.9, 0.999), eps=1e-8, milestones=(), gamma=0.1):
        self.lr, self.betas, self.eps = lr, betas, eps
        self.milestones, self.gamma = tuple(milestones), gamma
        self.state = {}
        self.epoch = 0

    def _lr(self, name):
        if callable(self.lr):
            module = name.split("$$$")[0]
            stripped = name.split("$$$")[1] if "$$$" in name else name
            lr = self.lr(module, stripped)["lr"]
        else:
            lr = self.lr
        k = sum(1 for m in self.milestones if m <= self.epoch)
        return lr * self.gamma ** k

    def step(self, params, grads):
        b1, b2 = self.betas
        for name, g in grads.items():
            p = params[name]
            st = self.state.setdefault(name, {"t": 0, "m": np.zeros_like(p), "v": np.zeros_like(p)})
            st["t"] += 1
            t = st["t"]
            st["m"] = b1 * st["m"] + (1 - b1) * g
            st["v"] = b2 * st["v"] + (1 - b2) * g * g
            denom = np.sqrt(st["v"]) / math.sqrt(1 - b2 ** t) + self.eps
            params[name] = (p - (self._lr(name) / (1 - b1 ** t)) * st["m"] / denom).astype(p.dtype)

    def scheduler_step(self):
        self.epoch += 1


# ------------------------------------------------------------------------------------------------
# initial parameters exactly as the reference creates them (vi.py:577-587, 702-721, 901-904, 928-929)
# ------------------------------------------------------------------------------------------------
def default_a_free(D, J):
    if D <= 1:
        return None
    f = np.ones((D, J), dtype=bool)
    for i in range(D):
        f[i, J - i:] = False                                       # vi.py:570-572
    return f


def init_irt_params(spec, J, dtype=np.float32, encoder=None, b0=None, a0=None):
    D, N, model = spec["D"], spec["N"], spec["model"]
    p = {"b": np.zeros((1, J), dtype) if b0 is None else np.asarray(b0, dtype).reshape(1, J)}
    if model != "irt_1pl":
        if a0 is None:
            a0 = np.ones((D, J), dtype)
            if D > 1:
                a0 = a0 * default_a_free(D, J)
        p["a"] = np.asarray(a0, dtype)
    if model in ("irt_3pl", "irt_4pl"):
        p["c"] = np.full((1, J), logit(np.asarray(0.1, dtype)), dtype)   # torch inverts in float32
    if model == "irt_4pl":
        p["d"] = np.full((1, J), logit(np.asarray(1.0, dtype) - np.asarray(0.1, dtype)), dtype)
    if spec["amortized"]:
        for k in ENC_KEYS:
            p["encoder$$$" + k] = np.asarray(encoder[k], dtype)
    else:
        p["x_local"] = np.zeros((N, D), dtype)
        if D == 1:
            p["x_scale"] = np.zeros((N, 1), dtype)
        elif spec["share_cov"]:
            p["x_scale"] = np.zeros((D, D), dtype)
        else:
            p["x_scale"] = np.zeros((N, D, D), dtype)
    return p


def init_hodina_params(spec, J, dtype=np.float32, encoder=None):
    K, N = spec["K"], spec["N"]
    p = {"lam0": np.zeros((1, K), dtype), "lam1": np.zeros((1, K), dtype),
         "g": np.full((1, J), logit(np.asarray(0.1, dtype)), dtype),
         "s": np.full((1, J), logit(np.asarray(0.1, dtype)), dtype)}
    if spec["amortized"]:
        for k in ENC_KEYS:
            p["encoder$$$" + k] = np.asarray(encoder[k], dtype)
    else:
        p["theta_local"] = np.zeros((N, 1), dtype)
        p["theta_scale"] = np.zeros((N, 1), dtype)
    return p


def init_ccdm_params(spec, J, dtype=np.float32, encoder=None):
    p = {"g": np.full((1, J), logit(np.asarray(0.1, dtype)), dtype),
         "s": np.full((1, J), logit(np.asarray(0.1, dtype)), dtype)}
    for k, v in (encoder or {}).items():
        p["encoder$$$" + k] = np.asarray(v, dtype)
    return p


def constrained(name, value):
    """Constrained value as pyro.param(name) returns it."""
    base = name.split("$$$")[-1]
    if name in ("c", "d", "g", "s"):
        return sigmoid(value)
    if name == "attr_p":
        return np.clip(sigmoid(value), np.finfo(np.float32).tiny, 1.0 - EPS32)
    if name in ("lam1", "theta_scale") or (name == "x_scale" and value.ndim == 2 and value.shape[1] == 1):
        return np.exp(value)
    if name == "x_scale":
        D = value.shape[-1]
        diag = np.exp(np.einsum("...ii->...i", value))
        return np.tril(value, -1) + diag[..., None] * np.eye(D, dtype=value.dtype)
    del base
    return value


def rmse_metric(est, true, D=1):
    """The reference's error metric (test.py:70-91; vi.py:645-655) -- a mean ABSOLUTE error."""
    out = {"b": float(np.abs(est["b"] - true["b"]).mean())}
    if "a" in est and "a" in true:
        J = est["a"].shape[1]
        out["a"] = float(np.abs(est["a"] - true["a"]).sum() / (D * J - D * (D - 1) / 2))
    for k in ("c", "d"):
        if k in est and k in true:
            out[k] = float(np.abs(est[k] - true[k]).mean())
    return out
