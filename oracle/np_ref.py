"""numpy restatement of the iVAE ELBO hot path (TEST INFRASTRUCTURE ONLY).

Every function cites the reference lines (relative to /root/reference) whose
arithmetic it follows.  The backward pass is written analytically (SURVEY.md
§8 a-bwd), so it is independent of any autograd engine; ``torch_ref`` holds
the autograd twin.  ``dtype`` selects fp32 (mimics the reference) or fp64
(ground truth for tolerance budgeting).

Pinned by tests/test_oracle_golden.py against fixtures generated from the
reference itself (tools/make_golden.py).
"""
from dataclasses import dataclass

import numpy as np

MLP_ORDER_IVAE = ("logl", "f", "g", "logv")   # registration order, models/nets.py:422-430, 137-143
MLP_ORDER_VAE = ("f", "g", "logv")            # models/nets.py:573-576, 203-207


@dataclass
class Spec:
    data_dim: int
    latent_dim: int
    aux_dim: int            # 0 -> VAE family (no prior MLP, prior variance 1)
    hidden_dim: int = 50
    n_layers: int = 3
    activation: str = "lrelu"
    slope: float = 0.1
    decoder_var: float = 0.01

    @property
    def mlps(self):
        return MLP_ORDER_IVAE if self.aux_dim > 0 else MLP_ORDER_VAE

    def mlp_io(self, name):
        """(in_dim, out_dim) of each MLP; models/nets.py:422-430."""
        if name == "logl":
            return self.aux_dim, self.latent_dim
        if name == "f":
            return self.latent_dim, self.data_dim
        return self.data_dim + self.aux_dim, self.latent_dim

    def layer_dims(self, name):
        """[(in,out)] per nn.Linear; models/nets.py:99-105."""
        i, o = self.mlp_io(name)
        if self.n_layers == 1:
            return [(i, o)]
        dims = [(i, self.hidden_dim)]
        dims += [(self.hidden_dim, self.hidden_dim)] * (self.n_layers - 2)
        dims += [(self.hidden_dim, o)]
        return dims

    def param_names(self):
        out = []
        for m in self.mlps:
            for li in range(self.n_layers):
                out += [f"{m}.fc.{li}.weight", f"{m}.fc.{li}.bias"]
        return out

    def param_shapes(self):
        """Shapes in param_names() order: weight (out, in) then bias (out,) per layer."""
        out = []
        for m in self.mlps:
            for i, o in self.layer_dims(m):
                out += [(o, i), (o,)]
        return out

    def unflatten(self, flat):
        """One model's flat arena (parameters() order) -> {name: array}."""
        P, off = {}, 0
        for n, shp in zip(self.param_names(), self.param_shapes()):
            k = 1
            for v in shp:
                k *= v
            P[n] = flat[off:off + k].reshape(shp)
            off += k
        assert off == flat.size
        return P

    def n_params(self):
        return sum((i + 1) * o for m in self.mlps for i, o in self.layer_dims(m))


@dataclass
class Coef:
    """loss = -mean(a*lpx - b*(lqc-lq) - c*(lq-lqi) - d*(lqi-lp)); nets.py:184,242,471."""
    a: float = 1.0
    b: float = 1.0
    c: float = 1.0
    d: float = 1.0
    N: float = 1.0
    pairwise: bool = False


# ----------------------------------------------------------------- activations
def act_fwd(p, kind, slope):
    """models/nets.py:88-95,109-112."""
    if kind == "lrelu":
        return np.where(p > 0, p, p * p.dtype.type(slope))
    if kind == "xtanh":
        return np.tanh(p) + p.dtype.type(slope) * p
    if kind == "sigmoid":
        return 1 / (1 + np.exp(-p))
    if kind == "none":
        return p
    raise ValueError(kind)


def act_bwd(p, kind, slope):
    one = p.dtype.type(1)
    if kind == "lrelu":
        return np.where(p > 0, one, p.dtype.type(slope))
    if kind == "xtanh":
        t = np.tanh(p)
        return one - t * t + p.dtype.type(slope)
    if kind == "sigmoid":
        s = 1 / (1 + np.exp(-p))
        return s * (one - s)
    if kind == "none":
        return np.ones_like(p)
    raise ValueError(kind)


# ------------------------------------------------------------------------- MLP
def mlp_forward(P, name, h, spec):
    """models/nets.py:114-121.  Returns (out, cache) where cache keeps layer
    inputs and pre-activations for the analytic backward."""
    ins, pres = [], []
    for li in range(spec.n_layers):
        W, b = P[f"{name}.fc.{li}.weight"], P[f"{name}.fc.{li}.bias"]
        ins.append(h)
        p = h @ W.T + b
        pres.append(p)
        h = p if li == spec.n_layers - 1 else act_fwd(p, spec.activation, spec.slope)
    return h, (ins, pres)


def mlp_backward(P, name, cache, dout, spec, need_dx):
    """Standard backprop of mlp_forward; returns (grads dict, dx or None)."""
    ins, pres = cache
    G = {}
    delta = dout
    for li in reversed(range(spec.n_layers)):
        W = P[f"{name}.fc.{li}.weight"]
        if li != spec.n_layers - 1:
            delta = delta * act_bwd(pres[li], spec.activation, spec.slope)
        G[f"{name}.fc.{li}.weight"] = delta.T @ ins[li]
        G[f"{name}.fc.{li}.bias"] = delta.sum(0)
        if li > 0 or need_dx:
            delta = delta @ W
    return G, (delta if need_dx else None)


# ------------------------------------------------------------- Gaussian terms
def log_normal(x, mu, v, log2pi):
    """models/nets.py:45-52 and 306-314 (element-wise, no reduction)."""
    half = x.dtype.type(0.5)
    return -half * (log2pi + np.log(v) + (x - mu) ** 2 / v)


def logsumexp(t, axis):
    m = t.max(axis=axis, keepdims=True)
    return (m + np.log(np.exp(t - m).sum(axis=axis, keepdims=True))).squeeze(axis)


# --------------------------------------------------------------------- forward
def elbo_forward(P, x, u, eps, spec, coef, dtype=np.float32, global_batch=None, pair_cols=None):
    """Forward of models/nets.py:450-475 (iVAE), 587-599 (VAE), 165-185
    (cleanIVAE), 224-243 (cleanVAE).  Returns a dict with every intermediate;
    ``loss`` is the quantity to MINIMISE (family A returns it directly, family
    B returns its negation).

    Row-shard form (the reference has no multi-GPU path): with ``global_batch`` = Bg the
    rows given are a shard of a batch of Bg and ``loss`` = -sum(row)/Bg, so shard losses add up to the
    full-batch loss; ``pair_cols`` = (g_all, v_all) are the encoder outputs of the WHOLE batch, the
    columns j of T[i,j,k] (nets.py:180,466)."""
    dt = np.dtype(dtype).type
    P = {k: np.asarray(v, dtype) for k, v in P.items()}
    x = np.asarray(x, dtype)
    eps = np.asarray(eps, dtype)
    B = x.shape[0]
    log2pi = dt(np.log(2 * np.pi))
    out = {}
    caches = {}
    if spec.aux_dim > 0:
        u = np.asarray(u, dtype)
        xu = np.concatenate([x, u], 1)                       # nets.py:437,152
        logl, caches["logl"] = mlp_forward(P, "logl", u, spec)
        l = np.exp(logl)                                     # nets.py:448,163
    else:
        xu = x
        logl = np.zeros((B, spec.latent_dim), dtype)
        l = np.ones((B, spec.latent_dim), dtype)             # nets.py:571,201
    g, caches["g"] = mlp_forward(P, "g", xu, spec)
    logv, caches["logv"] = mlp_forward(P, "logv", xu, spec)
    v = np.exp(logv)                                         # nets.py:440,155
    z = eps * np.sqrt(v) + g                                 # nets.py:301-304,145-149
    f, caches["f"] = mlp_forward(P, "f", z, spec)
    dv = dt(spec.decoder_var)
    lpx = log_normal(x, f, np.full_like(f, dv), log2pi).sum(-1)     # nets.py:459,175
    lqc = log_normal(z, g, v, log2pi).sum(-1)                       # nets.py:460,176
    lp = log_normal(z, np.zeros_like(z), l, log2pi).sum(-1)         # nets.py:461,177,235
    out.update(g=g, v=v, logv=logv, l=l, logl=logl, z=z, f=f, lpx=lpx, lqc=lqc, lp=lp,
               eps=eps, xu=xu, caches=caches)
    a, b, c, d = (dt(t) for t in (coef.a, coef.b, coef.c, coef.d))
    Bg = B if global_batch is None else int(global_batch)
    if coef.pairwise:
        gc, vc = (g, v) if pair_cols is None else (np.asarray(pair_cols[0], dtype), np.asarray(pair_cols[1], dtype))
        T = log_normal(z[:, None, :], gc[None, :, :], vc[None, :, :], log2pi)   # nets.py:180,466
        logMN = dt(np.log(Bg * coef.N))
        S = T.sum(-1)
        lq = logsumexp(S, 1) - logMN                                          # nets.py:181,468
        lqi = (logsumexp(T, 1) - logMN).sum(-1)                               # nets.py:182,469
        row = a * lpx - b * (lqc - lq) - c * (lq - lqi) - d * (lqi - lp)      # nets.py:184,471
        out.update(T=T, lq=lq, lqi=lqi)
    else:
        row = a * lpx - b * lqc + d * lp                                      # nets.py:475,599
    out["row"] = row
    out["loss"] = -row.mean(dtype=dtype) if global_batch is None else -row.sum(dtype=dtype) / dt(Bg)
    return out


# -------------------------------------------------------------------- backward
def pairwise_terms(z, g_cols, v_cols, coef, global_batch, dtype=np.float64):
    """Row statistics and gradient terms of T[i,j,k] = log N(z_ik; g_jk, v_jk) (nets.py:180-182, 466-469) for
    rows z of one shard against the columns of the whole batch: what ``ivae_pairwise`` returns."""
    dt = np.dtype(dtype).type
    z, gc, vc = (np.asarray(t, dtype) for t in (z, g_cols, v_cols))
    T = log_normal(z[:, None, :], gc[None], vc[None], dt(np.log(2 * np.pi)))
    S = T.sum(-1)
    lse_s, lse_k = logsumexp(S, 1), logsumexp(T, 1)
    logMN = dt(np.log(global_batch * coef.N))
    Wt = (coef.b - coef.c) * np.exp(S - lse_s[:, None])[:, :, None] + (coef.c - coef.d) * np.exp(T - lse_k[:, None, :])
    diff = z[:, None, :] - gc[None]
    rij = diff / vc[None]
    return dict(lq=lse_s - logMN, lqi=(lse_k - logMN).sum(-1), lse_s=lse_s, lse_k=lse_k, rz=(Wt * -rij).sum(1),
                cg=(Wt * rij).sum(0), cv=(Wt * -0.5 * (1 - diff * rij)).sum(0))


def elbo_backward(P, x, u, fw, spec, coef, dtype=np.float32, global_batch=None, pair=None):
    """Analytic gradient of ``loss`` w.r.t. every parameter (what
    ``loss.backward()`` produces at runners/ivae.py:82, wrappers.py:69).
    Row-shard form: ``global_batch`` as in elbo_forward; ``pair`` = dict(rz, cg, cv) holds the pairwise
    gradient terms of THESE rows with cg/cv already summed over every shard's rows."""
    dt = np.dtype(dtype).type
    P = {k: np.asarray(v, dtype) for k, v in P.items()}
    x = np.asarray(x, dtype)
    B = x.shape[0]
    w = dt(1.0) / dt(B if global_batch is None else global_batch)
    a, b, c, d = (dt(t) for t in (coef.a, coef.b, coef.c, coef.d))
    g, v, l, z, f, eps = fw["g"], fw["v"], fw["l"], fw["z"], fw["f"], fw["eps"]
    caches = fw["caches"]
    half = dt(0.5)
    G = {}
    # decoder: dL/df = -a*w*(x-f)/var
    df = -a * w * (x - f) / dt(spec.decoder_var)
    Gf, dz = mlp_backward(P, "f", caches["f"], df, spec, need_dx=True)
    G.update(Gf)
    r = (z - g) / v                      # (z-g)/v  == eps/sqrt(v)
    # z as a node: contributions of lqc and lp
    Gz = dz + (-w) * (b * r - d * z / l)
    dg = (-w) * (-b * r)
    dlogv = (-w) * (b * half * (1 - (z - g) ** 2 / v))
    if coef.pairwise and pair is not None:
        Gz = Gz + (-w) * np.asarray(pair["rz"], dtype)
        dg = dg + (-w) * np.asarray(pair["cg"], dtype)
        dlogv = dlogv + (-w) * np.asarray(pair["cv"], dtype)
    elif coef.pairwise:
        T = fw["T"]
        S = T.sum(-1)
        Pm = np.exp(S - logsumexp(S, 1)[:, None])                  # softmax_j sum_k T
        Q = np.exp(T - logsumexp(T, 1)[:, None, :])                # softmax_j per k
        Wt = (b - c) * Pm[:, :, None] + (c - d) * Q                # (i,j,k) weights of T
        diff = z[:, None, :] - g[None, :, :]
        rij = diff / v[None, :, :]
        Gz = Gz + (-w) * (Wt * (-rij)).sum(1)
        dg = dg + (-w) * (Wt * rij).sum(0)
        dlogv = dlogv + (-w) * (Wt * (-half) * (1 - diff * rij)).sum(0)
    dg = dg + Gz
    dlogv = dlogv + Gz * half * eps * np.sqrt(v)
    Gg, _ = mlp_backward(P, "g", caches["g"], dg, spec, need_dx=False)
    Gv, _ = mlp_backward(P, "logv", caches["logv"], dlogv, spec, need_dx=False)
    G.update(Gg)
    G.update(Gv)
    if spec.aux_dim > 0:
        dlogl = (-w) * d * (-half) * (1 - z * z / l)
        Gl, _ = mlp_backward(P, "logl", caches["logl"], dlogl, spec, need_dx=False)
        G.update(Gl)
    return G


# ------------------------------------------------------------------------ Adam
def adam_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam single-tensor arithmetic (torch 2.10 adam.py:457-547) as
    called at runners/ivae.py:38,91 and wrappers.py:44,70.  ``step`` is the
    1-based step count AFTER increment.  Returns new (p, m, v)."""
    dt = p.dtype.type
    m = m + (g - m) * dt(1 - beta1)                   # lerp_
    v = v * dt(beta2) + dt(1 - beta2) * g * g         # mul_().addcmul_()
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    step_size = lr / bc1
    denom = np.sqrt(v) / dt(np.sqrt(bc2)) + dt(eps)
    p = p - dt(step_size) * (m / denom)
    return p, m, v


# ------------------------------------------------------------------------- MCC
def corr_block(x, y):
    """Cross block ``np.corrcoef(x, y, rowvar=False)[:d, d:]`` of metrics/mcc.py:328
    in float64, written from the definition (two-pass centred moments)."""
    x = np.asarray(x, np.float64)
    y = np.asarray(y, np.float64)
    xc = x - x.mean(0)
    yc = y - y.mean(0)
    cxy = xc.T @ yc
    sx = np.sqrt((xc * xc).sum(0))
    sy = np.sqrt((yc * yc).sum(0))
    return np.clip(cxy / sx[:, None] / sy[None, :], -1, 1)


def mcc_from_corr(cc):
    """metrics/mcc.py:333-335: abs, Hungarian assignment, mean of matches."""
    from scipy.optimize import linear_sum_assignment
    cc = np.abs(cc)
    return cc[linear_sum_assignment(-1 * cc)].mean()


def mcc(x, y):
    return mcc_from_corr(corr_block(x, y))
