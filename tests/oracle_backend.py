"""Test double for `ivae_amd.ops` backed by the CPU oracle (oracle/np_ref.py, float64 arithmetic, float32
tensors at the interface).  Used ONLY by tests/test_dist_cpu.py to run the data-parallel exchange logic of
ivae_amd/dist.py on CPU ranks (gloo); it is never importable from the product package."""
from types import SimpleNamespace

import numpy as np
import torch

from oracle import np_ref


def make_coef(a=1., b=1., c=1., d=1., N=1., decoder_var=.01, pairwise=False, global_batch=1):
    return SimpleNamespace(a=a, b=b, c=c, d=d, dataset_size=N, decoder_var=decoder_var, pairwise=bool(pairwise),
                           global_batch=int(global_batch))


def _spec(shape, coef):
    return np_ref.Spec(shape.data_dim, shape.latent_dim, shape.aux_dim, shape.hidden_dim, shape.n_layers,
                       shape.activation, shape.slope, coef.decoder_var)


def _ocoef(coef):
    return np_ref.Coef(coef.a, coef.b, coef.c, coef.d, coef.dataset_size, coef.pairwise)


def _unflat(spec, flat):
    P, off = {}, 0
    flat = flat.detach().numpy().astype(np.float64)
    for name in spec.param_names():
        m, _, li, kind = name.split(".")
        i, o = spec.layer_dims(m)[int(li)]
        shp = (o, i) if kind == "weight" else (o,)
        n = int(np.prod(shp))
        P[name] = flat[off:off + n].reshape(shp)
        off += n
    return P


def _flat(spec, G):
    return torch.from_numpy(np.concatenate([G[n].reshape(-1) for n in spec.param_names()]).astype(np.float32))


class Workspace:
    def __init__(self):
        self.t = {}

    def view(self, name, batch, width=None, model=0):
        return self.t[name]


def forward(shape, coef, params, x, u, eps, ws, n_models=1):
    spec = _spec(shape, coef)
    plain = np_ref.Coef(coef.a, coef.b, coef.c, coef.d, coef.dataset_size, False)
    fw = np_ref.elbo_forward(_unflat(spec, params), x.numpy(), None if u is None else u.numpy(), eps.numpy(), spec,
                             plain, np.float64, global_batch=coef.global_batch)
    ws.fw = fw
    B, d = x.shape[0], shape.latent_dim
    for k in ("g", "v"):
        ws.t[k] = torch.from_numpy(fw[k].astype(np.float32))
    for k, shp in (("lq", (B,)), ("lqi", (B,)), ("lse_s", (B,)), ("lse_k", (B, d)), ("rz", (B, d)), ("cg", (B, d)),
                   ("cv", (B, d))):
        ws.t[k] = torch.zeros(shp)
    return torch.from_numpy(fw["z"].astype(np.float32))


def pairwise(coef, z, g_cols, v_cols, out=None):
    res = np_ref.pairwise_terms(z.numpy(), g_cols.numpy(), v_cols.numpy(), _ocoef(coef), coef.global_batch)
    res = {k: torch.from_numpy(np.ascontiguousarray(v).astype(np.float32)) for k, v in res.items()}
    for k, t in (out or {}).items():
        t.copy_(res[k])
        res[k] = t
    return res


def _into(gl_out, loss, grads):
    if gl_out is None:
        return loss, grads
    gl_out[:grads.numel()].copy_(grads)
    gl_out[grads.numel():].copy_(loss)
    return gl_out[grads.numel():], gl_out[:grads.numel()]


def backward(shape, coef, params, x, u, eps, ws, n_models=1, gl_out=None):
    spec = _spec(shape, coef)
    fw, oc = ws.fw, _ocoef(coef)
    row = coef.a * fw["lpx"] - coef.b * fw["lqc"] + coef.d * fw["lp"]
    pair = None
    if coef.pairwise:
        lq, lqi = ws.t["lq"].numpy().astype(np.float64), ws.t["lqi"].numpy().astype(np.float64)
        row = (coef.a * fw["lpx"] - coef.b * (fw["lqc"] - lq) - coef.c * (lq - lqi) - coef.d * (lqi - fw["lp"]))
        pair = {k: ws.t[k].numpy().astype(np.float64) for k in ("rz", "cg", "cv")}
    G = np_ref.elbo_backward(_unflat(spec, params), x.numpy(), None if u is None else u.numpy(), fw, spec, oc, np.float64,
                             global_batch=coef.global_batch, pair=pair)
    loss = torch.tensor([-row.sum() / coef.global_batch], dtype=torch.float32)
    return _into(gl_out, loss, _flat(spec, G))


def elbo_fwd_bwd(shape, coef, params, x, u, eps, ws, n_models=1, gl_out=None):
    z = forward(shape, coef, params, x, u, eps, ws)
    if coef.pairwise:
        if coef.global_batch != x.shape[0]:
            raise RuntimeError("pairwise shards need the phase entry points")
        pw = pairwise(coef, z, ws.t["g"], ws.t["v"])
        ws.t.update(pw)
    loss, grads = backward(shape, coef, params, x, u, eps, ws, gl_out=gl_out)
    return loss, z, grads


def adam_step(params, exp_avg, exp_avg_sq, grads, lr, step, beta1=0.9, beta2=0.999, eps=1e-8):
    p, m, v = np_ref.adam_step(params.numpy(), grads.numpy(), exp_avg.numpy(), exp_avg_sq.numpy(), step, lr, beta1,
                               beta2, eps)
    params.copy_(torch.from_numpy(p))
    exp_avg.copy_(torch.from_numpy(m))
    exp_avg_sq.copy_(torch.from_numpy(v))
