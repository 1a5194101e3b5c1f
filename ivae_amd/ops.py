"""Tensor-level entry points over the C ABI.  PyTorch supplies device memory and the
stream; every computation below happens in libivae_amd.so's HIP kernels."""
import contextlib
import ctypes as C
import math
from dataclasses import dataclass

import torch

from . import _lib
from ._lib import ACT_IDS, IvaeCoef, IvaeDims, IvaeWsLayout, check


# Tuning of the shape-generic kernels (IvaeDims.tile_rows / waves / wgrad_rows); 0 = the library picks.  Host-side default
# for every IvaeDims this module forms; the library itself keeps no state.
TUNING = {"tile_rows": 0, "waves": 0, "wgrad_rows": 0}


@dataclass(frozen=True)
class ModelShape:
    """Static shape of a model (constructor arguments of models/nets.py:125,190,392,550)."""
    data_dim: int
    latent_dim: int
    aux_dim: int            # 0: VAE family
    hidden_dim: int
    n_layers: int
    activation: str
    slope: float

    def dims(self, batch, n_models=1):
        if self.activation not in ACT_IDS:
            raise ValueError("Incorrect activation: {}".format(self.activation))
        return IvaeDims(batch, self.data_dim, self.latent_dim, self.aux_dim, self.hidden_dim, self.n_layers,
                        ACT_IDS[self.activation], float(self.slope), n_models, int(TUNING["tile_rows"]),
                        int(TUNING["waves"]), int(TUNING["wgrad_rows"]))

    @property
    def mlps(self):
        return ("logl", "f", "g", "logv") if self.aux_dim > 0 else ("f", "g", "logv")

    def mlp_io(self, name):
        if name == "logl":
            return self.aux_dim, self.latent_dim
        if name == "f":
            return self.latent_dim, self.data_dim
        return self.data_dim + self.aux_dim, self.latent_dim

    def n_params(self):
        n = C.c_int64()
        check(_lib.lib().ivae_param_count(C.byref(self.dims(1)), C.byref(n)), "ivae_param_count")
        return n.value


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _f32c(t, name):
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on a ROCm device: ivae_amd has no CPU path")
    return t if t.is_contiguous() else t.contiguous()


def make_coef(a=1., b=1., c=1., d=1., N=1., decoder_var=.01, pairwise=False, global_batch=1):
    return IvaeCoef(float(a), float(b), float(c), float(d), float(N), float(decoder_var), int(bool(pairwise)),
                    int(global_batch))


def workspace_bytes(dims, pairwise=True):
    n = C.c_size_t()
    check(_lib.lib().ivae_workspace_bytes(C.byref(dims), int(bool(pairwise)), C.byref(n)), "ivae_workspace_bytes")
    return n.value


def workspace_layout(dims, pairwise=True):
    lay = IvaeWsLayout()
    check(_lib.lib().ivae_workspace_layout(C.byref(dims), int(bool(pairwise)), C.byref(lay)), "ivae_workspace_layout")
    return lay


class Workspace:
    """Caller-owned scratch for one (shape, batch, n_models); grows on demand."""

    def __init__(self):
        self.buf = None
        self.key = None
        self.layout = None

    def get(self, dims, device, pairwise):
        """Scratch for calls with this (dims, coef.pairwise): the layout depends on both."""
        pairwise = bool(pairwise)
        key = tuple(getattr(dims, f) for f, _ in dims._fields_) + (str(device), pairwise)
        if key != self.key:
            nbytes = workspace_bytes(dims, pairwise)
            if self.buf is None or self.buf.numel() * 4 < nbytes or self.buf.device != device:
                self.buf = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
            self.layout = workspace_layout(dims, pairwise)
            self.key = key
        return self.buf

    def view(self, name, batch, width=None, model=0):
        """Row tensor `name` of the last call as a (batch[, width]) view."""
        off = getattr(self.layout, name) + model * self.layout.per_model
        n = batch * (width or 1)
        t = self.buf[off:off + n]
        return t.view(batch, width) if width else t


def _grad_loss_buffers(params, n_models, gl_out):
    """(grads, loss) as views of one flat [grads | loss] buffer (data-parallel callers all-reduce it as ONE message),
    or freshly allocated."""
    dev = params.device
    if gl_out is None:
        return torch.empty_like(params), torch.empty(n_models, dtype=torch.float32, device=dev)
    n = params.numel()
    if gl_out.dtype != torch.float32 or gl_out.device != dev or not gl_out.is_contiguous() or gl_out.numel() != n + n_models:
        raise ValueError("gl_out must be a contiguous float32 buffer of params.numel() + n_models elements")
    return gl_out[:n].view(params.shape), gl_out[n:]


def elbo_fwd_bwd(shape, coef, params, x, u, eps, ws, n_models=1, z_out=None, gl_out=None):
    """ivae_elbo_fwd_bwd: -> (loss (n_models,), z, grads flat like params).  z_out: write z there (a contiguous fp32
    tensor of z's shape, e.g. one row block of an epoch's buffer) instead of allocating.  gl_out: a flat buffer that
    receives [grads | loss]; the returned tensors are views of it."""
    x = _f32c(x, "x")
    eps = _f32c(eps, "eps")
    params = _f32c(params, "params")
    if shape.aux_dim > 0:
        u = _f32c(u, "u")
    else:
        u = None
    B = x.shape[-2]
    dims = shape.dims(B, n_models)
    dev = x.device
    with torch.cuda.device(dev):
        buf = ws.get(dims, dev, coef.pairwise)
        if z_out is None:
            z = torch.empty(x.shape[:-1] + (shape.latent_dim,), dtype=torch.float32, device=dev)
        else:
            z = z_out
            if (z.dtype != torch.float32 or not z.is_contiguous() or z.device != dev
                    or tuple(z.shape) != tuple(x.shape[:-1]) + (shape.latent_dim,)):
                raise ValueError("z_out must be a contiguous float32 tensor of z's shape on the inputs' device")
        grads, loss = _grad_loss_buffers(params, n_models, gl_out)
        check(_lib.lib().ivae_elbo_fwd_bwd(C.byref(dims), C.byref(coef), _ptr(params), _ptr(x), _ptr(u), _ptr(eps),
                                           _ptr(z), _ptr(loss), _ptr(grads), _ptr(buf), buf.numel() * 4,
                                           _stream(dev)), "ivae_elbo_fwd_bwd")
    return loss, z, grads


def elbo_adam_steps(shape, coef, params, exp_avg, exp_avg_sq, x_steps, u_steps, eps_steps, ws, step0, lr, beta1=0.9,
                    beta2=0.999, adam_eps=1e-8, n_models=1, z_out=None):
    """ivae_elbo_adam_steps: x_steps (n_steps, [n_models,] batch, data_dim) etc.; params and the Adam moments are
    updated in place.  `lr`: one learning rate, or a sequence of n_models of them (packed runs with their own
    schedulers).  `coef`: one IvaeCoef, or a sequence of n_steps of them (annealed loss weights: one per step).
    -> (loss (n_steps, n_models), z (like eps_steps))."""
    coef_steps = None
    if isinstance(coef, (list, tuple)):
        if len(coef) != x_steps.shape[0]:
            raise ValueError("coef must be one IvaeCoef or one per step")
        coef_steps = (IvaeCoef * len(coef))(*coef)
        # the workspace serves every step: sized for the pairwise terms if any step has them
        coef = next((c for c in coef if c.pairwise), coef[0])
    if isinstance(lr, (list, tuple)):
        if len(lr) != n_models:
            raise ValueError("lr must be one value or one per model")
        model_lr, lr = (C.c_double * n_models)(*[float(v) for v in lr]), float(lr[0])
    else:
        model_lr = None
    x_steps = _f32c(x_steps, "x_steps")
    eps_steps = _f32c(eps_steps, "eps_steps")
    u_steps = _f32c(u_steps, "u_steps") if shape.aux_dim > 0 else None
    for t, n in ((params, "params"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise RuntimeError(f"{n} must be a contiguous float32 ROCm tensor")
    n_steps, B = x_steps.shape[0], x_steps.shape[-2]
    if eps_steps.shape[0] != n_steps or (u_steps is not None and u_steps.shape[0] != n_steps):
        raise ValueError("x_steps, u_steps and eps_steps hold different numbers of steps")
    if params.numel() != n_models * shape.n_params() or exp_avg.numel() != params.numel() or exp_avg_sq.numel() != params.numel():
        raise ValueError("params / exp_avg / exp_avg_sq do not hold n_models * n_params floats")
    dims = shape.dims(B, n_models)
    dev = x_steps.device
    with torch.cuda.device(dev):
        buf = ws.get(dims, dev, coef.pairwise)
        if z_out is None:
            z = torch.empty_like(eps_steps)
        else:
            z = z_out
            if (z.dtype != torch.float32 or not z.is_contiguous() or z.device != dev or z.numel() != eps_steps.numel()):
                raise ValueError("z_out must be a contiguous float32 tensor of eps_steps' size on the inputs' device")
        loss = torch.empty(n_steps, n_models, dtype=torch.float32, device=dev)
        grads = torch.empty_like(params)
        check(_lib.lib().ivae_elbo_adam_steps(C.byref(dims), C.byref(coef), coef_steps, _ptr(params), _ptr(exp_avg), _ptr(exp_avg_sq),
                                              _ptr(x_steps), _ptr(u_steps), _ptr(eps_steps), int(n_steps), int(step0),
                                              float(lr), model_lr, float(beta1), float(beta2), float(adam_eps), _ptr(loss), _ptr(z),
                                              _ptr(grads), _ptr(buf), buf.numel() * 4, _stream(dev)), "ivae_elbo_adam_steps")
    return loss, z


def forward(shape, coef, params, x, u, eps, ws, n_models=1):
    """ivae_forward: fills the workspace row tensors, returns z."""
    x = _f32c(x, "x")
    eps = _f32c(eps, "eps")
    params = _f32c(params, "params")
    u = _f32c(u, "u") if shape.aux_dim > 0 else None
    B = x.shape[-2]
    dims = shape.dims(B, n_models)
    dev = x.device
    with torch.cuda.device(dev):
        buf = ws.get(dims, dev, coef.pairwise)
        z = torch.empty(x.shape[:-1] + (shape.latent_dim,), dtype=torch.float32, device=dev)
        check(_lib.lib().ivae_forward(C.byref(dims), C.byref(coef), _ptr(params), _ptr(x), _ptr(u), _ptr(eps), _ptr(z),
                                      _ptr(buf), buf.numel() * 4, _stream(dev)), "ivae_forward")
    return z


def pairwise(coef, z, g_cols, v_cols, out=None):
    """ivae_pairwise on explicit row/column tensors -> dict of outputs (single model).  `out`: tensors to write into
    instead of allocating, by name (any subset of lq lqi lse_s lse_k rz cg cv; e.g. views of the ELBO workspace, so
    that ivae_backward finds them in place)."""
    z, g_cols, v_cols = _f32c(z, "z"), _f32c(g_cols, "g"), _f32c(v_cols, "v")
    n_rows, d = z.shape
    n_cols = g_cols.shape[0]
    dev = z.device
    mk = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    shapes = dict(lq=(n_rows,), lqi=(n_rows,), lse_s=(n_rows,), lse_k=(n_rows, d), rz=(n_rows, d), cg=(n_cols, d),
                  cv=(n_cols, d))
    given = out or {}
    out = {}
    for k, shp in shapes.items():
        t = given.get(k)
        if t is None:
            t = mk(*shp)
        elif t.dtype != torch.float32 or t.device != dev or not t.is_contiguous() or tuple(t.shape) != shp:
            raise ValueError("out['{}'] must be a contiguous float32 tensor of shape {} on the inputs' device".format(k, shp))
        out[k] = t
    nb = C.c_size_t()
    check(_lib.lib().ivae_pairwise_workspace_bytes(n_rows, n_cols, d, C.byref(nb)), "ivae_pairwise_workspace_bytes")
    with torch.cuda.device(dev):
        scratch = mk(max(1, (nb.value + 3) // 4))
        check(_lib.lib().ivae_pairwise(n_rows, n_cols, d, C.byref(coef), _ptr(z), _ptr(g_cols), _ptr(v_cols),
                                       _ptr(out["lq"]), _ptr(out["lqi"]), _ptr(out["lse_s"]), _ptr(out["lse_k"]),
                                       _ptr(out["rz"]), _ptr(out["cg"]), _ptr(out["cv"]), _ptr(scratch),
                                       scratch.numel() * 4, _stream(dev)), "ivae_pairwise")
    return out


def backward(shape, coef, params, x, u, eps, ws, n_models=1, gl_out=None):
    """ivae_backward after forward (+pairwise results in the workspace).  gl_out: see elbo_fwd_bwd."""
    x = _f32c(x, "x")
    eps = _f32c(eps, "eps")
    u = _f32c(u, "u") if shape.aux_dim > 0 else None
    B = x.shape[-2]
    dims = shape.dims(B, n_models)
    dev = x.device
    with torch.cuda.device(dev):
        buf = ws.get(dims, dev, coef.pairwise)
        grads, loss = _grad_loss_buffers(params, n_models, gl_out)
        check(_lib.lib().ivae_backward(C.byref(dims), C.byref(coef), _ptr(params), _ptr(x), _ptr(u), _ptr(eps),
                                       _ptr(loss), _ptr(grads), _ptr(buf), buf.numel() * 4, _stream(dev)),
              "ivae_backward")
    return loss, grads


def adam_step(params, exp_avg, exp_avg_sq, grads, lr, step, beta1=0.9, beta2=0.999, eps=1e-8):
    """ivae_adam_step in place over flat contiguous fp32 tensors."""
    for t, n in ((params, "params"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq"), (grads, "grads")):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise RuntimeError(f"{n} must be a contiguous float32 ROCm tensor")
    dev = params.device
    with torch.cuda.device(dev):
        check(_lib.lib().ivae_adam_step(_ptr(params), _ptr(exp_avg), _ptr(exp_avg_sq), _ptr(grads), params.numel(),
                                        float(lr), float(beta1), float(beta2), float(eps), int(step), _stream(dev)),
              "ivae_adam_step")


def corr(x, y):
    """ivae_corr: float64 (d,d) cross-correlation block of fp32 (n,d) tensors, on device."""
    x, y = _f32c(x, "x"), _f32c(y, "y")
    if x.shape != y.shape or x.dim() != 2:
        raise ValueError("x and y have different shapes")
    n, d = x.shape
    dev = x.device
    nb = C.c_size_t()
    check(_lib.lib().ivae_corr_workspace_bytes(n, d, C.byref(nb)), "ivae_corr_workspace_bytes")
    with torch.cuda.device(dev):
        ws = torch.empty((nb.value + 7) // 8, dtype=torch.float64, device=dev)
        cc = torch.empty(d, d, dtype=torch.float64, device=dev)
        check(_lib.lib().ivae_corr(_ptr(x), _ptr(y), n, d, _ptr(cc), _ptr(ws), ws.numel() * 8, _stream(dev)),
              "ivae_corr")
    return cc


def corr_batched(x, y):
    """ivae_corr_batched: (n_blocks, n, d) fp32 pairs -> (n_blocks, d, d) float64 correlation blocks, one launch."""
    x, y = _f32c(x, "x"), _f32c(y, "y")
    if x.shape != y.shape or x.dim() != 3:
        raise ValueError("x and y must be (n_blocks, n, d) tensors of one shape")
    nb, n, d = x.shape
    dev = x.device
    with torch.cuda.device(dev):
        cc = torch.empty(nb, d, d, dtype=torch.float64, device=dev)
        check(_lib.lib().ivae_corr_batched(_ptr(x), _ptr(y), nb, n, d, _ptr(cc), _stream(dev)), "ivae_corr_batched")
    return cc


def train_steps_supported(shape, batch):
    return bool(_lib.lib().ivae_train_steps_supported(C.byref(shape.dims(batch))))


def train_steps(shape, coef, params, exp_avg, exp_avg_sq, x_data, u_data, batch_idx, step0, lr, eps=None, seed=0,
                beta1=0.9, beta2=0.999, adam_eps=1e-8, want_z=False, want_z_steps=False, table_steps=None, events=None):
    """ivae_train_steps: n_steps fused (elbo -> backward -> Adam) iterations for all packed models.
    params/exp_avg/exp_avg_sq (n_models, P) are updated in place; batch_idx (n_models, n_steps, B) int32.
    `lr`: one learning rate, or a sequence of n_models of them (packed runs with their own schedulers).
    Returns (loss (n_models, n_steps), z of the last step or None); with want_z_steps the second value is the
    (n_models, n_steps, B, latent) z of every step instead.
    table_steps=(first, n): batch_idx is instead an epoch's whole (n_models, rows) int32 permutation table and the call runs
    the n steps whose rows start at column first * 64 of every model's row (no slice is cut: the library takes the
    table's row stride).
    events=(start, stop): torch.cuda.Event pair recorded inside the library directly before the call's first kernel launch and
    after its last (a measurement hook: bench.py)."""
    for t, n in ((params, "params"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq"), (x_data, "x_data")):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise RuntimeError(f"{n} must be a contiguous float32 ROCm tensor")
    idx_stride, idx_off = 0, 0
    if table_steps is not None:
        if batch_idx.dtype != torch.int32 or not batch_idx.is_contiguous() or batch_idx.dim() != 2:
            raise TypeError("with table_steps, batch_idx must be a contiguous int32 (n_models, rows) table")
        first, n_steps = (int(v) for v in table_steps)
        n_models, B, idx_stride = batch_idx.shape[0], 64, batch_idx.shape[1]
        if first < 0 or n_steps <= 0 or (first + n_steps) * B > idx_stride:
            raise ValueError("table_steps reaches past the table's rows")
        idx_off = first * B * 4
    else:
        if batch_idx.dtype != torch.int32 or not batch_idx.is_contiguous() or batch_idx.dim() != 3:
            raise TypeError("batch_idx must be a contiguous int32 (n_models, n_steps, batch) tensor")
        n_models, n_steps, B = batch_idx.shape
    if params.shape[0] != n_models or x_data.shape[0] != n_models:
        raise ValueError("params / x_data / batch_idx disagree on n_models")
    model_lr = None
    if isinstance(lr, (list, tuple)):
        if len(lr) != n_models:
            raise ValueError("lr must be one value or one per model")
        ref = max(float(v) for v in lr)
        model_lr, lr = (C.c_double * n_models)(*[float(v) for v in lr]), (ref if ref > 0 else 1.0)
    if shape.aux_dim > 0:
        u_data = _f32c(u_data, "u_data")
    else:
        u_data = None
    if eps is not None:
        eps = _f32c(eps, "eps")
        if tuple(eps.shape) != (n_models, n_steps, B, shape.latent_dim):
            raise ValueError("eps must be (n_models, n_steps, batch, latent_dim)")
    dims = shape.dims(B, n_models)
    dev = params.device
    # (a device guard only when the caller's current device is another one: entering it costs microseconds per launch)
    with (torch.cuda.device(dev) if torch.cuda.current_device() != dev.index else contextlib.nullcontext()):
        loss = torch.empty(n_models, n_steps, dtype=torch.float32, device=dev)
        z = torch.empty(n_models, B, shape.latent_dim, dtype=torch.float32, device=dev) if want_z else None
        zs = (torch.empty(n_models, n_steps, B, shape.latent_dim, dtype=torch.float32, device=dev)
              if want_z_steps else None)
        if events is not None:
            # torch.cuda.Event pair (created with timing and recorded once, so that their handles exist): they are
            # recorded inside the library, directly around the launches (include/ivae_amd.h: ivae_debug_launch_events)
            check(_lib.lib().ivae_debug_launch_events(C.c_void_p(events[0].cuda_event), C.c_void_p(events[1].cuda_event)),
                  "ivae_debug_launch_events")
        check(_lib.lib().ivae_train_steps(C.byref(dims), C.byref(coef), _ptr(params), _ptr(exp_avg), _ptr(exp_avg_sq),
                                          _ptr(x_data), _ptr(u_data), x_data.shape[1],
                                          C.c_void_p(batch_idx.data_ptr() + idx_off), idx_stride, _ptr(eps),
                                          int(seed), n_steps, int(step0), float(lr), model_lr, float(beta1), float(beta2),
                                          float(adam_eps), _ptr(loss), _ptr(z), _ptr(zs), _stream(dev)),
              "ivae_train_steps")
    return loss, (zs if want_z_steps else z)
