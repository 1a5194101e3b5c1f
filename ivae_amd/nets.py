"""Drop-in model classes for the reference's ``models/nets.py`` hot path.

Same constructors, ``forward`` / ``elbo`` / ``anneal`` signatures and ``state_dict`` keys as
``MLP`` (nets.py:65-121), ``cleanIVAE`` (124-185), ``cleanVAE`` (188-243), ``Normal``
(293-314), ``iVAE`` (391-486) and ``VAE`` (548-599), so ``runners/ivae.py`` and
``models/wrappers.py`` can import them unchanged.  The arithmetic of ``forward`` and ``elbo``
(+ its backward) runs in libivae_amd.so's HIP kernels through ``ivae_amd.ops``; the modules
only own parameters (ordinary ``nn.Parameter`` views into one flat fp32 arena) and scalars.
There is no CPU path: calling ``forward``/``elbo`` off a ROCm device raises.
"""
from numbers import Number

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

from . import ops
from .ops import ModelShape


def weights_init(m):
    """nets.py:10-12."""
    if isinstance(m, nn.Linear):
        nn.init.xavier_uniform_(m.weight.data)


class Normal:
    """Marker for the only distribution the fused path implements (nets.py:293-314).
    ``sample``/``log_pdf`` are kept for callers that use them directly (plain torch ops)."""
    name = 'gauss'

    def __init__(self, device='cpu'):
        self.device = device
        self.c = 2 * np.pi * torch.ones(1).to(self.device)

    def sample(self, mu, v):
        eps = torch.randn_like(mu)
        return eps.mul(v.sqrt()).add(mu)

    def log_pdf(self, x, mu, v, reduce=True, param_shape=None):
        if param_shape is not None:
            mu, v = mu.view(param_shape), v.view(param_shape)
        lpdf = -0.5 * (torch.log(self.c) + v.log() + (x - mu).pow(2).div(v))
        return lpdf.sum(dim=-1) if reduce else lpdf


class MLP(nn.Module):
    """Container with the reference's constructor and ``fc`` ModuleList (nets.py:66-107).
    Inside the fused models only its parameters are used; its own ``forward`` (plain torch
    ops) exists for callers that evaluate a sub-network by itself."""

    def __init__(self, input_dim, output_dim, hidden_dim, n_layers, activation='none', slope=.1, device='cpu'):
        super().__init__()
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.n_layers = n_layers
        self.device = device
        self.slope = slope
        if isinstance(hidden_dim, Number):
            self.hidden_dim = [hidden_dim] * (self.n_layers - 1)
        elif isinstance(hidden_dim, list):
            self.hidden_dim = hidden_dim
        else:
            raise ValueError('Wrong argument type for hidden_dim: {}'.format(hidden_dim))
        if isinstance(activation, str):
            self.activation = [activation] * (self.n_layers - 1)
        elif isinstance(activation, list):
            self.activation = activation
        else:
            raise ValueError('Wrong argument type for activation: {}'.format(activation))
        for act in self.activation:
            if act not in ops.ACT_IDS:
                raise ValueError('Incorrect activation: {}'.format(act))
        dims = [self.input_dim] + list(self.hidden_dim[:self.n_layers - 1]) + [self.output_dim]
        self.fc = nn.ModuleList([nn.Linear(dims[i], dims[i + 1]) for i in range(self.n_layers)])
        self.to(self.device)

    @staticmethod
    def xtanh(x, alpha=.1):
        return x.tanh() + alpha * x

    def _act(self, kind, h):
        if kind == 'lrelu':
            return F.leaky_relu(h, negative_slope=self.slope)
        if kind == 'xtanh':
            return self.xtanh(h, alpha=self.slope)
        if kind == 'sigmoid':
            return torch.sigmoid(h)
        return h

    def forward(self, x):
        h = x
        for c in range(self.n_layers):
            h = self.fc[c](h)
            if c != self.n_layers - 1:
                h = self._act(self.activation[c], h)
        return h


class _ElboFn(torch.autograd.Function):
    """One fused launch sequence computes loss, z and every parameter gradient; backward only
    scales the stashed flat gradient by the incoming scalar (runners/ivae.py:82 passes 1,
    wrappers.py:69 passes -1 through ``elbo.mul(-1)``)."""

    @staticmethod
    def forward(ctx, model, coef, x, u, eps, *params):
        loss, z, grads = ops.elbo_fwd_bwd(model._shape, coef, model._flat_params(), x, u, eps, model._ws)
        ctx.flat_grads = grads
        ctx.sizes = [p.numel() for p in params]
        ctx.shapes = [p.shape for p in params]
        ctx.mark_non_differentiable(z)
        return loss[0], z

    @staticmethod
    def backward(ctx, gl, gz):
        flat = ctx.flat_grads * gl
        outs, off = [], 0
        for n, s in zip(ctx.sizes, ctx.shapes):
            outs.append(flat[off:off + n].view(s))
            off += n
        return (None, None, None, None, None) + tuple(outs)


class _FusedModel(nn.Module):
    """Shared machinery: flat parameter arena, workspace, kernel dispatch."""

    def _finish_init(self, data_dim, latent_dim, aux_dim, hidden_dim, n_layers, activation, slope):
        if not isinstance(hidden_dim, Number) or not isinstance(activation, str):
            raise NotImplementedError('the fused path supports one hidden width and one activation for all layers')
        self._shape = ModelShape(int(data_dim), int(latent_dim), int(aux_dim), int(hidden_dim), int(n_layers),
                                 activation, float(slope))
        self._ws = ops.Workspace()
        self._arena = None
        self._pack()

    # -- parameter arena -------------------------------------------------------------
    def _pack(self):
        ps = list(self.parameters())
        total = sum(p.numel() for p in ps)
        arena = torch.empty(total, dtype=ps[0].dtype, device=ps[0].device)
        off = 0
        with torch.no_grad():
            for p in ps:
                n = p.numel()
                arena[off:off + n].copy_(p.data.reshape(-1))
                p.data = arena[off:off + n].view(p.shape)
                off += n
        self._arena = arena

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        if getattr(self, '_shape', None) is not None:
            self._pack()
        return out

    def _flat_params(self):
        """The arena, re-packed first if some parameter no longer aliases it."""
        off = 0
        base = self._arena.data_ptr()
        for p in self.parameters():
            if p.data_ptr() != base + 4 * off or p.device != self._arena.device:
                self._pack()
                break
            off += p.numel()
        if self._arena.dtype != torch.float32:
            raise TypeError('ivae_amd kernels are fp32 only')
        return self._arena

    # -- kernels ---------------------------------------------------------------------
    def _eps(self, x, eps):
        """The draw of nets.py:147 / nets.py:302 (torch.randn), on the inputs' device."""
        if eps is None:
            eps = torch.randn(x.shape[0], self._shape.latent_dim, dtype=torch.float32, device=x.device)
        return eps

    FORWARD_ROWS = 65536     # rows per ivae_forward call: a full-dataset forward (runners/ivae.py:118-122) runs in blocks

    def _forward_tensors(self, x, u, eps=None):
        sh = self._shape
        eps = self._eps(x, eps)
        B = x.shape[0]
        params = self._flat_params()
        mk = lambda w: torch.empty(B, w, dtype=torch.float32, device=x.device)
        f, g, v, l, z = mk(sh.data_dim), mk(sh.latent_dim), mk(sh.latent_dim), mk(sh.latent_dim), mk(sh.latent_dim)
        # every row's forward is independent of the others, so blocks give the same values as one call; the workspace
        # (hidden activations of every layer) stays bounded for N = 1e6-row datasets
        for lo in range(0, B, self.FORWARD_ROWS):
            hi = min(B, lo + self.FORWARD_ROWS)
            n = hi - lo
            coef = ops.make_coef(decoder_var=self._decoder_var_f, global_batch=n)
            z[lo:hi] = ops.forward(sh, coef, params, x[lo:hi], None if u is None else u[lo:hi], eps[lo:hi], self._ws)
            g[lo:hi] = self._ws.view('g', n, sh.latent_dim)
            v[lo:hi] = self._ws.view('v', n, sh.latent_dim)
            l[lo:hi] = self._ws.view('l', n, sh.latent_dim)
            f[lo:hi] = self._ws.view('f', n, sh.data_dim)
        return f, g, v, z, l

    def direct_step(self, optimizer, x, u, a, b, c, d, N, pairwise, eps=None, z_out=None):
        """zero_grad -> elbo -> backward -> optimizer.step() (runners/ivae.py:75-91) without the autograd and
        per-parameter optimizer machinery: ivae_elbo_fwd_bwd writes the flat gradient, `optimizer.step_flat`
        (ivae_amd.FusedAdam) consumes it.  Same kernels, same noise draw, same update as the drop-in sequence.
        Returns (loss, z) like `elbo` (loss detached)."""
        eps = self._eps(x, eps)
        coef = ops.make_coef(a, b, c, d, N, self._decoder_var_f, pairwise, x.shape[0])
        flat = self._flat_params()
        loss, z, grads = ops.elbo_fwd_bwd(self._shape, coef, flat, x, u, eps, self._ws, z_out=z_out)
        optimizer.step_flat(flat, grads)
        return loss[0], z

    def direct_steps(self, optimizer, xs, us, a, b, c, d, N, pairwise, eps=None, z_out=None, coefs=None):
        """`direct_step` over the batches xs (n, B, data_dim) [us (n, B, aux_dim)] with ONE library call
        (`ivae_elbo_adam_steps`): same kernels, same update, constant learning rate over the call; the noise of all n steps
        is one torch.randn draw.  Loss weights: (a, b, c, d, N, pairwise) for every step, or `coefs` = one such tuple per
        step (the per-iteration schedule of `iVAE.anneal`).  Returns (losses (n,), z (n, B, latent_dim))."""
        if eps is None:
            eps = torch.randn(xs.shape[0], xs.shape[1], self._shape.latent_dim, dtype=torch.float32, device=xs.device)
        if coefs is None:
            coef = ops.make_coef(a, b, c, d, N, self._decoder_var_f, pairwise, xs.shape[1])
        else:
            coef = [ops.make_coef(a_, b_, c_, d_, N_, self._decoder_var_f, pw_, xs.shape[1]) for a_, b_, c_, d_, N_, pw_ in coefs]
        flat = self._flat_params()
        m, v, taken, group = optimizer.flat_state(flat)
        b1, b2 = group['betas']
        loss, z = ops.elbo_adam_steps(self._shape, coef, flat, m, v, xs, us, eps, self._ws, taken, group['lr'], b1, b2,
                                      group['eps'], z_out=z_out)
        optimizer.advance(xs.shape[0])
        return loss[:, 0], z

    def _loss(self, x, u, a, b, c, d, N, pairwise, eps=None):
        if torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in (x, u)):
            # the reference's autograd would hand back dLoss/dx; the kernels form parameter gradients only -- say so
            # instead of leaving x.grad silently empty
            raise NotImplementedError('the fused path does not differentiate with respect to its inputs (x, u): detach them')
        eps = self._eps(x, eps)
        coef = ops.make_coef(a, b, c, d, N, self._decoder_var_f, pairwise, x.shape[0])
        return _ElboFn.apply(self, coef, x, u, eps, *self.parameters())


class cleanIVAE(_FusedModel):
    """nets.py:124-185."""

    def __init__(self, data_dim, latent_dim, aux_dim, n_layers=3, activation='xtanh', hidden_dim=50, slope=.1):
        super().__init__()
        self.data_dim = data_dim
        self.latent_dim = latent_dim
        self.aux_dim = aux_dim
        self.hidden_dim = hidden_dim
        self.n_layers = n_layers
        self.activation = activation
        self.slope = slope
        self.prior_mean = torch.zeros(1)
        self.logl = MLP(aux_dim, latent_dim, hidden_dim, n_layers, activation=activation, slope=slope)
        self.f = MLP(latent_dim, data_dim, hidden_dim, n_layers, activation=activation, slope=slope)
        self.decoder_var = .1 * torch.ones(1)
        self._decoder_var_f = float(self.decoder_var)
        self.g = MLP(data_dim + aux_dim, latent_dim, hidden_dim, n_layers, activation=activation, slope=slope)
        self.logv = MLP(data_dim + aux_dim, latent_dim, hidden_dim, n_layers, activation=activation, slope=slope)
        self._finish_init(data_dim, latent_dim, aux_dim, hidden_dim, n_layers, activation, slope)

    def forward(self, x, u, eps=None):
        """-> (f, g, v, s, l), nets.py:165-170."""
        f, g, v, z, l = self._forward_tensors(x, u, eps)
        return f, g, v, z, l

    def elbo(self, x, u, N, a=1., b=1., c=1., d=1., eps=None):
        """-> (loss, z): the NEGATIVE elbo, nets.py:172-185."""
        return self._loss(x, u, a, b, c, d, N, True, eps)


class cleanVAE(_FusedModel):
    """nets.py:188-243 (beta-VAE / beta-TCVAE baselines via a,b,c,d)."""

    def __init__(self, data_dim, latent_dim, n_layers=3, activation='xtanh', hidden_dim=50, slope=.1):
        super().__init__()
        self.data_dim = data_dim
        self.latent_dim = latent_dim
        self.hidden_dim = hidden_dim
        self.n_layers = n_layers
        self.activation = activation
        self.slope = slope
        self.prior_mean = torch.zeros(1)
        self.prior_var = torch.ones(1)
        self.f = MLP(latent_dim, data_dim, hidden_dim, n_layers, activation=activation, slope=slope)
        self.decoder_var = .1 * torch.ones(1)
        self._decoder_var_f = float(self.decoder_var)
        self.g = MLP(data_dim, latent_dim, hidden_dim, n_layers, activation=activation, slope=slope)
        self.logv = MLP(data_dim, latent_dim, hidden_dim, n_layers, activation=activation, slope=slope)
        self._finish_init(data_dim, latent_dim, 0, hidden_dim, n_layers, activation, slope)

    def forward(self, x, eps=None):
        """-> (f, g, v, s), nets.py:224-228."""
        f, g, v, z, _ = self._forward_tensors(x, None, eps)
        return f, g, v, z

    def elbo(self, x, u, N, a=1., b=1., c=1., d=1., eps=None):
        """-> (loss, z), nets.py:230-243 (``u`` is accepted and ignored, as there)."""
        return self._loss(x, None, a, b, c, d, N, True, eps)


class iVAE(_FusedModel):
    """nets.py:391-486."""

    def __init__(self, latent_dim, data_dim, aux_dim, prior=None, decoder=None, encoder=None,
                 n_layers=3, hidden_dim=50, activation='lrelu', slope=.1, device='cpu', anneal=False):
        super().__init__()
        self.data_dim = data_dim
        self.latent_dim = latent_dim
        self.aux_dim = aux_dim
        self.hidden_dim = hidden_dim
        self.n_layers = n_layers
        self.activation = activation
        self.slope = slope
        self.anneal_params = anneal
        for dist in (prior, decoder, encoder):
            if dist is not None and getattr(dist, 'name', None) != 'gauss':
                raise NotImplementedError('the fused iVAE path implements Gaussian prior/encoder/decoder only')
        self.prior_dist = prior if prior is not None else Normal(device=device)
        self.decoder_dist = decoder if decoder is not None else Normal(device=device)
        self.encoder_dist = encoder if encoder is not None else Normal(device=device)
        self.prior_mean = torch.zeros(1).to(device)
        self.logl = MLP(aux_dim, latent_dim, hidden_dim, n_layers, activation=activation, slope=slope, device=device)
        self.f = MLP(latent_dim, data_dim, hidden_dim, n_layers, activation=activation, slope=slope, device=device)
        self.decoder_var = .01 * torch.ones(1).to(device)
        self._decoder_var_f = float(.01 * torch.ones(1))
        self.g = MLP(data_dim + aux_dim, latent_dim, hidden_dim, n_layers, activation=activation, slope=slope,
                     device=device)
        self.logv = MLP(data_dim + aux_dim, latent_dim, hidden_dim, n_layers, activation=activation, slope=slope,
                        device=device)
        self.apply(weights_init)
        self._training_hyperparams = [1., 1., 1., 1., 1]
        self._finish_init(data_dim, latent_dim, aux_dim, hidden_dim, n_layers, activation, slope)

    def forward(self, x, u, eps=None):
        """-> (f, var), (g, v), z, (prior_mean, l); nets.py:450-455."""
        f, g, v, z, l = self._forward_tensors(x, u, eps)
        return (f, self.decoder_var), (g, v), z, (self.prior_mean, l)

    def elbo(self, x, u, eps=None):
        """-> (elbo, z): the POSITIVE elbo, nets.py:457-475."""
        if self.anneal_params:
            a, b, c, d, N = self._training_hyperparams
            loss, z = self._loss(x, u, a, b, c, d, N, True, eps)
        else:
            loss, z = self._loss(x, u, 1., 1., 1., 1., 1., False, eps)
        return loss.neg(), z

    def anneal(self, N, max_iter, it):
        """nets.py:477-486."""
        thr = int(max_iter / 1.6)
        a = 0.5 / self._decoder_var_f
        self._training_hyperparams[-1] = N
        self._training_hyperparams[0] = min(2 * a, a + a * it / thr)
        self._training_hyperparams[1] = max(1, a * .3 * (1 - it / thr))
        self._training_hyperparams[2] = min(1, it / thr)
        self._training_hyperparams[3] = max(1, a * .5 * (1 - it / thr))
        if it > thr:
            self.anneal_params = False


class VAE(_FusedModel):
    """nets.py:548-599."""

    def __init__(self, latent_dim, data_dim, decoder=None, encoder=None,
                 n_layers=3, hidden_dim=50, activation='lrelu', slope=.1, device='cpu'):
        super().__init__()
        self.data_dim = data_dim
        self.latent_dim = latent_dim
        self.hidden_dim = hidden_dim
        self.n_layers = n_layers
        self.activation = activation
        self.slope = slope
        for dist in (decoder, encoder):
            if dist is not None and getattr(dist, 'name', None) != 'gauss':
                raise NotImplementedError('the fused VAE path implements Gaussian encoder/decoder only')
        self.decoder_dist = decoder if decoder is not None else Normal(device=device)
        self.encoder_dist = encoder if encoder is not None else Normal(device=device)
        self.prior_dist = Normal(device)
        self.prior_mean = torch.zeros(1).to(device)
        self.prior_var = torch.ones(1).to(device)
        self.f = MLP(latent_dim, data_dim, hidden_dim, n_layers, activation=activation, slope=slope, device=device)
        self.decoder_var = .01 * torch.ones(1).to(device)
        self._decoder_var_f = float(.01 * torch.ones(1))
        self.g = MLP(data_dim, latent_dim, hidden_dim, n_layers, activation=activation, slope=slope, device=device)
        self.logv = MLP(data_dim, latent_dim, hidden_dim, n_layers, activation=activation, slope=slope, device=device)
        self._finish_init(data_dim, latent_dim, 0, hidden_dim, n_layers, activation, slope)

    def forward(self, x, eps=None):
        """-> (f, var), (g, v), z, (prior_mean, prior_var); nets.py:587-591."""
        f, g, v, z, _ = self._forward_tensors(x, None, eps)
        return (f, self.decoder_var), (g, v), z, (self.prior_mean, self.prior_var)

    def elbo(self, x, eps=None):
        """-> (elbo, z), nets.py:593-599."""
        loss, z = self._loss(x, None, 1., 1., 1., 1., 1., False, eps)
        return loss.neg(), z
