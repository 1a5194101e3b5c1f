"""Packed-model training loop: many independent iVAE runs (seeds / sweeps) advanced in lock-step
on one GPU, each doing the reference's per-step sequence

    zero_grad -> elbo -> backward -> Adam.step        (runners/ivae.py:69-91, wrappers.py:60-70)

on its own shuffled dataset, without the per-step host syncs of the reference loops (`loss.item()`
runners/ivae.py:84, host MCC :86): losses stay on the device until asked for.
Data lives in HBM for the whole run; batches are formed on the device (DataLoader(shuffle=True,
drop_last=True) semantics, runners/ivae.py:25).
"""
import math

import torch

from . import ops


def init_packed_params(shape, n_models, seed, device, xavier=True):
    """Per-model parameters with the reference's initial distributions: weights xavier-uniform
    (nets.py:10-12,432) or nn.Linear's default U(+-1/sqrt(fan_in)) when xavier=False
    (cleanIVAE/cleanVAE/VAE), biases U(+-1/sqrt(fan_in)).  Returns (n_models, P)."""
    g = torch.Generator(device=device).manual_seed(seed)
    parts = []
    H, L = shape.hidden_dim, shape.n_layers
    for m in shape.mlps:
        i, o = shape.mlp_io(m)
        dims = [i] + [H] * (L - 1) + [o]
        for k in range(L):
            fi, fo = dims[k], dims[k + 1]
            wb = math.sqrt(6.0 / (fi + fo)) if xavier else 1.0 / math.sqrt(fi)
            parts.append((torch.rand(n_models, fo * fi, generator=g, device=device) * 2 - 1) * wb)
            parts.append((torch.rand(n_models, fo, generator=g, device=device) * 2 - 1) / math.sqrt(fi))
    return torch.cat(parts, 1).contiguous()


class PlateauLR:
    """`ReduceLROnPlateau(optimizer, factor, patience)` (mode 'min', relative threshold 1e-4, no cooldown, min_lr 0, eps
    1e-8 -- the defaults the reference's loops run with, runners/ivae.py:39, wrappers.py:45) for every run of a pack at
    once: `step(losses)` takes the runs' mean epoch losses and returns their learning rates.  Same decisions as torch's
    scheduler run per model (tests/test_api_cpu.py)."""

    def __init__(self, n_models, lr, factor=0.1, patience=0, threshold=1e-4, eps=1e-8):
        self.lr = [float(lr)] * n_models
        self.factor, self.patience, self.threshold, self.eps = factor, patience, threshold, eps
        self.best = [math.inf] * n_models
        self.bad = [0] * n_models

    def step(self, losses):
        for i, cur in enumerate(float(v) for v in losses):
            if cur < self.best[i] * (1.0 - self.threshold):          # torch: is_better, mode 'min', threshold_mode 'rel'
                self.best[i] = cur
                self.bad[i] = 0
            else:
                self.bad[i] += 1
            if self.bad[i] > self.patience:
                new = self.lr[i] * self.factor
                if self.lr[i] - new > self.eps:
                    self.lr[i] = new
                self.bad[i] = 0
        return list(self.lr)


class PackedTrainer:
    """Generic-kernel implementation (any shape): one ivae_forward / ivae_backward / ivae_adam_step
    launch sequence per step covers all models."""
    impl = "generic kernels (k_fwd_enc, k_dec, k_bwd_enc, k_reduce, k_adam)"
    rng = "torch.randn on device"
    dominant_kernel = "k_fwd_enc+k_dec+k_bwd_enc+k_reduce"

    def __init__(self, shape, n_models, batch, lr=1e-2, seed=0, device="cuda", coef=None, xavier=True,
                 dataset_size=None):
        self.shape, self.M, self.B, self.lr = shape, n_models, batch, lr
        self.dev = torch.device(device)
        self.params = init_packed_params(shape, n_models, seed, self.dev, xavier)
        self.exp_avg = torch.zeros_like(self.params)
        self.exp_avg_sq = torch.zeros_like(self.params)
        self.t = 0
        self.gen = torch.Generator(device=self.dev).manual_seed(seed + 12345)
        self.coef_kw = coef or dict(a=1., b=1., c=1., d=1., pairwise=False)
        self.decoder_var = .01
        self.ws = ops.Workspace()
        self.kernel_ms = []
        self._ev = []
        self._loss = None
        self._batches = None
        self._cursor = 0
        self.N = dataset_size

    def set_data(self, x, u):
        self.x, self.u = x, u
        self.N = self.N or x.shape[1]
        self._batches = None

    def _reshuffle(self):
        M, N, B = self.M, self.x.shape[1], self.B
        nb = N // B                                   # drop_last=True
        perm = torch.rand(M, N, generator=self.gen, device=self.dev).argsort(1)[:, :nb * B]
        gx = torch.gather(self.x, 1, perm.unsqueeze(-1).expand(-1, -1, self.x.shape[2]))
        self._bx = gx.view(M, nb, B, -1).transpose(0, 1).contiguous()          # (nb, M, B, dd)
        if self.shape.aux_dim > 0:
            gu = torch.gather(self.u, 1, perm.unsqueeze(-1).expand(-1, -1, self.u.shape[2]))
            self._bu = gu.view(M, nb, B, -1).transpose(0, 1).contiguous()
        self._batches, self._cursor = nb, 0

    def step(self, timed=False):
        if self._batches is None or self._cursor >= self._batches:
            self._reshuffle()
        x = self._bx[self._cursor]
        u = self._bu[self._cursor] if self.shape.aux_dim > 0 else None
        self._cursor += 1
        eps = torch.randn(self.M, self.B, self.shape.latent_dim, generator=self.gen, device=self.dev)
        coef = ops.make_coef(N=self.N, decoder_var=self.decoder_var, global_batch=self.B, **self.coef_kw)
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        loss, z, grads = ops.elbo_fwd_bwd(self.shape, coef, self.params, x, u, eps, self.ws, self.M)
        if timed:
            e1.record()
            self._ev.append((e0, e1))
        self.t += 1
        ops.adam_step(self.params.view(-1), self.exp_avg.view(-1), self.exp_avg_sq.view(-1), grads.view(-1),
                      self.lr, self.t)
        self._loss, self._z = loss, z
        return loss

    def run(self, n_steps):
        """n_steps steps with one `ivae_elbo_adam_steps` call per stretch of an epoch (same kernels and update as
        `step`; the noise of a stretch is one draw).  `self.lr`: one learning rate or a list of one per model."""
        done, out = 0, []
        while done < n_steps:
            if self._batches is None or self._cursor >= self._batches:
                self._reshuffle()
            k = min(n_steps - done, self._batches - self._cursor)
            xs = self._bx[self._cursor:self._cursor + k]
            us = self._bu[self._cursor:self._cursor + k] if self.shape.aux_dim > 0 else None
            eps = torch.randn(k, self.M, self.B, self.shape.latent_dim, generator=self.gen, device=self.dev)
            coef = ops.make_coef(N=self.N, decoder_var=self.decoder_var, global_batch=self.B, **self.coef_kw)
            loss, z = ops.elbo_adam_steps(self.shape, coef, self.params.view(-1), self.exp_avg.view(-1),
                                          self.exp_avg_sq.view(-1), xs, us, eps, self.ws, self.t, self.lr, n_models=self.M)
            self._cursor += k
            self.t += k
            done += k
            self._loss, self._z = loss[-1], z[-1]
            out.append(loss.t())
        self.last_losses = torch.cat(out, 1) if len(out) > 1 else out[0]          # (n_models, n_steps)
        return self._loss

    def losses(self):
        return self._loss

    def mean_kernel_seconds(self):
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in self._ev]
        return 1e-3 * sum(ms) / max(len(ms), 1)

    def dominant_flops_per_sample(self, total):
        """The timed launch group is the whole ELBO forward + backward: all of SURVEY §8(d)'s per-sample FLOPs."""
        return total


class FusedPackedTrainer:
    """Fused-kernel implementation: `ivae_train_steps` advances every packed model by up to 128
    steps per launch (parameters in LDS, Adam moments in registers, in-kernel Philox noise).
    Host work per launch: none beyond the call (a launch points into the epoch's device-side permutation table)."""
    impl = "fused persistent kernel (k_train_steps)"
    rng = "in-kernel Philox4x32-10 + Box-Muller"
    dominant_kernel = "k_train_steps"
    CHUNK = 125          # steps per launch (divides the 625-step epoch of N=40000, B=64)

    def __init__(self, shape, n_models, batch, lr=1e-2, seed=0, device="cuda", xavier=True, decoder_var=.01,
                 a=1., b=1., d=1.):
        if not ops.train_steps_supported(shape, batch):
            raise RuntimeError("shape not covered by the fused kernel; use PackedTrainer")
        self.shape, self.M, self.B, self.lr, self.seed = shape, n_models, batch, lr, seed
        self.dev = torch.device(device)
        self.params = init_packed_params(shape, n_models, seed, self.dev, xavier)
        self.exp_avg = torch.zeros_like(self.params)
        self.exp_avg_sq = torch.zeros_like(self.params)
        self.t = 0
        self.gen = torch.Generator(device=self.dev).manual_seed(seed + 12345)
        self.coef = ops.make_coef(a=a, b=b, d=d, decoder_var=decoder_var, global_batch=batch)
        self._ev, self._launch_steps = [], []
        self._perm, self._cursor, self._nb = None, 0, 0
        self._loss = None

    def set_data(self, x, u):
        self.x = x.contiguous()
        self.u = u.contiguous() if u is not None else None
        self._perm = None

    # -- input pipeline: ONE device-side permutation table per epoch; a launch points into it (batch_idx_stride of
    #    ivae_train_steps), so no per-launch index slice exists.  (Cutting slices or drawing the next epoch's permutation on
    #    a side stream WHILE a launch runs was measured and dropped: the kernel owns every CU, and whatever shares them
    #    delays some workgroups' start -- 125-step launches 4.12 -> 4.85 ms.)
    def _reshuffle(self):
        N = self.x.shape[1]
        self._nb = N // self.B                                      # drop_last=True (runners/ivae.py:25)
        self._perm = torch.rand(self.M, N, generator=self.gen, device=self.dev).argsort(1).to(torch.int32)
        self._cursor = 0

    def prepare_timing(self, n_launches):
        """Create the HIP events of the next `n_launches` timed launches now (an event is created by the driver at its
        first record: ~10 us each that would otherwise sit between the launches being timed)."""
        pool = [torch.cuda.Event(enable_timing=True) for _ in range(2 * n_launches)]
        for e in pool:
            e.record()
        torch.cuda.synchronize(self.dev)
        self._ev_pool = pool

    def prepare_indices(self, n_steps):
        """Draw the epoch's permutation now if the next launch would have to (kept for callers that time launches)."""
        if self._perm is None or self._cursor >= self._nb:
            self._reshuffle()

    def _event(self):
        pool = getattr(self, "_ev_pool", None)
        if pool:
            return pool.pop()
        e = torch.cuda.Event(enable_timing=True)
        e.record()                       # (an event's handle exists from its first record on)
        return e

    def run(self, n_steps, timed=False, want_z=False):
        """Advance all models by n_steps; returns the (n_models, n_steps) loss tensor (device)."""
        out, z = [], None
        while n_steps > 0:
            if self._perm is None or self._cursor >= self._nb:
                self._reshuffle()
            k = min(n_steps, self.CHUNK, self._nb - self._cursor)
            ev = None
            if timed:       # the events are recorded inside the library, directly around the launch: recorded from here,
                # around the call, they would also hold the host's ~20 us between the record and the launch
                ev = (self._event(), self._event())
            loss, z = ops.train_steps(self.shape, self.coef, self.params, self.exp_avg, self.exp_avg_sq, self.x, self.u,
                                      self._perm, self.t, self.lr, seed=self.seed, want_z=want_z and k == n_steps,
                                      table_steps=(self._cursor, k), events=ev)
            if timed:
                self._ev.append(ev)
                self._launch_steps.append(k)
            out.append(loss)
            self.t += k
            self._cursor += k
            n_steps -= k
        self._loss = out[-1][:, -1]
        self._z = z
        return torch.cat(out, 1) if len(out) > 1 else out[0]

    def step(self, timed=False):
        return self.run(1, timed)

    def losses(self):
        return self._loss

    def mean_kernel_seconds(self):
        """Average duration of one k_train_steps launch and the steps it covered."""
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in self._ev]
        return 1e-3 * sum(ms) / max(len(ms), 1), sum(self._launch_steps) / max(len(self._launch_steps), 1)
