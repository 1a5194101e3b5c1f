"""Training loops around the hot path: counterparts of `runners/ivae.py::runner` (family A, what
`main.py --config *.yaml` runs) and `models/wrappers.py::IVAE_wrapper / VAE_wrapper` (family B, library API).

Same inputs, schedules and return values as the reference loops; what differs is where things live:
  * the dataset sits in HBM for the whole run and batches are formed on the device from a device-side
    permutation (DataLoader(shuffle=True[, drop_last=True]) semantics, runners/ivae.py:25, wrappers.py:28);
  * no per-step host sync: the reference calls `loss.item()` and a host MCC on every step
    (runners/ivae.py:84-86, wrappers.py:72-76); here losses accumulate on the device, the per-step
    correlation blocks of the batch MCC are computed by the `ivae_corr` kernel into a device buffer, and the
    Hungarian assignments of a whole epoch run on the host once the epoch's last launch is queued;
  * when the model is one the fused persistent kernel covers (plain-ELBO iVAE/VAE, 3 lrelu layers of width 50,
    batch 64 -- BASELINE configs[1]) the wrappers advance a whole epoch per `ivae_train_steps` launch chain
    with parameters in LDS; any other shape takes the generic kernels through the drop-in modules.
Shuffling and noise come from torch's device generators, so trajectories match the reference statistically
(same distribution of permutations and of eps), not draw for draw; tests/test_gpu_e2e.py replays recorded
reference permutations and eps through the same kernels for the draw-for-draw check.
"""
import itertools
import time

import numpy as np
import torch
from scipy.optimize import linear_sum_assignment
from torch import optim

from . import ops
from .data import ResidentDataset
from .mcc import corr_matrix_gpu, mean_corr_coef as mcc
from .nets import VAE, cleanIVAE, cleanVAE, iVAE
from .optim import FusedAdam
from .runlog import Logger, checkpoint


class _ReduceLROnPlateau(optim.lr_scheduler.ReduceLROnPlateau):
    """The reference passes verbose=True (runners/ivae.py:39, wrappers.py:45), which current torch rejects; keep
    the behaviour (announce reductions) without the argument."""

    def step(self, metrics, *a, **kw):
        before = [g['lr'] for g in self.optimizer.param_groups]
        super().step(metrics, *a, **kw)
        for i, (b, g) in enumerate(zip(before, self.optimizer.param_groups)):
            if g['lr'] != b:
                print('reducing learning rate of group {} to {:.4e}.'.format(i, g['lr']))


def _epoch_batches(n, batch_size, drop_last, device, generator):
    perm = torch.randperm(n, device=device, generator=generator)
    stop = (n // batch_size) * batch_size if drop_last else n
    return [perm[i:i + batch_size] for i in range(0, stop, batch_size)]


_PERMS = {}


def _assignment_scores(cc_abs):
    """Mean matched |correlation| per block (metrics/mcc.py:333-335); 0 where the block is not finite, like the
    bare `except: perf = 0` of runners/ivae.py:85-88.

    The reference solves one assignment problem per training step on the host (scipy); an epoch of K packed seeds is
    625 K of them, which at K = 64 cost more host time than the epoch's kernels cost GPU time.  Up to d = 6 the d!
    assignments are therefore scored for all blocks at once in numpy (the optimum is the optimum however it is found; the
    score is formed from the matched entries exactly as the reference forms it); wider blocks go through scipy one by one."""
    cc_abs = np.asarray(cc_abs)
    n = cc_abs.shape[0]
    d = cc_abs.shape[-1] if cc_abs.ndim == 3 else 0
    if n == 0 or d == 0 or d > 6 or cc_abs.shape[1] != d:
        return [float(cc[linear_sum_assignment(-1 * cc)].mean()) if np.isfinite(cc).all() else 0. for cc in cc_abs]
    if d not in _PERMS:
        perms = np.array(list(itertools.permutations(range(d))), dtype=np.intp)                  # (d!, d)
        onehot = np.zeros((len(perms), d * d))
        onehot[np.arange(len(perms))[:, None], np.arange(d)[None, :] * d + perms] = 1.          # assignment -> its d entries
        _PERMS[d] = perms, np.ascontiguousarray(onehot.T)
    perms, onehot_t = _PERMS[d]
    rows = np.arange(d)
    finite = np.isfinite(cc_abs).all((1, 2))
    safe = np.where(finite[:, None, None], cc_abs, 0.)
    best = (safe.reshape(n, d * d) @ onehot_t).argmax(1)                     # totals of all d! assignments: one matrix product
    matched = safe[np.arange(n)[:, None], rows[None, :], perms[best]]        # (n, d), row order as scipy returns it
    return [float(v) for v in np.where(finite, matched.mean(-1), 0.)]


# ------------------------------------------------------------------------------------------------ family A
class _SeedPack:
    """The `n_sims` models of one `main.py --n-sims K` run as ONE packed model set: their parameter arenas become the
    rows of one (K, P) buffer (each model's `nn.Parameter`s stay views of its row, so `state_dict()` is unchanged),
    their Adam moments the rows of two more, and an epoch of all K runs is one `ivae_elbo_adam_steps` call whose
    launches cover every model (grid dimension = model).  Every run keeps its own FusedAdam -- state_dict layout of
    torch.optim.Adam, the learning rate its own ReduceLROnPlateau sets -- over views of those rows."""

    def __init__(self, models, lr):
        self.models = models
        self.shape = models[0]._shape
        P = models[0]._flat_params().numel()
        dev = models[0]._flat_params().device
        self.params = torch.empty(len(models), P, dtype=torch.float32, device=dev)
        self.m = torch.zeros_like(self.params)
        self.v = torch.zeros_like(self.params)
        self.steps_taken = 0
        self.ws = ops.Workspace()
        self.optimizers = []
        for i, model in enumerate(models):
            with torch.no_grad():
                self.params[i].copy_(model._flat_params())
                off = 0
                for p in model.parameters():          # re-home the model's arena in row i
                    n = p.numel()
                    p.data = self.params[i, off:off + n].view(p.shape)
                    off += n
            model._arena = self.params[i]
            opt = FusedAdam(model.parameters(), lr=lr)
            off = 0
            for p in model.parameters():
                n = p.numel()
                opt.state[p] = {'step': torch.tensor(0.), 'exp_avg': self.m[i, off:off + n].view(p.shape),
                                'exp_avg_sq': self.v[i, off:off + n].view(p.shape)}
                off += n
            self.optimizers.append(opt)

    def epoch(self, xs, us, a, b, c, d, N, eps=None, z_out=None):
        """xs (n_steps, K, B, data_dim) [us (n_steps, K, B, aux_dim)]: zero_grad -> elbo -> backward -> step
        (runners/ivae.py:75-91) for every step and every run.  -> losses (n_steps, K), z (n_steps, K, B, latent_dim)."""
        K = len(self.models)
        if eps is None:
            eps = torch.randn(xs.shape[0], K, xs.shape[2], self.shape.latent_dim, dtype=torch.float32, device=xs.device)
        coef = ops.make_coef(a, b, c, d, N, self.models[0]._decoder_var_f, True, xs.shape[2])
        lrs = [opt.param_groups[0]['lr'] for opt in self.optimizers]
        g0 = self.optimizers[0].param_groups[0]
        loss, z = ops.elbo_adam_steps(self.shape, coef, self.params, self.m, self.v, xs, us, eps, self.ws, self.steps_taken,
                                      lrs, g0['betas'][0], g0['betas'][1], g0['eps'], n_models=K, z_out=z_out)
        self.steps_taken += xs.shape[0]
        for opt in self.optimizers:
            for st in opt.state.values():
                st['step'] = torch.tensor(float(self.steps_taken))
        return loss, z


def runner(args, config, *, replay=None, trace=None, init_seeds=False, init_perturb=0):
    """runners/ivae.py:13-128.  `args`: data_path, seed, n_sims.  `config`: the yaml keys of configs/*.yaml plus
    `device`.  Returns (perfs, loss_hists, perf_hists).

    The reference trains its `n_sims` seeds one after the other; here they are independent models of one packed set and
    advance together, an epoch of ALL seeds per library call (a 256-CU GPU is idle with one batch-64 model).  Each seed
    still has its own shuffles, noise, scheduler and learning rate, and the returned lists are in seed order.

    Test hooks (tests/test_gpu_runner_trace.py replays runs recorded from the reference's runner draw for draw):
    `replay` supplies the random draws instead of the device generators -- `replay.order(seed, epoch)` the epoch's row
    order (N // B * B indices), `replay.eps(seed, epoch)` the (N // B, B, d_latent) noise of its steps,
    `replay.eval_eps(seed)` the (N, d_latent) noise of the final full-data sample; `trace` (a dict) receives the
    per-step losses, and per epoch the loss weights and the learning rates after the schedulers' steps; `init_seeds`:
    the model of seed s is drawn after torch.manual_seed(s), i.e. as `main.py --seed s --n-sims 1` draws it, so that
    recorded single-seed runs of the reference can be replayed as one pack; `init_perturb` = k > 0: every initial
    parameter is moved by ~1e-6 relative, as tools/make_golden.py moved them for the reference's "twin" runs (a
    sensitivity probe: 12 500 Adam steps at lr 1e-2 amplify it to a visibly different trajectory)."""
    st = time.time()
    device = torch.device(config.device)
    print('Executing script on: {}\n'.format(device))
    if config.gamma > 0:
        raise NotImplementedError('FactorVAE discriminator branch (gamma > 0) is outside the fused hot path; every '
                                  'shipped config has gamma: 0')
    dset = ResidentDataset.cached(args.data_path, config.nps, config.ns, config.dl, config.dd, config.nl, config.s,
                                  config.p, config.act, uncentered=config.uncentered, noisy=config.noisy)
    d_data, d_latent, d_aux = dset.get_dims()
    X, U, S = dset.x.to(device), dset.u.to(device), dset.s.to(device)
    N, B = len(dset), config.batch_size
    nb = N // B                               # DataLoader(shuffle=True, drop_last=True), runners/ivae.py:25
    gen = torch.Generator(device=device)
    gen.manual_seed(int(torch.initial_seed()) % (2 ** 63))

    seeds = list(range(args.seed, args.seed + args.n_sims))
    K = len(seeds)
    models = []
    for sd in seeds:
        if init_seeds:
            torch.manual_seed(sd)
        if config.ica:
            models.append(cleanIVAE(data_dim=d_data, latent_dim=d_latent, aux_dim=d_aux, hidden_dim=config.hidden_dim,
                                    n_layers=config.n_layers, activation=config.activation, slope=.1).to(device))
        else:
            models.append(cleanVAE(data_dim=d_data, latent_dim=d_latent, hidden_dim=config.hidden_dim,
                                   n_layers=config.n_layers, activation=config.activation, slope=.1).to(device))
    if init_perturb:
        pgen = torch.Generator().manual_seed(4242 + int(init_perturb))
        with torch.no_grad():
            for model in models:
                for p in model.parameters():
                    p.mul_((1 + 1e-6 * torch.randn(p.shape, generator=pgen)).to(device))
    pack = _SeedPack(models, config.lr)
    schedulers = [_ReduceLROnPlateau(opt, factor=0.1, patience=0) for opt in pack.optimizers]
    loss_hists, perf_hists = [[] for _ in seeds], [[] for _ in seeds]
    tag = (lambda i: '') if K == 1 else (lambda i: '[seed {}] '.format(seeds[i]))

    def finish(epoch, train_loss, cc_host):
        scores = np.array(_assignment_scores(cc_host), dtype=np.float64).reshape(nb, K).sum(0) / nb
        for i in range(K):
            perf_hists[i].append(float(scores[i]))
            print('{}==> Epoch {}/{}:\ttrain loss: {:.6f}\ttrain perf: {:.6f}'.format(tag(i), epoch, config.epochs,
                                                                                       train_loss[i], scores[i]))

    pending = None
    for epoch in range(1, config.epochs + 1):
        for model in models:
            model.train()
        a, b, c, d = config.a, config.b, config.c, config.d
        if config.anneal:                                     # runners/ivae.py:51-60
            c = 0
            if epoch > config.epochs / 1.6:
                a, b, c, d = 2 * config.a, 1, 1, 1
        # one gather per epoch and seed; batches are views
        if replay is None:
            perm = torch.stack([torch.randperm(N, device=device, generator=gen)[:nb * B] for _ in seeds], 1)
            eps = None
        else:
            perm = torch.stack([replay.order(sd, epoch).to(device) for sd in seeds], 1)
            eps = torch.stack([replay.eps(sd, epoch).to(device) for sd in seeds], 1).contiguous()
        perm = perm.view(nb, B, K).transpose(1, 2).contiguous()                      # (nb, K, B)
        Xe, Se = X[perm], S[perm]                                                     # (nb, K, B, .)
        Ue = U[perm] if config.ica else None
        Ze = torch.empty(nb, K, B, d_latent, device=device)
        losses, _ = pack.epoch(Xe, Ue, a, b, c, d, N, eps=eps, z_out=Ze)              # (nb, K)
        # the per-step batch scores (runners/ivae.py:86) of all seeds, one launch per epoch
        cc = ops.corr_batched(Se.reshape(nb * K, B, d_latent), Ze.view(nb * K, B, d_latent))
        # The assignment problems of the PREVIOUS epoch's batch scores are host work (Hungarian step on the host, as in the
        # reference): they are solved here, while the GPU runs the launches just queued -- those needed the previous epoch's
        # losses (the schedulers below) but not its scores.  An epoch's line is printed one epoch late, in order.
        if pending is not None:
            finish(*pending)
            pending = None
        # the epoch's one sync; summed as the reference does (`train_loss += loss.item()`, runners/ivae.py:84,112: float64,
        # step by step), which also makes a seed's figure independent of how many seeds are packed beside it -- for
        # packs of fewer than 128 models: from 128 on the in-workgroup pairwise kernel walks with 4 waves per model instead
        # of 16 / 8 and sums its partials in another grouping (last-bit differences, tests/test_gpu_sweep.py)
        train_loss = losses.cpu().numpy().astype(np.float64).sum(0) / nb
        cc_host = cc.abs().cpu().numpy()
        for i in range(K):
            loss_hists[i].append(float(train_loss[i]))
            if not config.no_scheduler:
                schedulers[i].step(float(train_loss[i]))
        if trace is not None:
            trace.setdefault('step_loss', []).append(losses.cpu())
            trace.setdefault('coef', []).append((a, b, c, d))
            trace.setdefault('lr', []).append([opt.param_groups[0]['lr'] for opt in pack.optimizers])
        pending = (epoch, train_loss, cc_host)
    if pending is not None:
        finish(*pending)
    print('\ntotal runtime: {}'.format(time.time() - st))

    # full-dataset score on a SAMPLE of q(z|x,u), as the reference does (runners/ivae.py:118-123)
    perfs = []
    with torch.no_grad():
        for sd, model in zip(seeds, models):
            ev = None if replay is None else replay.eval_eps(sd).to(device)
            s = model(X, U, eps=ev)[3] if config.ica else model(X, eps=ev)[3]
            perfs.append(mcc(S, s))
    return perfs, loss_hists, perf_hists


# ------------------------------------------------------------------------------------------------ family B
_PLAIN = (1., 1., 1., 1., 1., False)          # loss weights (a, b, c, d, N, pairwise) of the plain ELBO


class _FusedEpochs:
    """Epoch driver for models the fused kernel covers: full batches go through `ivae_train_steps` (the model's
    own arena is the kernel's parameter buffer), a ragged final batch through the generic kernels with the same
    Adam state."""

    def __init__(self, model, lr, seed):
        self.model = model
        self.shape = model._shape
        self.params = model._flat_params()
        self.m = torch.zeros_like(self.params)
        self.v = torch.zeros_like(self.params)
        self.t = 0
        self.seed = 0 if seed is None else int(seed)
        self.coef = ops.make_coef(decoder_var=model._decoder_var_f, global_batch=64)
        self.ws = ops.Workspace()
        self.group = {'lr': lr, 'betas': (0.9, 0.999), 'eps': 1e-8}
        self.param_groups = [self.group]           # what ReduceLROnPlateau mutates

    def full_batches(self, X, U, idx, eps=None):
        """idx: (k, 64) int32 rows [eps (k, 64, d): the steps' noise instead of the kernel's own generator].
        -> (loss (k,), z (k, 64, d)) on the device."""
        loss, z = ops.train_steps(self.shape, self.coef, self.params.view(1, -1), self.m.view(1, -1), self.v.view(1, -1),
                                  X.unsqueeze(0), None if U is None else U.unsqueeze(0), idx.unsqueeze(0), self.t,
                                  self.group['lr'], eps=None if eps is None else eps.unsqueeze(0).contiguous(),
                                  seed=self.seed, want_z_steps=True)
        self.t += idx.shape[0]
        return loss[0], z[0]

    def ragged_batch(self, x, u, eps=None):
        if eps is None:
            eps = torch.randn(x.shape[0], self.shape.latent_dim, device=x.device)
        coef = ops.make_coef(decoder_var=self.model._decoder_var_f, global_batch=x.shape[0])
        loss, z, grads = ops.elbo_fwd_bwd(self.shape, coef, self.params, x, u, eps, self.ws)
        self.t += 1
        ops.adam_step(self.params, self.m, self.v, grads, self.group['lr'], self.t)
        return loss[0], z

    def as_optimizer(self):
        """A FusedAdam holding this state, for `checkpoint()` (state_dict layout of torch.optim.Adam)."""
        opt = FusedAdam(self.model.parameters(), lr=self.group['lr'])
        off = 0
        for p in self.model.parameters():
            n = p.numel()
            opt.state[p] = {'step': torch.tensor(float(self.t)), 'exp_avg': self.m[off:off + n].view(p.shape),
                            'exp_avg_sq': self.v[off:off + n].view(p.shape)}
            off += n
        return opt


def _train_family_b(model, dset, S, batch_size, max_iter, lr, device, anneal, log_folder, ckpt_folder, scheduler_tol,
                    seed, with_u, replay=None, trace=None):
    """The loop of models/wrappers.py:54-95 / :137-176.  Test hooks, as in `runner` (tests/test_gpu_wrapper_trace.py replays
    runs recorded from the reference's wrappers draw for draw): `replay.order(epoch)` the epoch's N rows in DataLoader
    order, `replay.eps(it, rows)` the noise of iteration `it` (1-based), `replay.eval_eps()` that of the final full-data
    forward, `replay.init_state()` (optional) initial parameters to start from instead of the drawn ones; `trace` (a dict) receives per-iteration losses and loss weights, the learning rate after each epoch and
    the iterations that got a checkpoint."""
    N = len(dset)
    X = dset.x.to(device)
    U = dset.u.to(device) if with_u else None
    Z = dset.s.to(device) if S is not None else None
    n_batches = -(-N // batch_size)
    max_epochs = int(max_iter // n_batches + 1)
    gen = torch.Generator(device=device)
    gen.manual_seed(int(torch.initial_seed()) % (2 ** 63))
    fused = (not anneal and batch_size == 64 and ops.train_steps_supported(model._shape, 64))
    if trace is not None:
        trace['init'] = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    if replay is not None and hasattr(replay, 'init_state'):
        # a run recorded on a CPU: iVAE redraws its weights with xavier_uniform_ AFTER the move to the device (nets.py:432,
        # as the reference does), i.e. from the device's generator, which no CPU run can reproduce
        with torch.no_grad():
            for k, v in model.state_dict().items():
                if k in replay.init_state():
                    v.copy_(replay.init_state()[k])
    if fused:
        drv = _FusedEpochs(model, lr, seed)
        # ReduceLROnPlateau wants an Optimizer: a placeholder holds the learning rate the kernel is launched with
        scheduler = _ReduceLROnPlateau(optim.SGD([torch.zeros(1, requires_grad=True)], lr=lr), factor=0.1,
                                       patience=scheduler_tol)
        optimizer = None
    else:
        optimizer = FusedAdam(model.parameters(), lr=lr)
        scheduler = _ReduceLROnPlateau(optimizer, factor=0.1, patience=scheduler_tol)

    logger = Logger(log_dir=log_folder)
    exp_id = logger.exp_id
    if log_folder is None:
        ckpt_folder = None
    logger.add('elbo')
    logger.add('perf')
    ckpt_every = int(max_iter / 5)
    d = model.latent_dim

    def last(key):
        vals = logger.log_dict[key]
        return vals[-1] if vals else float('nan')       # (the reference would raise here inside the first epoch)

    pending = []          # [(epoch, mean loss, |correlation| blocks on the host)] of the epoch whose scores are still due

    def finish_scores():
        """The batch scores (wrappers.py:75) of the epoch before: assignment problems on the host, solved while the GPU runs
        the launches of the current epoch (which needed that epoch's loss for the scheduler, not its scores)."""
        while pending:
            ep, elbo, cc_host = pending.pop(0)
            for score in _assignment_scores(cc_host):
                logger.update('perf', score)
            logger.log(keys=('perf',))
            print('epoch {}/{} \tloss: {}\tperf: {}'.format(ep, max_epochs, elbo, logger.get_last('perf')))

    def maybe_checkpoint(it):
        """wrappers.py:78-80: a checkpoint whenever `it` is a multiple of max_iter / 5, holding the state AT that iteration."""
        if ckpt_folder is not None and ckpt_every > 0 and it % ckpt_every == 0:
            finish_scores()                   # the checkpoint stores the last logged epoch's loss and score
            opt = drv.as_optimizer() if fused else optimizer
            if trace is not None:
                trace.setdefault('ckpt_its', []).append(it)
            checkpoint(ckpt_folder, exp_id, it, model, opt, last('elbo'), last('perf'))

    def segments(it, n):
        """[a, b) pieces of the next n iterations, cut where a checkpoint falls due (so every piece is one library call
        and the state a checkpoint stores is the state of its iteration)."""
        cuts, a = [], 0
        if ckpt_folder is not None and ckpt_every > 0:
            m = (it // ckpt_every + 1) * ckpt_every
            while m < it + n:
                cuts.append(m - it)
                m += ckpt_every
        for c in cuts + [n]:
            if c > a:
                yield a, c
            a = c

    print("Training..")
    it = 0
    model.train()
    while it < max_iter:
        epoch = it // n_batches + 1
        if replay is None:
            batches = _epoch_batches(N, batch_size, False, device, gen)
        else:
            order = replay.order(epoch).to(device)
            batches = [order[i:i + batch_size] for i in range(0, N, batch_size)]

        it0 = it

        def noise(a, b):         # batches a .. b - 1 of this epoch of the replayed run (None: the device generators draw)
            if replay is None:
                return None
            return torch.stack([replay.eps(it0 + 1 + k, batches[k].numel()) for k in range(a, b)]).to(device)

        def note(losses, coefs):
            if trace is not None:
                trace.setdefault('step_loss', []).append(losses.detach().reshape(-1).cpu())
                trace.setdefault('step_coef', []).extend(coefs)
        cc = torch.empty(len(batches), d, d, dtype=torch.float64, device=device) if S is not None else None
        n_full = N // batch_size
        if fused:
            drv.group['lr'] = scheduler.optimizer.param_groups[0]['lr']
            if n_full:
                idx = torch.stack(batches[:n_full]).to(torch.int32)
                zs = torch.empty(n_full, batch_size, d, device=device)
                for a0, b0 in segments(it, n_full):
                    loss, zs[a0:b0] = drv.full_batches(X, U, idx[a0:b0].contiguous(), eps=noise(a0, b0))
                    logger.update('elbo', loss.sum(), weight=b0 - a0)
                    note(loss, [_PLAIN] * (b0 - a0))
                    it += b0 - a0
                    maybe_checkpoint(it)
                if cc is not None:              # the per-step batch scores (wrappers.py:75), one launch per epoch
                    cc[:n_full] = ops.corr_batched(zs, Z[idx.long().view(-1)].view(n_full, batch_size, d))
            for i in range(n_full, len(batches)):
                e1 = noise(i, i + 1)
                l1, z1 = drv.ragged_batch(X[batches[i]], None if U is None else U[batches[i]], None if e1 is None else e1[0])
                logger.update('elbo', l1)
                note(l1, [_PLAIN])
                if cc is not None:
                    cc[i] = corr_matrix_gpu(z1, Z[batches[i]])
                it += 1
                maybe_checkpoint(it)
        else:
            Ze = torch.empty(n_full, batch_size, d, device=device) if cc is not None else None
            if n_full:
                # the epoch's full batches go through ivae_elbo_adam_steps, one call per piece; with anneal on, the loss
                # weights of every iteration of the piece (model.anneal is called before each step, wrappers.py:62-63)
                # travel with the call as one coefficient set per step
                idx = torch.stack(batches[:n_full])
                for a0, b0 in segments(it, n_full):
                    coefs = None
                    if anneal:
                        coefs = []
                        for k in range(a0, b0):
                            model.anneal(N, max_iter, it + 1 + k - a0)
                            if with_u and model.anneal_params:
                                coefs.append(tuple(model._training_hyperparams) + (True,))
                            else:
                                coefs.append((1., 1., 1., 1., 1., False))
                    losses, _ = model.direct_steps(optimizer, X[idx[a0:b0]], U[idx[a0:b0]] if with_u else None, 1., 1., 1.,
                                                   1., 1., False, eps=noise(a0, b0),
                                                   z_out=None if Ze is None else Ze[a0:b0], coefs=coefs)
                    logger.update('elbo', losses.sum(), weight=b0 - a0)
                    note(losses, coefs or [_PLAIN] * (b0 - a0))
                    it += b0 - a0
                    maybe_checkpoint(it)
            for i in range(n_full, len(batches)):        # the ragged final batch (wrappers.py:28 has no drop_last)
                bidx = batches[i]
                e1 = noise(i, i + 1)
                e1 = None if e1 is None else e1[0]
                it += 1
                if anneal:
                    model.anneal(N, max_iter, it)
                # zero_grad -> elbo -> mul(-1).backward() -> step (wrappers.py:64-70) as one launch sequence; the loss
                # coefficients are what model.elbo would pass (nets.py:457-475)
                if with_u and model.anneal_params:
                    a_, b_, c_, d_, N_ = model._training_hyperparams
                    loss, z_est = model.direct_step(optimizer, X[bidx], U[bidx], a_, b_, c_, d_, N_, True, eps=e1)
                    note(loss, [(a_, b_, c_, d_, N_, True)])
                else:
                    loss, z_est = model.direct_step(optimizer, X[bidx], U[bidx] if with_u else None, 1., 1., 1., 1., 1.,
                                                    False, eps=e1)
                    note(loss, [_PLAIN])
                logger.update('elbo', loss)
                if cc is not None:
                    cc[i] = corr_matrix_gpu(z_est, Z[bidx])
                maybe_checkpoint(it)
            if Ze is not None and n_full:
                cc[:n_full] = ops.corr_batched(Ze, Z[torch.stack(batches[:n_full]).view(-1)].view(n_full, batch_size, d))
        finish_scores()                       # the previous epoch's, under this epoch's launches (all queued by now)
        if cc is not None:
            cc_host = cc.abs().cpu().numpy()
            logger.log(keys=('elbo',))
            pending.append((epoch, logger.get_last('elbo'), cc_host))
        else:
            logger.log()
            print('epoch {}/{} \tloss: {}'.format(epoch, max_epochs, logger.get_last('elbo')))
        scheduler.step(logger.get_last('elbo'))
        if trace is not None:
            trace.setdefault('lr', []).append(scheduler.optimizer.param_groups[0]['lr'])
    finish_scores()

    with torch.no_grad():
        ev = None if replay is None else replay.eval_eps().to(device)
        decoder_params, encoder_params, z, prior_params = model(X, U, eps=ev) if with_u else model(X, eps=ev)
    params = {'decoder': decoder_params, 'encoder': encoder_params, 'prior': prior_params}
    return z, model, params, logger


def IVAE_wrapper(X, U, S=None, batch_size=256, max_iter=7e4, seed=None, n_layers=3, hidden_dim=200, lr=1e-2, cuda=True,
                 activation='lrelu', slope=.1, discrete=False, inference_dim=None,
                 anneal=False, log_folder=None, ckpt_folder=None, scheduler_tol=3, *, replay=None, trace=None):
    """models/wrappers.py:14-95.  Returns (z, model, params, logger).  (`replay`, `trace`: test hooks, see _train_family_b.)"""
    if discrete:
        raise NotImplementedError('DiscreteIVAE is outside the fused hot path (SURVEY.md §2 row 15)')
    if not cuda:
        raise RuntimeError('ivae_amd has no CPU path: cuda=False is not available')
    if seed is not None:
        torch.manual_seed(seed)
        np.random.seed(seed)
    device = torch.device('cuda')
    print('training on {}'.format(torch.cuda.get_device_name(device)))
    dset = ResidentDataset(X, U, S)
    data_dim, latent_dim, aux_dim = dset.get_dims()
    if inference_dim is not None:
        latent_dim = inference_dim
    model = iVAE(latent_dim, data_dim, aux_dim, activation=activation, device=device, n_layers=n_layers,
                 hidden_dim=hidden_dim, slope=slope, anneal=anneal)
    return _train_family_b(model, dset, S, batch_size, max_iter, lr, device, anneal, log_folder, ckpt_folder,
                           scheduler_tol, seed, with_u=True, replay=replay, trace=trace)


def VAE_wrapper(X, S=None, batch_size=256, max_iter=7e4, seed=None, n_layers=3, hidden_dim=200, lr=1e-2, cuda=True,
                activation='lrelu', slope=.1, discrete=False, inference_dim=None,
                log_folder=None, ckpt_folder=None, scheduler_tol=3, *, replay=None, trace=None):
    """models/wrappers.py:98-176."""
    if discrete:
        raise NotImplementedError('DiscreteVAE is outside the fused hot path (SURVEY.md §2 row 15)')
    if not cuda:
        raise RuntimeError('ivae_amd has no CPU path: cuda=False is not available')
    if seed is not None:
        torch.manual_seed(seed)
        np.random.seed(seed)
    device = torch.device('cuda')
    print('training on {}'.format(torch.cuda.get_device_name(device)))
    dset = ResidentDataset(X, X, S)
    data_dim, latent_dim, _ = dset.get_dims()
    if inference_dim is not None:
        latent_dim = inference_dim
    model = VAE(latent_dim, data_dim, activation=activation, device=device, n_layers=n_layers, hidden_dim=hidden_dim,
                slope=slope)
    return _train_family_b(model, dset, S, batch_size, max_iter, lr, device, False, log_folder, ckpt_folder,
                           scheduler_tol, seed, with_u=False, replay=replay, trace=trace)
