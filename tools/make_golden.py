#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the reference implementation.

Runs ONLY in the development container (the reference lives read-only at
/root/reference and never travels to the GPU box).  This is the one file in the
repo that imports the reference; it copies none of its source.  Each fixture
holds inputs and the reference's outputs for one hot-path case:

  params.*        state_dict before the step          x, u, eps     inputs
  g v l f z       forward tensors                     lpx lqc lp    per-row log-densities
  loss            the value to MINIMISE (family A's return; minus family B's)
  grad.*          d loss / d param from loss.backward()
  adam{k}.*       params / exp_avg / exp_avg_sq after k optimiser steps
                  (steps 1-3 at lr 1e-2, step 4 after an lr drop to 1e-3)

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py
"""
import json
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

from models.nets import iVAE, VAE, cleanIVAE, cleanVAE, Normal, log_normal  # noqa: E402
from metrics.mcc import mean_corr_coef_np  # noqa: E402


def onehot_u(B, M, gen):
    u = torch.zeros(B, M)
    u[torch.arange(B), torch.randint(0, M, (B,), generator=gen)] = 1
    return u


def make_case(name, family, dims, B, coef=None, anneal_it=None, max_iter=1000, N=40000, seed=0,
              act="lrelu", H=50, n_layers=3, slope=.1, dense_u=False, store_steps=True):
    torch.manual_seed(seed)
    d_lat, d_data, M = dims
    if family == "iVAE":
        model = iVAE(d_lat, d_data, M, n_layers=n_layers, hidden_dim=H, activation=act, slope=slope,
                     anneal=anneal_it is not None)
    elif family == "VAE":
        model = VAE(d_lat, d_data, n_layers=n_layers, hidden_dim=H, activation=act, slope=slope)
    elif family == "cleanIVAE":
        model = cleanIVAE(d_data, d_lat, M, n_layers=n_layers, activation=act, hidden_dim=H, slope=slope)
    elif family == "cleanVAE":
        model = cleanVAE(d_data, d_lat, n_layers=n_layers, activation=act, hidden_dim=H, slope=slope)
    else:
        raise ValueError(family)
    has_u = family in ("iVAE", "cleanIVAE")
    clean = family.startswith("clean")
    gen = torch.Generator().manual_seed(1000 + seed)
    out = {}
    meta = dict(name=name, family=family, latent_dim=d_lat, data_dim=d_data, aux_dim=M if has_u else 0,
                hidden_dim=H, n_layers=n_layers, activation=act, slope=slope, batch=B, N=N,
                decoder_var=float(model.decoder_var.item()), torch=torch.__version__, anneal_it=anneal_it, max_iter=max_iter)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)

    def one_step(k):
        x = torch.randn(B, d_data, generator=gen) * 1.5
        if has_u:
            u = torch.rand(B, M, generator=gen) if dense_u else onehot_u(B, M, gen)
        else:
            u = torch.zeros(B, 1)
        if anneal_it is not None:
            model.anneal(N, max_iter, anneal_it + k)
        opt.zero_grad()
        st = torch.get_rng_state()
        if clean:
            a, b, c, d = coef
            loss, z = model.elbo(x, u, N, a=a, b=b, c=c, d=d)
            co = dict(a=a, b=b, c=c, d=d, N=N, pairwise=True)
        else:
            pair = bool(getattr(model, "anneal_params", False))
            if pair:
                a, b, c, d, Nn = model._training_hyperparams
                co = dict(a=a, b=b, c=c, d=d, N=Nn, pairwise=True)
            else:
                co = dict(a=1., b=1., c=1., d=1., N=N, pairwise=False)
            elbo, z = model.elbo(x, u) if has_u else model.elbo(x)
            loss = elbo.mul(-1)
        # recover eps exactly: the reference draws torch.randn(B, d) (nets.py:147,302)
        torch.set_rng_state(st)
        eps = torch.randn(B, d_lat)
        torch.set_rng_state(st)
        with torch.no_grad():
            fw = model(x, u) if has_u else model(x)
        if clean:
            f, g, v, z2 = fw[0], fw[1], fw[2], fw[3]
            l = fw[4] if has_u else torch.ones_like(g)
        else:
            (f, _), (g, v), z2, (_, l) = fw
            if not has_u:
                l = torch.ones_like(g)
        assert torch.equal(z2, z.detach()), "rng replay failed"
        assert torch.equal(z2, eps.mul(v.sqrt()).add(g)), "eps recovery failed"
        with torch.no_grad():
            dvar = model.decoder_var
            if clean:
                lpx = log_normal(x, f, dvar).sum(-1)
                lqc = log_normal(z2, g, v).sum(-1)
                lp = log_normal(z2, None, l).sum(-1)
            else:
                nd = Normal()
                lpx = nd.log_pdf(x, f, dvar)
                lqc = nd.log_pdf(z2, g, v)
                lp = nd.log_pdf(z2, torch.zeros(1), l)
        rec = dict(x=x, u=u, eps=eps, g=g, v=v, l=l.expand_as(g), f=f, z=z2, lpx=lpx, lqc=lqc, lp=lp,
                   loss=loss.detach())
        return loss, rec, co

    # step 0 record (pre-update params, grads)
    for kname, val in model.state_dict().items():
        out["params." + kname] = val.detach().clone().numpy()
    loss, rec, co = one_step(0)
    loss.backward()
    for k_, v_ in rec.items():
        out[k_] = v_.numpy()
    for kname, p in model.named_parameters():
        out["grad." + kname] = p.grad.detach().clone().numpy()
    meta["coef"] = co
    opt.step()
    step_meta = [co]

    def dump_adam(k):
        sd = opt.state_dict()["state"]
        for i, (kname, p) in enumerate(model.named_parameters()):
            out[f"adam{k}.p.{kname}"] = p.detach().clone().numpy()
            out[f"adam{k}.m.{kname}"] = sd[i]["exp_avg"].clone().numpy()
            out[f"adam{k}.v.{kname}"] = sd[i]["exp_avg_sq"].clone().numpy()

    if store_steps:
        dump_adam(1)
    for k in (1, 2, 3):
        if k == 3:
            for gparam in opt.param_groups:
                gparam["lr"] = 1e-3
        loss, rec, co = one_step(k)
        loss.backward()
        opt.step()
        step_meta.append(co)
        for k_ in ("x", "u", "eps", "loss"):
            out[f"step{k}.{k_}"] = rec[k_].numpy()
        if store_steps or k == 3:
            dump_adam(k + 1)
    meta["steps"] = step_meta
    meta["lrs"] = [1e-2, 1e-2, 1e-2, 1e-3]
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name:28s} loss={float(out['loss']):.6f} params={sum(v.size for k, v in out.items() if k.startswith('params.'))}")


def make_mcc():
    rng = np.random.RandomState(7)
    out = {}
    cases = [(64, 10), (64, 5), (256, 20), (5000, 10), (37, 3)]
    for i, (n, d) in enumerate(cases):
        s = rng.randn(n, d).astype(np.float32)
        mix = rng.randn(d, d).astype(np.float32)
        y = (np.tanh(s @ mix) + 0.3 * rng.randn(n, d)).astype(np.float32)
        out[f"x{i}"], out[f"y{i}"] = s, y
        out[f"cc{i}"] = np.corrcoef(s, y, rowvar=False)[:d, d:]
        out[f"mcc{i}"] = np.array(mean_corr_coef_np(s, y))
    out["n_cases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(OUT, "mcc.npz"), **out)
    print("mcc fixtures", [float(out[f"mcc{i}"]) for i in range(len(cases))])


DATASET_CASES = [
    dict(n_per_seg=50, n_seg=6, d_sources=3, d_data=3, n_layers=3, seed=1, prior='gauss', activation='xtanh'),
    dict(n_per_seg=40, n_seg=5, d_sources=4, d_data=4, n_layers=3, seed=7, prior='gauss', activation='lrelu'),
    dict(n_per_seg=30, n_seg=4, d_sources=2, d_data=4, n_layers=2, seed=3, prior='lap', activation='sigmoid'),
    dict(n_per_seg=30, n_seg=4, d_sources=3, d_data=3, n_layers=4, seed=5, prior='hs', activation='none',
         uncentered=True, noisy=0.1),
    dict(n_per_seg=1000, n_seg=40, d_sources=5, d_data=5, n_layers=3, seed=1, prior='gauss', activation='xtanh'),   # configs/ivae.yaml
    dict(n_per_seg=1000, n_seg=40, d_sources=10, d_data=10, n_layers=3, seed=1, prior='gauss', activation='lrelu'),  # BASELINE cfg 2
]


def make_datasets():
    """sha256 + leading values of the reference generator's outputs (data/data.py:200-304, called as
    load_tcl_data does: repeat_linearity=True, batch_size=0).  np.int was removed from numpy 2: the reference
    needs the in-process alias below to run at all."""
    import hashlib
    if not hasattr(np, "int"):
        np.int = int
    from data.data import generate_data
    out = []
    for kw in DATASET_CASES:
        S, X, U, M, L = generate_data(batch_size=0, repeat_linearity=True, **kw)
        rec = dict(kwargs=kw)
        for k, a in (("s", S), ("x", X), ("u", U), ("m", M), ("L", L)):
            a = np.ascontiguousarray(a)
            rec[k] = dict(sha256=hashlib.sha256(a.tobytes()).hexdigest(), dtype=str(a.dtype), shape=list(a.shape),
                          head=[float(v) for v in a.reshape(-1)[:6]], sum=float(a.astype(np.float64).sum()))
        out.append(rec)
        print("dataset", kw["n_per_seg"], kw["n_seg"], kw["d_sources"], rec["x"]["sha256"][:12])
    json.dump(out, open(os.path.join(OUT, "datasets.json"), "w"), indent=1)


def trace_batches(seed, epoch, N, B):
    """Batch order used by the trace fixtures (DataLoader(shuffle=True, drop_last=True) semantics)."""
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(100 * seed + epoch))
    return perm[:(N // B) * B].view(-1, B)


def make_traces(n_seeds=4, epochs=2, B=64):
    """Reference iVAE (BASELINE config 2) trained on CPU for `epochs` epochs with a recorded batch order
    and per-step noise seeds -> per-step loss trace and final MCC, for the end-to-end parity test
    (tests/test_gpu_e2e.py replays the same batches and eps through the fused kernel)."""
    if not hasattr(np, "int"):
        np.int = int
    from data.data import generate_data
    S, X, U, _, _ = generate_data(1000, 40, 10, 10, n_layers=3, seed=1, prior='gauss', activation='lrelu',
                                  batch_size=0, repeat_linearity=True)
    Xt, Ut = torch.from_numpy(X), torch.from_numpy(U)
    N = X.shape[0]
    out = dict(n_seeds=np.array(n_seeds), epochs=np.array(epochs), batch=np.array(B))
    for seed in range(n_seeds):
        torch.manual_seed(seed)
        model = iVAE(10, 10, 40, n_layers=3, hidden_dim=50, activation='lrelu', slope=.1)
        opt = torch.optim.Adam(model.parameters(), lr=1e-2)
        losses, step = [], 0
        for epoch in range(epochs):
            for idx in trace_batches(seed, epoch, N, B):
                torch.manual_seed(1_000_000 * (seed + 1) + step)      # the elbo's randn(B, d) draw
                opt.zero_grad()
                elbo, z = model.elbo(Xt[idx], Ut[idx])
                elbo.mul(-1).backward()
                opt.step()
                losses.append(-elbo.item())
                step += 1
        with torch.no_grad():
            torch.manual_seed(777 + seed)
            (_, _), (g, v), z, _ = model(Xt, Ut)
        out[f"loss{seed}"] = np.array(losses, np.float32)
        out[f"mcc_sample{seed}"] = np.array(mean_corr_coef_np(S, z.numpy()))
        out[f"mcc_mean{seed}"] = np.array(mean_corr_coef_np(S, g.numpy()))
        print(f"trace seed {seed}: loss {losses[0]:.2f} -> {np.mean(losses[-50:]):.3f}  mcc(sample) "
              f"{float(out[f'mcc_sample{seed}']):.4f} mcc(mean) {float(out[f'mcc_mean{seed}']):.4f}")
    np.savez_compressed(os.path.join(OUT, "trace_cfg2.npz"), **out)


def make_traces_long(n_seeds=4, epochs=10, B=64, n_twins=2):
    """make_traces at 10 epochs (6250 steps): the reference iVAE of BASELINE configs[1] per seed, plus n_twins runs with
    the initial parameters moved by 1e-6 relative -> tests/golden/trace_cfg2_long.npz (per-epoch mean losses, first and
    last step losses, final MCC of a sampled z and of the mean g).  Batch order and noise are reproducible from the seeds
    (trace_batches, the per-step manual_seed)."""
    torch.set_num_threads(1)
    if not hasattr(np, "int"):
        np.int = int
    from data.data import generate_data
    S, X, U, _, _ = generate_data(1000, 40, 10, 10, n_layers=3, seed=1, prior='gauss', activation='lrelu',
                                  batch_size=0, repeat_linearity=True)
    Xt, Ut = torch.from_numpy(X), torch.from_numpy(U)
    N = X.shape[0]
    nb = N // B

    def run(seed, perturb):
        torch.manual_seed(seed)
        model = iVAE(10, 10, 40, n_layers=3, hidden_dim=50, activation='lrelu', slope=.1)
        if perturb:
            gen = torch.Generator().manual_seed(4242 + perturb)
            with torch.no_grad():
                for p in model.parameters():
                    p.mul_(1 + 1e-6 * torch.randn(p.shape, generator=gen))
        opt = torch.optim.Adam(model.parameters(), lr=1e-2)
        losses, step = [], 0
        for epoch in range(epochs):
            for idx in trace_batches(seed, epoch, N, B):
                torch.manual_seed(1_000_000 * (seed + 1) + step)
                opt.zero_grad()
                elbo, z = model.elbo(Xt[idx], Ut[idx])
                elbo.mul(-1).backward()
                opt.step()
                losses.append(-elbo.item())
                step += 1
        with torch.no_grad():
            torch.manual_seed(777 + seed)
            (_, _), (g, v), z, _ = model(Xt, Ut)
        return np.array(losses, np.float32), float(mean_corr_coef_np(S, z.numpy())), float(mean_corr_coef_np(S, g.numpy()))

    out = dict(n_seeds=np.array(n_seeds), epochs=np.array(epochs), batch=np.array(B))
    for seed in range(n_seeds):
        runs = [run(seed, k) for k in range(n_twins + 1)]
        L = np.stack([r[0] for r in runs])
        out[f"first_losses{seed}"] = L[0, :20]
        out[f"epoch_loss{seed}"] = L.reshape(n_twins + 1, epochs, nb).astype(np.float64).mean(2)       # row 0: the run, rows 1..: twins
        out[f"late_loss{seed}"] = L[:, -200:].astype(np.float64).mean(1)
        out[f"mcc_sample{seed}"] = np.array([r[1] for r in runs])
        out[f"mcc_mean{seed}"] = np.array([r[2] for r in runs])
        print(f"long trace seed {seed}: epoch loss {out[f'epoch_loss{seed}'][0].round(2).tolist()} mcc(sample) "
              f"{out[f'mcc_sample{seed}'].round(4).tolist()} mcc(mean) {out[f'mcc_mean{seed}'].round(4).tolist()}", flush=True)
    np.savez_compressed(os.path.join(OUT, "trace_cfg2_long.npz"), **out)


def make_trace_twins(n_twins=4):
    """Sensitivity of the REFERENCE trajectory of make_traces: the same runs with the initial parameters moved by ~1e-6
    relative (a few fp32 roundings; same batches, same noise).  1250 Adam steps at lr 1e-2 amplify that; the spread of
    the twins' final MCC and late losses is what tests/test_gpu_e2e.py holds ivae_amd to, instead of a guessed tolerance.
    Adds sens_* arrays to tests/golden/trace_cfg2.npz after checking that the stored baseline reproduces bit for bit."""
    if not hasattr(np, "int"):
        np.int = int
    from data.data import generate_data
    path = os.path.join(OUT, "trace_cfg2.npz")
    old = dict(np.load(path))
    n_seeds, epochs, B = int(old["n_seeds"]), int(old["epochs"]), int(old["batch"])
    S, X, U, _, _ = generate_data(1000, 40, 10, 10, n_layers=3, seed=1, prior='gauss', activation='lrelu',
                                  batch_size=0, repeat_linearity=True)
    Xt, Ut = torch.from_numpy(X), torch.from_numpy(U)
    N = X.shape[0]

    def run(seed, perturb):
        torch.manual_seed(seed)
        model = iVAE(10, 10, 40, n_layers=3, hidden_dim=50, activation='lrelu', slope=.1)
        if perturb:
            gen = torch.Generator().manual_seed(4242 + perturb)
            with torch.no_grad():
                for p in model.parameters():
                    p.mul_(1 + 1e-6 * torch.randn(p.shape, generator=gen))
        opt = torch.optim.Adam(model.parameters(), lr=1e-2)
        losses, step = [], 0
        for epoch in range(epochs):
            for idx in trace_batches(seed, epoch, N, B):
                torch.manual_seed(1_000_000 * (seed + 1) + step)
                opt.zero_grad()
                elbo, z = model.elbo(Xt[idx], Ut[idx])
                elbo.mul(-1).backward()
                opt.step()
                losses.append(-elbo.item())
                step += 1
        with torch.no_grad():
            torch.manual_seed(777 + seed)
            (_, _), (g, v), z, _ = model(Xt, Ut)
        return np.array(losses, np.float32), float(mean_corr_coef_np(S, z.numpy())), float(mean_corr_coef_np(S, g.numpy()))

    for seed in range(n_seeds):
        base = run(seed, 0)
        assert np.array_equal(base[0], old[f"loss{seed}"]) and base[1] == float(old[f"mcc_sample{seed}"]), "baseline does not reproduce"
        tw = [run(seed, k) for k in range(1, n_twins + 1)]
        old[f"sens_loss{seed}"] = np.stack([t[0] for t in tw])
        old[f"sens_mcc_sample{seed}"] = np.array([t[1] for t in tw])
        old[f"sens_mcc_mean{seed}"] = np.array([t[2] for t in tw])
        print(f"twins seed {seed}: mcc_sample {base[1]:.4f} twins {old[f'sens_mcc_sample{seed}'].round(4).tolist()}  "
              f"mcc_mean {base[2]:.4f} twins {old[f'sens_mcc_mean{seed}'].round(4).tolist()}")
    np.savez_compressed(path, **old)


# ---- family-A loop traces: the reference's own runners/ivae.py::runner, run unmodified ------------------------------
RUNNER_CASES = [
    # name, yaml file of the reference, config overrides, seeds, epochs
    ("ivae", "ivae.yaml", {}, (0, 1), 2),
    ("btcvae", "btcvae.yaml", {}, (0, 1), 2),
    ("ivae_anneal", "ivae.yaml", {"anneal": True}, (0,), 3),     # epoch > epochs/1.6 switches the weights at epoch 2
    ("vae", "vae.yaml", {}, (0,), 2),                            # the other two yaml files of the reference (cleanVAE with
    ("bvae", "bvae.yaml", {}, (0,), 2),                          # the VAE / beta-VAE loss weights): every config is replayed
]


def noise_seed(seed, step):
    """Seed of the torch.randn(B, d) draw of training step `step` (the trace fixes it so the draw can be replayed)."""
    return 1_000_000 * (seed + 1) + step


def eval_seed(seed):
    return 777 + seed


def make_runner_traces(only=()):
    """Run `runners.ivae.runner(args, config)` of the reference (runners/ivae.py:13-128) exactly as main.py:47-62 does --
    np.random.seed / torch.manual_seed(seed), yaml -> Namespace -- on CPU, with recording shims only:
      * the dataset's __getitem__ logs the row order the DataLoader(shuffle=True, drop_last=True) asked for;
      * the model's elbo() seeds the global generator per step (noise_seed) so that its randn draw can be replayed, and
        logs the loss and the (a, b, c, d) it was called with; the full-dataset forward of :118-122 is seeded by eval_seed;
      * ReduceLROnPlateau drops the `verbose` argument current torch rejects and logs the learning rate after step();
      * np.int alias for numpy 2 (data/data.py:32).
    -> tests/golden/trace_runner.npz: per run the batch order, per-step losses, per-epoch loss / perf / lr / weights and
    the final full-data MCC, replayed through ivae_amd.runners.runner by tests/test_gpu_runner_trace.py."""
    import argparse
    import tempfile
    import yaml
    if not hasattr(np, "int"):
        np.int = int
    import runners.ivae as R

    rec = argparse.Namespace()

    class RecDataset(R.SyntheticDataset):
        def __getitem__(self, index):
            rec.order.append(int(index))
            return super().__getitem__(index)

    def recording(base):
        class Rec(base):
            def __init__(self, *a, **kw):
                super().__init__(*a, **kw)
                if rec.perturb:      # sensitivity runs: initial parameters moved by ~1e-6 relative (a few fp32 roundings)
                    gen = torch.Generator().manual_seed(4242 + rec.perturb)
                    with torch.no_grad():
                        for p in self.parameters():
                            p.mul_(1 + 1e-6 * torch.randn(p.shape, generator=gen))

            def elbo(self, x, u, N, a=1., b=1., c=1., d=1.):
                torch.manual_seed(noise_seed(rec.seed, rec.step))
                loss, z = super().elbo(x, u, N, a=a, b=b, c=c, d=d)
                rec.losses.append(float(loss.item()))
                rec.coefs.append((a, b, c, d))
                rec.step += 1
                return loss, z

            def forward(self, x, *rest):
                if x.shape[0] == rec.N:
                    torch.manual_seed(eval_seed(rec.seed))
                return super().forward(x, *rest)
        return Rec

    class RecPlateau(torch.optim.lr_scheduler.ReduceLROnPlateau):
        def __init__(self, optimizer, *a, verbose=None, **kw):
            super().__init__(optimizer, *a, **kw)

        def step(self, metrics, *a, **kw):
            super().step(metrics, *a, **kw)
            rec.lrs.append(self.optimizer.param_groups[0]["lr"])

    saved = (R.SyntheticDataset, R.cleanIVAE, R.cleanVAE, torch.optim.lr_scheduler.ReduceLROnPlateau)
    R.SyntheticDataset, R.cleanIVAE, R.cleanVAE = RecDataset, recording(R.cleanIVAE), recording(R.cleanVAE)
    torch.optim.lr_scheduler.ReduceLROnPlateau = RecPlateau
    out, index = {}, []
    cases = RUNNER_CASES
    if only:            # add the named cases to the existing file, leave its other runs as they are
        old = np.load(os.path.join(OUT, "trace_runner.npz"))
        out = {k: old[k] for k in old.files if k != "index"}
        index = [r for r in json.loads(str(old["index"])) if r["name"] not in only]
        out = {k: v for k, v in out.items() if k.split(".")[0] not in only}
        cases = [c for c in RUNNER_CASES if c[0] in only]
    try:
        with tempfile.TemporaryDirectory() as tmp:
            for name, yml, over, seeds, epochs in cases:
                with open(os.path.join(REF, "configs", yml)) as f:
                    cfg = yaml.safe_load(f)
                cfg.update(over)
                cfg["epochs"] = epochs
                for seed in seeds:
                    config = argparse.Namespace(**cfg)
                    config.device = torch.device("cpu")
                    args = argparse.Namespace(data_path=tmp + "/", seed=seed, n_sims=1)

                    def run(perturb):
                        rec.seed, rec.step, rec.N, rec.perturb = seed, 0, cfg["nps"] * cfg["ns"], perturb
                        rec.order, rec.losses, rec.coefs, rec.lrs = [], [], [], []
                        np.random.seed(seed)
                        torch.manual_seed(seed)
                        return R.runner(args, config)

                    key = f"{name}.s{seed}"
                    if True:
                        # How far does the REFERENCE move when its initial parameters move by 1e-6 relative (same batch
                        # order: the shuffle is seeded by the per-step noise seeds; same noise)?  Hundreds of Adam steps
                        # amplify round-off; the replay test holds ivae_amd to a multiple of this measured spread.
                        sens = [run(k) for k in (1, 2, 3, 4)]
                        out[key + ".sens_loss_hist"] = np.array([r_[1][0] for r_ in sens], np.float64)
                        out[key + ".sens_perf_hist"] = np.array([r_[2][0] for r_ in sens], np.float64)
                        out[key + ".sens_perf"] = np.array([r_[0][0] for r_ in sens], np.float64)
                    perfs, loss_hists, perf_hists = run(0)
                    if True:
                        print(f"  sensitivity {key}: perf_hist {out[key + '.sens_perf_hist'].tolist()} perf {out[key + '.sens_perf'].tolist()}")
                    B = cfg["batch_size"]
                    nb = rec.N // B
                    assert len(rec.order) == epochs * nb * B and len(rec.losses) == epochs * nb
                    out[key + ".order"] = np.array(rec.order, np.uint16 if rec.N <= 65536 else np.int32)
                    out[key + ".step_loss"] = np.array(rec.losses, np.float32)
                    out[key + ".step_coef"] = np.array(rec.coefs[::nb], np.float32)       # one row per epoch
                    out[key + ".lr"] = np.array(rec.lrs, np.float64)
                    out[key + ".loss_hist"] = np.array(loss_hists[0], np.float64)
                    out[key + ".perf_hist"] = np.array(perf_hists[0], np.float64)
                    out[key + ".perf"] = np.array(perfs[0], np.float64)
                    index.append(dict(key=key, name=name, seed=seed, epochs=epochs, config=cfg))
                    print(f"runner trace {key}: loss {loss_hists[0]} perf {perf_hists[0]} lr {rec.lrs} full perf {perfs[0]:.4f}")
    finally:
        R.SyntheticDataset, R.cleanIVAE, R.cleanVAE, torch.optim.lr_scheduler.ReduceLROnPlateau = saved
    out["index"] = np.array(json.dumps(index))
    np.savez_compressed(os.path.join(OUT, "trace_runner.npz"), **out)


# ---- full-length family-A runs (configs/ivae.yaml:29, btcvae.yaml: 20 epochs = 12 500 steps) -------------------------
FULL_CASES = {
    # name: yaml file of the reference, config overrides, epochs
    "ivae": ("ivae.yaml", {}, 20),
    "btcvae": ("btcvae.yaml", {}, 20),
}
FULL_SEEDS = (0, 1, 2, 3)
FULL_TWINS = (1, 2)


def order_seed(seed):
    """Seed of the generator the DataLoader of a full-length run shuffles with (a shim passes it: the row order of every
    epoch then follows from this number, and the fixture need not hold 20 x 40 000 indices per run)."""
    return 5_000_000 + seed


def epoch_orders(seed, N, B, epochs):
    """Row order of each epoch of a full-length run: what torch's DataLoader(shuffle=True, drop_last=True,
    generator=manual_seed(order_seed(seed))) asks its dataset for, epoch after epoch (tests/ rebuild it the same way)."""
    from torch.utils.data import DataLoader

    class Rows(torch.utils.data.Dataset):
        def __len__(self):
            return N

        def __getitem__(self, i):
            return i

    dl = DataLoader(Rows(), batch_size=B, shuffle=True, drop_last=True, generator=torch.Generator().manual_seed(order_seed(seed)))
    return [torch.cat([b for b in dl]) for _ in range(epochs)]


def order_digest(rows):
    """A few numbers that pin an epoch's row order (first rows and a position-weighted sum)."""
    r = np.asarray(rows, np.int64)
    return np.array([r[0], r[1], r[2], r[-1], int((r * (np.arange(r.size) % 1009 + 1)).sum() % (2 ** 53))], np.int64)


def make_full_trace(name, seed):
    """One full-length run of the reference's `runner` (runners/ivae.py:13-128, all 20 epochs of the yaml config) plus
    FULL_TWINS runs of the same with the initial parameters moved by 1e-6 relative -> tests/golden/_full/<name>.s<seed>.npz.
    Same recording shims as make_runner_traces, plus one: the DataLoader is given generator=manual_seed(order_seed(seed))."""
    import argparse
    import tempfile
    import yaml
    from torch.utils.data import DataLoader
    torch.set_num_threads(1)
    if not hasattr(np, "int"):
        np.int = int
    import runners.ivae as R
    yml, over, epochs = FULL_CASES[name]
    rec = argparse.Namespace()

    class RecDataset(R.SyntheticDataset):
        def __getitem__(self, index):
            rec.order.append(int(index))
            return super().__getitem__(index)

    def recording(base):
        class Rec(base):
            def __init__(self, *a, **kw):
                super().__init__(*a, **kw)
                if rec.perturb:
                    gen = torch.Generator().manual_seed(4242 + rec.perturb)
                    with torch.no_grad():
                        for p in self.parameters():
                            p.mul_(1 + 1e-6 * torch.randn(p.shape, generator=gen))

            def elbo(self, x, u, N, a=1., b=1., c=1., d=1.):
                torch.manual_seed(noise_seed(rec.seed, rec.step))
                loss, z = super().elbo(x, u, N, a=a, b=b, c=c, d=d)
                rec.losses.append(float(loss.item()))
                rec.coefs.append((a, b, c, d))
                rec.step += 1
                return loss, z

            def forward(self, x, *rest):
                if x.shape[0] == rec.N:
                    torch.manual_seed(eval_seed(rec.seed))
                return super().forward(x, *rest)
        return Rec

    class RecPlateau(torch.optim.lr_scheduler.ReduceLROnPlateau):
        def __init__(self, optimizer, *a, verbose=None, **kw):
            super().__init__(optimizer, *a, **kw)

        def step(self, metrics, *a, **kw):
            super().step(metrics, *a, **kw)
            rec.lrs.append(self.optimizer.param_groups[0]["lr"])

    def seeded_loader(dset, **kw):
        return DataLoader(dset, generator=torch.Generator().manual_seed(order_seed(rec.seed)), **kw)

    R.SyntheticDataset, R.cleanIVAE, R.cleanVAE, R.DataLoader = RecDataset, recording(R.cleanIVAE), recording(R.cleanVAE), seeded_loader
    torch.optim.lr_scheduler.ReduceLROnPlateau = RecPlateau
    with open(os.path.join(REF, "configs", yml)) as f:
        cfg = yaml.safe_load(f)
    cfg.update(over)
    cfg["epochs"] = epochs
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        config = argparse.Namespace(**cfg)
        config.device = torch.device("cpu")
        args = argparse.Namespace(data_path=tmp + "/", seed=seed, n_sims=1)

        def run(perturb):
            rec.seed, rec.step, rec.N, rec.perturb = seed, 0, cfg["nps"] * cfg["ns"], perturb
            rec.order, rec.losses, rec.coefs, rec.lrs = [], [], [], []
            np.random.seed(seed)
            torch.manual_seed(seed)
            return R.runner(args, config)

        key = f"{name}.s{seed}"
        perfs, loss_hists, perf_hists = run(0)
        B = cfg["batch_size"]
        nb = rec.N // B
        assert len(rec.order) == epochs * nb * B and len(rec.losses) == epochs * nb
        rebuilt = torch.cat(epoch_orders(seed, rec.N, B, epochs)).numpy()
        assert np.array_equal(rebuilt, np.array(rec.order)), "epoch_orders does not reproduce the DataLoader's order"
        out[key + ".order_digest"] = np.stack([order_digest(rec.order[e * nb * B:(e + 1) * nb * B]) for e in range(epochs)])
        out[key + ".first_epoch_step_loss"] = np.array(rec.losses[:nb], np.float32)
        out[key + ".step_coef"] = np.array(rec.coefs[::nb], np.float32)
        out[key + ".lr"] = np.array(rec.lrs, np.float64)
        out[key + ".loss_hist"] = np.array(loss_hists[0], np.float64)
        out[key + ".perf_hist"] = np.array(perf_hists[0], np.float64)
        out[key + ".perf"] = np.array(perfs[0], np.float64)
        print(f"full trace {key}: loss {loss_hists[0][0]:.3f} -> {loss_hists[0][-1]:.3f} perf_hist {perf_hists[0][-1]:.4f} "
              f"lr {rec.lrs} full perf {perfs[0]:.4f}", flush=True)
        sens = []
        for k in FULL_TWINS:
            r_ = run(k)
            sens.append((r_, list(rec.lrs)))
            print(f"  twin {k} of {key}: perf {r_[0][0]:.4f} lr {rec.lrs}", flush=True)
        out[key + ".sens_loss_hist"] = np.array([r_[1][0] for r_, _ in sens], np.float64)
        out[key + ".sens_perf_hist"] = np.array([r_[2][0] for r_, _ in sens], np.float64)
        out[key + ".sens_perf"] = np.array([r_[0][0] for r_, _ in sens], np.float64)
        out[key + ".sens_lr"] = np.array([l for _, l in sens], np.float64)
        out[key + ".config"] = np.array(json.dumps(dict(key=key, name=name, seed=seed, epochs=epochs, config=cfg)))
    os.makedirs(os.path.join(OUT, "_full"), exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "_full", key + ".npz"), **out)


def merge_full_traces():
    """tests/golden/_full/*.npz -> tests/golden/trace_full.npz (one index entry per run)."""
    out, index = {}, []
    d = os.path.join(OUT, "_full")
    for fn in sorted(os.listdir(d)):
        g = np.load(os.path.join(d, fn))
        for k in g.files:
            if k.endswith(".config"):
                index.append(json.loads(str(g[k])))
            else:
                out[k] = g[k]
    out["index"] = np.array(json.dumps(index))
    np.savez_compressed(os.path.join(OUT, "trace_full.npz"), **out)
    print("merged", len(index), "runs")


# ---- family-B loop traces: the reference's own models/wrappers.py, run unmodified ---------------------------------------
WRAPPER_CASES = [
    # key, wrapper, data kwargs (ivae_amd.data.generate_data == data/data.py, bit for bit), wrapper kwargs
    # (the reference prints logger.get_last() at a checkpoint, which raises before the first epoch has been logged:
    #  every case keeps its first checkpoint, at int(max_iter / 5), beyond the first epoch)
    ("ivae_fused", "IVAE", dict(n_per_seg=25, n_seg=20, d_sources=10, n_layers=3, seed=3, activation="lrelu"),
     dict(batch_size=64, max_iter=50, seed=5, hidden_dim=50, lr=1e-2)),                 # the fused kernel's shape; 7 full batches + 52 rows
    ("ivae_anneal", "IVAE", dict(n_per_seg=25, n_seg=20, d_sources=10, n_layers=3, seed=3, activation="lrelu"),
     dict(batch_size=64, max_iter=50, seed=6, hidden_dim=50, lr=1e-2, anneal=True, scheduler_tol=0)),    # weights change every iteration
    # (the switch at iteration 31 raises the loss of that epoch: with patience 0 the rate drops there)
    ("ivae_generic", "IVAE", dict(n_per_seg=30, n_seg=12, d_sources=4, n_layers=2, seed=4, activation="xtanh"),
     dict(batch_size=100, max_iter=27, seed=7, hidden_dim=30, n_layers=2, lr=1e-2, activation="xtanh", scheduler_tol=0)),
    ("vae_fused", "VAE", dict(n_per_seg=25, n_seg=20, d_sources=10, n_layers=3, seed=3, activation="lrelu"),
     dict(batch_size=64, max_iter=45, seed=8, hidden_dim=50, lr=1e-2)),
]


def wrapper_noise_seed(seed, it):
    return 2_000_000 * (seed + 1) + it


def make_wrapper_traces():
    """Run the reference's `IVAE_wrapper` / `VAE_wrapper` (models/wrappers.py:14-176, cuda=False) with recording shims only:
    the dataset logs the row order its DataLoader(shuffle=True) asks for (no drop_last: the last batch is ragged), `elbo`
    seeds the global generator per iteration so that its randn draw can be replayed and logs the loss, the full-data
    forward at the end is seeded, ReduceLROnPlateau drops `verbose` and logs the rate after every step().  Logger and
    checkpoints go to a scratch folder; the iterations that got a checkpoint and the Logger's per-epoch means are recorded.
    -> tests/golden/trace_wrapper.npz, replayed through ivae_amd.runners by tests/test_gpu_wrapper_trace.py."""
    import argparse
    import glob
    import tempfile
    if not hasattr(np, "int"):
        np.int = int
    import models.wrappers as W
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from ivae_amd.data import generate_data

    rec = argparse.Namespace()

    class RecDataset(W.CustomSyntheticDataset):
        def __getitem__(self, index):
            rec.order.append(int(index))
            return super().__getitem__(index)

    def recording(base):
        class Rec(base):
            def elbo(self, *a):
                rec.it += 1
                if rec.it == 1:
                    rec.init = {k: v.detach().clone().numpy() for k, v in self.state_dict().items()}
                torch.manual_seed(wrapper_noise_seed(rec.seed, rec.it))
                if getattr(self, "anneal_params", False):
                    rec.coefs.append(tuple(float(t) for t in self._training_hyperparams) + (1.,))
                else:
                    rec.coefs.append((1., 1., 1., 1., 1., 0.))
                e, z = super().elbo(*a)
                rec.losses.append(-float(e.item()))
                return e, z

            def forward(self, x, *rest):
                if x.shape[0] == rec.N:
                    torch.manual_seed(eval_seed(rec.seed))
                return super().forward(x, *rest)
        return Rec

    class RecPlateau(torch.optim.lr_scheduler.ReduceLROnPlateau):
        def __init__(self, optimizer, *a, verbose=None, **kw):
            super().__init__(optimizer, *a, **kw)

        def step(self, metrics, *a, **kw):
            super().step(metrics, *a, **kw)
            rec.lrs.append(self.optimizer.param_groups[0]["lr"])

    saved = (W.CustomSyntheticDataset, W.iVAE, W.VAE, W.optim.lr_scheduler.ReduceLROnPlateau)
    W.CustomSyntheticDataset, W.iVAE, W.VAE = RecDataset, recording(W.iVAE), recording(W.VAE)
    W.optim.lr_scheduler.ReduceLROnPlateau = RecPlateau
    out, index = {}, []
    try:
        for key, which, dkw, wkw in WRAPPER_CASES:
            S, X, U, _, _ = generate_data(batch_size=0, repeat_linearity=True, **dkw)
            with tempfile.TemporaryDirectory() as tmp:
                rec.seed, rec.it, rec.N = wkw["seed"], 0, X.shape[0]
                rec.order, rec.losses, rec.coefs, rec.lrs = [], [], [], []
                kw = dict(wkw, cuda=False, log_folder=tmp + "/log/", ckpt_folder=tmp + "/ckpt/")
                os.makedirs(kw["log_folder"]); os.makedirs(kw["ckpt_folder"])
                if which == "IVAE":
                    z, model, params, logger = W.IVAE_wrapper(X, U, S, **kw)
                else:
                    z, model, params, logger = W.VAE_wrapper(X, S, **kw)
                ck = sorted(int(os.path.basename(f).split("_ckpt_")[1].split(".")[0]) for f in glob.glob(tmp + "/**/*_ckpt_*.pth", recursive=True))
                out[key + ".order"] = np.array(rec.order, np.int32)
                out[key + ".step_loss"] = np.array(rec.losses, np.float32)
                out[key + ".step_coef"] = np.array(rec.coefs, np.float64)
                out[key + ".lr"] = np.array(rec.lrs, np.float64)
                out[key + ".ckpt_its"] = np.array(ck, np.int64)
                out[key + ".log_elbo"] = np.array(logger.log_dict["elbo"], np.float64)
                out[key + ".log_perf"] = np.array(logger.log_dict["perf"], np.float64)
                out[key + ".z"] = z.detach().numpy()
                for k, v in rec.init.items():
                    out[key + ".init." + k] = v
                index.append(dict(key=key, wrapper=which, data=dkw, kwargs=wkw, n_iter=rec.it, N=int(X.shape[0])))
                print(f"wrapper trace {key}: {rec.it} iterations, loss {logger.log_dict['elbo']} perf {logger.log_dict['perf']} "
                      f"lr {rec.lrs} checkpoints at {ck}")
    finally:
        W.CustomSyntheticDataset, W.iVAE, W.VAE, W.optim.lr_scheduler.ReduceLROnPlateau = saved
    out["index"] = np.array(json.dumps(index))
    np.savez_compressed(os.path.join(OUT, "trace_wrapper.npz"), **out)


def make_random_cases(n=24):
    """More reference-made fixtures at random small shapes (round 3): every model class, 1-4 layers, every activation,
    one-hot and dense u, ragged batches, random loss weights for the decomposed ELBO, annealed iVAE at random iterations
    (both sides of the switch to the plain ELBO) -- the tests pick up every *.npz of tests/golden."""
    r = np.random.RandomState(2024)
    for i in range(n):
        family = ["iVAE", "VAE", "cleanIVAE", "cleanVAE"][i % 4]
        dl, dd = int(r.randint(2, 13)), int(r.randint(2, 13))
        M = int(r.randint(2, 21)) if family in ("iVAE", "cleanIVAE") else 0
        kw = dict(B=int(r.randint(3, 91)), H=int(r.randint(5, 41)), n_layers=int(r.randint(1, 5)),
                  act=str(r.choice(["lrelu", "xtanh", "sigmoid", "none"])), slope=float(r.choice([.1, .2])),
                  seed=100 + i, N=int(r.choice([4000, 40000, 1000000])), store_steps=False)
        if family in ("iVAE", "cleanIVAE"):
            kw["dense_u"] = bool(r.randint(2))
        if family.startswith("clean"):
            kw["coef"] = tuple(float(t) for t in np.round(r.uniform(.2, 20., 4), 3))
        elif family == "iVAE" and r.randint(2):
            kw["anneal_it"] = int(r.choice([0, 5, 300, 620, 700]))
        make_case("rnd_%02d_%s" % (i, family.lower()), family, (dl, dd, M), **kw)


def main():
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "wrapper_traces":
        return make_wrapper_traces()
    if len(sys.argv) > 1 and sys.argv[1] == "random_cases":
        return make_random_cases()
    if len(sys.argv) > 1 and sys.argv[1] == "runner_traces":          # runner_traces [case names to (re)make and merge]
        return make_runner_traces(tuple(sys.argv[2:]))
    if len(sys.argv) > 1 and sys.argv[1] == "datasets":
        return make_datasets()
    if len(sys.argv) > 1 and sys.argv[1] == "traces":
        return make_traces()
    if len(sys.argv) > 1 and sys.argv[1] == "trace_twins":
        return make_trace_twins()
    if len(sys.argv) > 1 and sys.argv[1] == "full_trace":          # full_trace <ivae|btcvae> <seed>
        return make_full_trace(sys.argv[2], int(sys.argv[3]))
    if len(sys.argv) > 1 and sys.argv[1] == "traces_long":
        return make_traces_long()
    if len(sys.argv) > 1 and sys.argv[1] == "full_merge":
        return merge_full_traces()
    # family B (models/nets.py:391-486, 548-599)
    make_case("ivae_plain_tiny", "iVAE", (3, 3, 4), B=5, H=8)
    make_case("ivae_plain_cfg2", "iVAE", (10, 10, 40), B=64, H=50)
    make_case("ivae_plain_b70", "iVAE", (4, 6, 7), B=70, H=20, dense_u=True)
    make_case("ivae_plain_l2_sigmoid", "iVAE", (3, 5, 4), B=9, H=12, n_layers=2, act="sigmoid")
    make_case("ivae_plain_l4_none", "iVAE", (3, 4, 5), B=11, H=7, n_layers=4, act="none")
    make_case("ivae_plain_l1", "iVAE", (3, 3, 4), B=6, H=5, n_layers=1)
    make_case("ivae_anneal_tiny", "iVAE", (3, 3, 4), B=5, H=8, anneal_it=300)
    make_case("ivae_anneal_cfg2", "iVAE", (10, 10, 40), B=64, H=50, anneal_it=1)
    make_case("ivae_anneal_switch", "iVAE", (3, 3, 4), B=8, H=8, anneal_it=623)  # crosses thr=625 -> plain
    make_case("vae_plain_tiny", "VAE", (3, 4, 0), B=7, H=8)
    make_case("vae_plain_cfg2", "VAE", (10, 10, 0), B=64, H=50)
    make_case("ivae_plain_cfg4s", "iVAE", (20, 20, 200), B=96, H=50, N=1000000, store_steps=False)
    # family A (models/nets.py:124-243), coefficients from configs/{ivae,bvae,btcvae,vae}.yaml:24-27
    make_case("cleanivae_tiny", "cleanIVAE", (3, 3, 4), B=5, H=8, act="xtanh", coef=(100, 1, 0, 10))
    make_case("cleanivae_cfg1", "cleanIVAE", (5, 5, 40), B=64, H=100, act="xtanh", coef=(100, 1, 0, 10),
              store_steps=False)
    make_case("cleanivae_lrelu_b70", "cleanIVAE", (4, 6, 7), B=70, H=20, act="lrelu", coef=(2, 3, 1.5, 0.5))
    make_case("cleanvae_tiny", "cleanVAE", (3, 4, 0), B=6, H=8, act="xtanh", coef=(500, 5, 5, 5))
    make_case("cleanvae_bvae", "cleanVAE", (5, 5, 0), B=64, H=100, act="xtanh", coef=(500, 5, 5, 5),
              store_steps=False)
    make_case("cleanvae_btcvae", "cleanVAE", (5, 5, 0), B=64, H=100, act="xtanh", coef=(1000, 1, 15, 1),
              store_steps=False)
    make_case("cleanvae_btcvae_b200", "cleanVAE", (5, 5, 0), B=200, H=100, act="xtanh", coef=(1000, 1, 15, 1),
              store_steps=False)
    make_mcc()


if __name__ == "__main__":
    main()
