"""Family-A loop parity: `ivae_amd.runners.runner` against runs of the REFERENCE's own
`runners/ivae.py::runner` recorded by tools/make_golden.py (`runner_traces`: ivae.yaml and btcvae.yaml, 2 epochs, seeds
0-1; ivae.yaml with anneal on, 3 epochs, which crosses the weight switch of runners/ivae.py:51-60 and makes
ReduceLROnPlateau(factor .1, patience 0) cut the learning rate twice; vae.yaml and bvae.yaml, 2 epochs, seed 0 -- every
yaml file the reference ships for this loop is replayed).

The replay feeds the recorded batch order (DataLoader(shuffle=True, drop_last=True), :25) and re-draws the recorded
noise from its seeds (CPU generator, identical on every machine), so everything else is the loop under test: model
initialisation from torch.manual_seed(seed), drop_last, loss weights per epoch, the scheduler on the mean epoch loss, the
per-step batch MCC averaged per epoch (:86,106), and the final score on a SAMPLE of q(z|x,u) (:118-123)."""
import argparse
import json
import os

import numpy as np
import pytest
import torch

from ivae_amd.runners import runner

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "trace_runner.npz")


class Replay:
    def __init__(self, g, key, N, B, d):
        self.rows = torch.from_numpy(g[key + ".order"].astype(np.int64))
        self.N, self.B, self.d = N, B, d
        self.nb = N // B

    def order(self, seed, epoch):
        n = self.nb * self.B
        return self.rows[(epoch - 1) * n:epoch * n]

    def eps(self, seed, epoch):
        out = torch.empty(self.nb, self.B, self.d)
        for i in range(self.nb):
            torch.manual_seed(1_000_000 * (seed + 1) + (epoch - 1) * self.nb + i)      # tools/make_golden.py::noise_seed
            out[i] = torch.randn(self.B, self.d)
        return out

    def eval_eps(self, seed):
        torch.manual_seed(777 + seed)                                                   # tools/make_golden.py::eval_seed
        return torch.randn(self.N, self.d)


def _runs():
    g = np.load(GOLD)
    return [(r["key"], r) for r in json.loads(str(g["index"]))]


@pytest.mark.parametrize("key", [k for k, _ in _runs()])
def test_runner_replays_reference_run(key, tmp_path):
    g = np.load(GOLD)
    r = dict(_runs())[key]
    cfg, seed, epochs = r["config"], r["seed"], r["epochs"]
    config = argparse.Namespace(**cfg)
    config.device = torch.device("cuda:0")
    args = argparse.Namespace(data_path=str(tmp_path) + "/", seed=seed, n_sims=1)
    N, B, d = cfg["nps"] * cfg["ns"], cfg["batch_size"], cfg["dl"]
    nb = N // B
    trace = {}
    np.random.seed(seed)                      # main.py:56-57
    torch.manual_seed(seed)
    perfs, loss_hists, perf_hists = runner(args, config, replay=Replay(g, key, N, B, d), trace=trace)

    ref_steps = g[key + ".step_loss"]
    got_steps = torch.cat(trace["step_loss"])[:, 0].numpy()
    assert got_steps.shape == ref_steps.shape == (epochs * nb,)                        # drop_last: N // B steps per epoch
    # same initial parameters (seeded init), same batch, same noise: the first loss to the north_star bar
    assert abs(got_steps[0] - ref_steps[0]) <= 1e-5 * abs(ref_steps[0]), (got_steps[0], ref_steps[0])
    scale = np.abs(ref_steps[:20]).max()
    assert np.abs(got_steps[:20] - ref_steps[:20]).max() <= 1e-3 * scale
    # loss weights per epoch: the anneal switch (runners/ivae.py:51-60)
    assert np.allclose(np.array(trace["coef"], np.float32), g[key + ".step_coef"])
    # ReduceLROnPlateau(factor .1, patience 0) on the mean epoch loss (:39,113-114): the same cuts at the same epochs
    assert np.allclose([l[0] for l in trace["lr"]], g[key + ".lr"], rtol=1e-12), (trace["lr"], g[key + ".lr"])
    # Epoch means of loss and of the per-step batch MCC, and the final full-data MCC of a sampled z.  The first losses
    # above agree to rounding; from there Adam amplifies fp32 round-off (its first updates are +-lr whatever the size of
    # a gradient, so an element whose gradient sits at the rounding floor may move the other way): the per-step loss
    # difference grows from 1e-7 to 1e-2 within 50 steps.  The fixture therefore holds, for every run, four runs of the
    # REFERENCE ITSELF with its initial parameters moved by 1e-6 relative (same batches, same noise): after two epochs
    # they differ from the recorded run by up to 0.1 in MCC.  The bar is the reference's own envelope: every statistic
    # must lie within the range spanned by the recorded run and its four perturbed twins, widened by that range once
    # more (+ a floor).  north_star's "MCC +-0.01 over fixed seeds" is a statement about seed averages: test below.
    base = key

    def within(got, ref, twins, floor, what):
        lo, hi = min(ref, *twins), max(ref, *twins)
        m = (hi - lo) + floor
        assert lo - m <= got <= hi + m, (what, got, ref, list(twins))

    ref_loss, ref_perf = g[key + ".loss_hist"], g[key + ".perf_hist"]
    tw_loss, tw_perf, tw_final = g[base + ".sens_loss_hist"], g[base + ".sens_perf_hist"], g[base + ".sens_perf"]
    for e in range(epochs):
        within(loss_hists[0][e], ref_loss[e], tw_loss[:, e], 0.002 * abs(ref_loss[e]) + 0.1, ("loss", e))
        within(perf_hists[0][e], ref_perf[e], tw_perf[:, e], 0.01, ("perf", e))
    ref_final = float(g[key + ".perf"])
    within(perfs[0], ref_final, tw_final, 0.01, "final perf")
    _DIFFS[key] = (perfs[0], float(np.mean(np.append(tw_final, ref_final))), float(np.std(np.append(tw_final, ref_final))))


_DIFFS = {}


def test_final_mcc_mean_over_runs_is_inside_the_reference_spread():
    """north_star's "final MCC within +-0.01 over fixed seeds", in the only form a chaotic trajectory admits: the mean
    over the five replayed runs of (our final MCC - the mean of the reference's five trajectories of that run) must be
    within two standard errors of those trajectories' own scatter, plus the 0.01."""
    if len(_DIFFS) < len(_runs()):
        pytest.skip("needs every replay of this file in the same session")
    diff = np.mean([ours - ref_mean for ours, ref_mean, _ in _DIFFS.values()])
    sem = np.sqrt(np.mean([sd ** 2 for _, _, sd in _DIFFS.values()]) / len(_DIFFS))
    assert abs(diff) <= 2 * sem + 0.01, (diff, sem, _DIFFS)


def test_packed_seeds_equal_separate_runs(tmp_path):
    """`--n-sims 2` trains both seeds as one packed set: with the recorded draws of ivae.s0 and ivae.s1 replayed, each
    packed model must do exactly what it does alone -- the same per-step losses bit for bit (same kernels, same
    arithmetic, no coupling between packed models) -- and keep its own scheduler."""
    g = np.load(GOLD)
    r = dict(_runs())["ivae.s0"]
    cfg = dict(r["config"], epochs=1)
    N, B, d = cfg["nps"] * cfg["ns"], cfg["batch_size"], cfg["dl"]

    class Both:
        def __init__(self):
            self.r = {0: Replay(g, "ivae.s0", N, B, d), 1: Replay(g, "ivae.s1", N, B, d)}

        def order(self, seed, epoch):
            return self.r[seed].order(seed, epoch)

        def eps(self, seed, epoch):
            return self.r[seed].eps(seed, epoch)

        def eval_eps(self, seed):
            return self.r[seed].eval_eps(seed)

    def run(seed, n_sims, init_seed):
        config = argparse.Namespace(**cfg)
        config.device = torch.device("cuda:0")
        args = argparse.Namespace(data_path=str(tmp_path) + "/", seed=seed, n_sims=n_sims)
        trace = {}
        torch.manual_seed(init_seed)
        out = runner(args, config, replay=Both(), trace=trace)
        return out, torch.cat(trace["step_loss"])

    (perfs, lh, ph), steps = run(0, 2, 0)
    assert len(perfs) == 2 and steps.shape == (N // B, 2)
    (p0, l0, _), s0 = run(0, 1, 0)                   # seed 0 alone: same initial weights as the packed run's first model
    assert torch.equal(steps[:, 0], s0[:, 0]) and lh[0] == l0[0] and perfs[0] == p0[0]
    assert not torch.equal(steps[:, 0], steps[:, 1])
