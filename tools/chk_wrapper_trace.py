"""Prints how far the replays of tests/test_gpu_wrapper_trace.py land from the recorded reference runs (per-step loss over
the first epoch, epoch means of loss and batch MCC, final MCC): the margins the test's bars were chosen against.
usage (GPU box): python tools/chk_wrapper_trace.py"""
import contextlib
import io
import os
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ivae_amd.data import generate_data                         # noqa: E402
from ivae_amd.mcc import mean_corr_coef                         # noqa: E402
from ivae_amd.runners import IVAE_wrapper, VAE_wrapper          # noqa: E402
from tests.test_gpu_wrapper_trace import GOLD, Replay, _runs    # noqa: E402

g = np.load(GOLD)
for key, r in _runs():
    kw, N = dict(r["kwargs"]), r["N"]
    S, X, U, _, _ = generate_data(batch_size=0, repeat_linearity=True, **r["data"])
    nb = -(-N // kw["batch_size"])
    trace = {}
    with tempfile.TemporaryDirectory() as tmp, contextlib.redirect_stdout(io.StringIO()):
        hooks = dict(replay=Replay(g, key, kw["seed"], N, S.shape[1]), trace=trace, log_folder=tmp + "/l/", ckpt_folder=tmp + "/c/")
        os.makedirs(tmp + "/l"), os.makedirs(tmp + "/c")
        if r["wrapper"] == "IVAE":
            z, model, params, logger = IVAE_wrapper(X, U, S, **kw, **hooks)
        else:
            z, model, params, logger = VAE_wrapper(X, S, **kw, **hooks)
    ref, got = g[key + ".step_loss"], torch.cat(trace["step_loss"]).numpy()
    rel = np.abs(got - ref) / np.abs(ref)
    elbo, perf = np.array(logger.log_dict["elbo"]), np.array(logger.log_dict["perf"])
    print(f"{key:13s} step 1 {rel[0]:.1e}  epoch 1 max {rel[:nb].max():.1e}  all steps max {rel.max():.1e} | "
          f"epoch loss {np.abs(elbo - g[key + '.log_elbo']).max() / np.abs(g[key + '.log_elbo']).max():.1e} of the largest  "
          f"epoch perf {np.abs(perf - g[key + '.log_perf']).max():.4f} | final MCC {mean_corr_coef(z.cpu().numpy(), S):.4f} "
          f"vs {mean_corr_coef(g[key + '.z'], S):.4f}", flush=True)
