#!/usr/bin/env python3
"""One-off, development container only (imports the reference; listed in .gpurunignore): host-side logic of the drop-in
modules against the reference's, on CPU -- (1) iVAE.anneal's loss weights over random (N, max_iter, it); (2) parameter
initialisation of all four model classes from the same torch seed over random constructor arguments (bit-exact,
state_dict keys and shapes); (3) PlateauLR against torch's ReduceLROnPlateau driven by the reference's settings."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(1, "/root/reference")
from models import nets as R           # noqa: E402  (the reference)
from ivae_amd import nets as O         # noqa: E402

bad = 0
r = np.random.RandomState(0)
# (1) anneal
for case in range(300):
    N = int(r.choice([1000, 40000, 1000000])); max_iter = int(r.choice([10, 100, 12500, 70000]))
    it = int(r.randint(0, max_iter + 5))
    a, b = R.iVAE(3, 4, 5, anneal=True), O.iVAE(3, 4, 5, anneal=True)
    a.anneal(N, max_iter, it); b.anneal(N, max_iter, it)
    if a.anneal_params != b.anneal_params or list(a._training_hyperparams) != list(b._training_hyperparams):
        bad += 1
        print("anneal differs", N, max_iter, it, a.anneal_params, b.anneal_params, a._training_hyperparams, b._training_hyperparams)
# (2) initialisation
for case in range(60):
    dl, dd, M = int(r.randint(1, 12)), int(r.randint(1, 12)), int(r.randint(1, 45))
    H, L = int(r.choice([1, 7, 50, 100])), int(r.choice([1, 2, 3, 4]))
    act = str(r.choice(["lrelu", "xtanh", "sigmoid", "none"]))
    seed = int(r.randint(0, 10000))
    pairs = [(lambda: R.iVAE(dl, dd, M, n_layers=L, hidden_dim=H, activation=act), lambda: O.iVAE(dl, dd, M, n_layers=L, hidden_dim=H, activation=act)),
             (lambda: R.cleanIVAE(dd, dl, M, n_layers=L, activation=act, hidden_dim=H), lambda: O.cleanIVAE(dd, dl, M, n_layers=L, activation=act, hidden_dim=H)),
             (lambda: R.cleanVAE(dd, dl, n_layers=L, activation=act, hidden_dim=H), lambda: O.cleanVAE(dd, dl, n_layers=L, activation=act, hidden_dim=H)),
             (lambda: R.VAE(dl, dd, n_layers=L, hidden_dim=H, activation=act), lambda: O.VAE(dl, dd, n_layers=L, hidden_dim=H, activation=act))]
    for mk_r, mk_o in pairs:
        torch.manual_seed(seed); mr = mk_r(); st_r = torch.get_rng_state()
        torch.manual_seed(seed); mo = mk_o(); st_o = torch.get_rng_state()
        sr, so = mr.state_dict(), mo.state_dict()
        if list(sr.keys()) != list(so.keys()) or any(sr[k].shape != so[k].shape or not torch.equal(sr[k], so[k]) for k in sr):
            bad += 1
            print("init differs", type(mr).__name__, dict(dl=dl, dd=dd, M=M, H=H, L=L, act=act, seed=seed))
        # the generator must be left in the same state as well (what is drawn next, e.g. the first eps, is then equal)
        if not torch.equal(st_r, st_o):
            bad += 1
            print("generator state after construction differs", type(mr).__name__, seed)
print("differences:", bad)
