"""FusedAdam <-> torch.optim.Adam state_dict round trips: the reference checkpoints
{'model_state_dict', 'optimizer_state_dict', ...} with torch.save (utils/utils.py:55-64) and has no resume code of its
own, so the contract is torch's: a state_dict saved by either optimiser loads into the other and training continues
as if nothing had happened.  Checked by continuing the reference's 4-step fixture trajectory (3 steps at lr 1e-2, an lr
drop to 1e-3, 1 step) across a save / load in the middle."""
import numpy as np
import pytest
import torch

import ivae_amd
from ivae_amd.runlog import checkpoint
from tests.test_api_cpu import _build
from tests.util import Case

pytestmark = pytest.mark.gpu


def _step(model, opt, c, k, d):
    x, u, eps = (torch.from_numpy(t).to(d) for t in c.inputs(k))
    opt.zero_grad()
    elbo, _ = model.elbo(x, u, eps=eps)
    elbo.mul(-1).backward()
    opt.step()


def _check(model, c, k):
    for n, p in model.named_parameters():
        err = np.abs(p.detach().cpu().numpy() - c.z[f"adam{k}.p.{n}"])
        lr = c.meta["lrs"][k - 1]
        assert err.max() < 2e-2 * lr and err.mean() < 1e-2 * lr, (k, n, err.max(), err.mean())   # test_module_training_trajectory's bar


@pytest.mark.parametrize("first,second", [("torch", "fused"), ("fused", "torch"), ("fused", "fused")])
def test_checkpoint_round_trip_continues_reference_trajectory(first, second, tmp_path):
    c = Case("ivae_plain_cfg2")
    d = torch.device("cuda:0")
    mk = {"torch": torch.optim.Adam, "fused": ivae_amd.FusedAdam}
    model = _build(c)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in c.params.items()})
    model = model.to(d)
    opt = mk[first](model.parameters(), lr=1e-2)
    _step(model, opt, c, 0, d)
    _step(model, opt, c, 1, d)
    _check(model, c, 2)
    path = checkpoint(str(tmp_path) + "/", 7, 2, model, opt, 0., 0.)           # the reference's .pth layout
    ck = torch.load(path, map_location="cpu", weights_only=True)               # state arrives as separate CPU tensors
    assert set(ck) == {"iteration", "model_state_dict", "optimizer_state_dict", "loss", "perf"}
    assert int(ck["optimizer_state_dict"]["state"][0]["step"]) == 2

    model2 = _build(c).to(d)
    model2.load_state_dict(ck["model_state_dict"])
    opt2 = mk[second](model2.parameters(), lr=1.0)                             # lr comes from the checkpoint
    opt2.load_state_dict(ck["optimizer_state_dict"])
    assert opt2.param_groups[0]["lr"] == 1e-2
    _step(model2, opt2, c, 2, d)
    _check(model2, c, 3)
    for g in opt2.param_groups:
        g["lr"] = 1e-3
    _step(model2, opt2, c, 3, d)
    _check(model2, c, 4)
    sd = opt2.state_dict()
    assert int(sd["state"][0]["step"]) == 4
    # the moments too: exp_avg / exp_avg_sq after 4 steps against the fixture
    for i, (n, _) in enumerate(model2.named_parameters()):
        m = sd["state"][i]["exp_avg"].cpu().numpy()
        v = sd["state"][i]["exp_avg_sq"].cpu().numpy()
        assert np.abs(m - c.z[f"adam4.m.{n}"]).max() <= 2e-3 * np.abs(c.z[f"adam4.m.{n}"]).max() + 1e-7, n
        assert np.abs(v - c.z[f"adam4.v.{n}"]).max() <= 2e-3 * np.abs(c.z[f"adam4.v.{n}"]).max() + 1e-9, n


def test_load_into_an_optimizer_that_already_stepped_replaces_its_cached_state():
    """A FusedAdam that has run the direct path (flat moment buffers cached) and then loads a state_dict must continue
    from the LOADED moments and step count, bit for bit like a fresh optimiser given the same state."""
    c = Case("ivae_plain_cfg2")
    d = torch.device("cuda:0")
    x, u, eps = (torch.from_numpy(t).to(d) for t in c.inputs(0))

    def fresh():
        m = _build(c)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in c.params.items()})
        return m.to(d)

    src_model = fresh()
    src = ivae_amd.FusedAdam(src_model.parameters(), lr=1e-2)
    for _ in range(3):
        src_model.direct_step(src, x, u, 1., 1., 1., 1., 1., False, eps=eps)
    state = src.state_dict()
    params = {k: v.clone() for k, v in src_model.state_dict().items()}

    used_model = fresh()
    used = ivae_amd.FusedAdam(used_model.parameters(), lr=1e-2)
    for _ in range(5):                                    # different history: caches hold 5 steps of other moments
        used_model.direct_step(used, x * 0.5, u, 1., 1., 1., 1., 1., False, eps=eps)
    used_model.load_state_dict(params)
    used.load_state_dict(state)

    new_model = fresh()
    new_model.load_state_dict(params)
    new = ivae_amd.FusedAdam(new_model.parameters(), lr=1e-2)
    new.load_state_dict(state)

    for model, opt in ((used_model, used), (new_model, new)):
        model.direct_step(opt, x, u, 1., 1., 1., 1., 1., False, eps=eps)
        model.direct_steps(opt, x[None], u[None], 1., 1., 1., 1., 1., False, eps=eps[None])
    assert int(used.state_dict()["state"][0]["step"]) == int(new.state_dict()["state"][0]["step"]) == 5
    for a, b in zip(used_model.parameters(), new_model.parameters()):
        assert torch.equal(a, b)
    for i in range(len(list(new_model.parameters()))):
        assert torch.equal(used.state_dict()["state"][i]["exp_avg_sq"], new.state_dict()["state"][i]["exp_avg_sq"])
