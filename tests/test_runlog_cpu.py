"""Host-side run records (ivae_amd/runlog.py): running means, experiment ids, file layouts of the reference's
utils/utils.py:55-64,145-225."""
import json
import multiprocessing as mp

import numpy as np
import torch

from ivae_amd.runlog import Logger, checkpoint, next_experiment_id


def _take_ids(folder, n, q):
    q.put([next_experiment_id(folder) for _ in range(n)])


def test_running_means_and_exports(tmp_path):
    log = Logger(log_dir=str(tmp_path), run='x')
    assert log.exp_id == 1
    log.add('elbo'); log.add('perf')
    for v in (1., 2., torch.tensor(3.)):
        log.update('elbo', v)
    log.update('perf', torch.tensor(8.), weight=4)          # a sum of four observations
    log.log()
    log.update('elbo', 10.)
    log.log()
    assert log.log_dict['elbo'] == [2., 10.] and log.log_dict['perf'] == [2., 0.]
    assert log.get_last('elbo') == 10. and len(log) == 2 and set(log.keys()) == {'elbo', 'perf'}
    log.add_metadata(note='n')
    log.save_to_json(method='full')
    log.save_to_json()
    recs = [json.loads(l) for l in open(tmp_path / 'log.json')]
    assert recs[0]['elbo'] == [2., 10.] and recs[1]['elbo'] == 10. and recs[1]['metadata'] == {'run': 'x', 'note': 'n'}
    assert recs[0]['id'] == 2 and recs[1]['id'] == 3
    path = log.save_to_npz()
    assert path.endswith('4.npz') and np.load(path)['elbo'].tolist() == [2., 10.]
    assert Logger().exp_id is None and Logger().save_to_json() is None


def test_experiment_ids_are_unique_across_processes(tmp_path):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_take_ids, args=(str(tmp_path), 20, q)) for _ in range(3)]
    for p in procs:
        p.start()
    ids = sum((q.get() for _ in procs), [])
    for p in procs:
        p.join(30)
    assert sorted(ids) == list(range(1, 61))


def test_checkpoint_layout(tmp_path):
    model = torch.nn.Linear(3, 2)
    opt = torch.optim.Adam(model.parameters())
    model(torch.ones(1, 3)).sum().backward()
    opt.step()
    f = checkpoint(str(tmp_path) + '/', 7, 100, model, opt, 1.5, .5)
    assert f.endswith('/7/7_ckpt_100.pth')
    ck = torch.load(f, weights_only=True)
    assert ck['iteration'] == 100 and ck['loss'] == 1.5 and ck['perf'] == .5
    assert set(ck['model_state_dict']) == {'weight', 'bias'} and 'state' in ck['optimizer_state_dict']
