"""Command line of the family-A training loop: `python -m ivae_amd --config ivae.yaml [--run DIR] [--doc TAG]
[--n-sims K] [--seed S]` -- the flags, directory layout (`<run>/{logs,checkpoints,datasets}/<doc>`) and result
file (`<run>/<config>_<seed>_<n_sims>.p`, a pickle of (perfs, loss_hists, perf_hists)) of the reference's
`main.py:13-67`.  `--config` is looked up in ./configs first, then in the package's own configs directory.
The TCL baseline (`tcl: true`) is a different algorithm on TF1 and is not part of this package."""
import argparse
import os
import pickle
import sys

import numpy as np
import torch
import yaml

from .runners import runner


def _namespace(obj):
    if isinstance(obj, dict):
        return argparse.Namespace(**{k: _namespace(v) for k, v in obj.items()})
    return obj


def _find_config(name):
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for cand in (name, os.path.join('configs', name), os.path.join(here, 'configs', name)):
        if os.path.isfile(cand):
            return cand
    raise FileNotFoundError('config file {} not found'.format(name))


def main(argv=None):
    ap = argparse.ArgumentParser(prog='python -m ivae_amd', description=__doc__.split('\n')[0])
    ap.add_argument('--config', type=str, default='ivae.yaml', help='Path to the config file')
    ap.add_argument('--run', type=str, default='run', help='Path for saving running related data.')
    ap.add_argument('--doc', type=str, default='', help='A string for documentation purpose')
    ap.add_argument('--n-sims', type=int, default=1, help='Number of simulations to run')
    ap.add_argument('--seed', type=int, default=0, help='Random seed')
    args = ap.parse_args(argv)

    for attr, sub in (('log', 'logs'), ('checkpoints', 'checkpoints'), ('data_path', 'datasets')):
        path = os.path.join(args.run, sub, args.doc)
        os.makedirs(path, exist_ok=True)
        setattr(args, attr, path)
    if not args.data_path.endswith(os.sep):
        args.data_path += os.sep            # dataset file names are appended to it (data.py:358)

    with open(_find_config(args.config)) as f:
        config = _namespace(yaml.safe_load(f))
    if getattr(config, 'tcl', False):
        raise NotImplementedError('the TCL baseline is not part of ivae_amd')
    if not torch.cuda.is_available():
        raise RuntimeError('ivae_amd has no CPU path: a ROCm device is required')
    config.device = torch.device('cuda')

    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    result = runner(args, config)
    stem = os.path.splitext(os.path.basename(args.config))[0]
    fname = os.path.join(args.run, '_'.join([stem, str(args.seed), str(args.n_sims)]) + '.p')
    with open(fname, 'wb') as f:
        pickle.dump(result, f)
    return 0


if __name__ == '__main__':
    sys.exit(main())
