"""Logging / checkpoint helpers with the on-disk contracts of the reference's utils/utils.py:
``checkpoint`` writes ``<path>/<exp_id>/<exp_id>_ckpt_<it>.pth`` holding
{iteration, model_state_dict, optimizer_state_dict, loss, perf} (utils.py:55-64); ``Logger`` keeps
running means per key, appends them per epoch and hands out experiment ids from a flock-protected
``.expid`` counter (utils.py:107-225).  Host plumbing around the hot path."""
import fcntl
import json
import os
import time

import numpy as np
import torch


def make_dir(dir_name):
    if not dir_name.endswith('/'):
        dir_name += '/'
    os.makedirs(dir_name, exist_ok=True)
    return dir_name


def checkpoint(path, exp_id, iteration, model, optimizer, loss, perf):
    sub = make_dir(path + str(exp_id) + '/')
    target = sub + str(exp_id) + '_ckpt_' + str(iteration) + '.pth'
    print('.. checkpoint at iteration {} ..'.format(iteration))
    torch.save({'iteration': iteration, 'model_state_dict': model.state_dict(),
                'optimizer_state_dict': optimizer.state_dict(), 'loss': loss, 'perf': perf}, target)
    return target


class Averager:
    def __init__(self):
        self.reset()

    def reset(self):
        self.val, self.count, self.avg, self.sum = 0, 0, 0, 0

    def update(self, val):
        self.val = val
        self.sum += val
        self.count += 1
        self.avg = self.sum / self.count


class Logger:
    def __init__(self, log_dir=None, **metadata):
        self.metadata = metadata
        self.log_dict, self.running_means = {}, {}
        self.log_dir = log_dir
        self.exp_id = self._get_exp_id(log_dir) if log_dir is not None else None

    def add(self, key):
        self.running_means[key] = Averager()
        self.log_dict[key] = []

    def update(self, key, val):
        self.running_means[key].update(val)

    def log(self):
        for key in self.keys():
            self.log_dict[key].append(self.running_means[key].avg * 1.)
            self.running_means[key].reset()

    def get_last(self, key):
        return self.log_dict[key][-1]

    def keys(self):
        return self.log_dict.keys()

    def __len__(self):
        return len(self.log_dict)

    def add_metadata(self, **metadata):
        self.metadata.update(metadata)

    def save_to_npz(self, log_dir=None):
        log_dir = log_dir if log_dir is not None else self.log_dir
        if log_dir is None:
            print("Can't save to npz: log path not specified")
            return
        path = make_dir(log_dir) + '{}.npz'.format(self._get_exp_id(log_dir))
        np.savez_compressed(path, **{k: np.array(v) for k, v in self.log_dict.items()})
        print('Log data saved to {}'.format(path))

    def save_to_json(self, log_dir=None, method='last'):
        log_dir = log_dir if log_dir is not None else self.log_dir
        if log_dir is None:
            print("Can't save to json: log path not specified")
            return
        if method not in ('last', 'full'):
            raise ValueError('Incorrect method {}'.format(method))
        rec = {'id': self._get_exp_id(log_dir)}
        for k in self.keys():
            rec[k] = self.get_last(k) if method == 'last' else self.log_dict[k]
        rec['metadata'] = self.metadata
        with open(make_dir(log_dir) + 'log.json', 'a') as f:
            f.write(json.dumps(rec) + '\n')

    @staticmethod
    def _get_exp_id(log_folder):
        """Next experiment id from `<folder>/.expid`, exclusive-locked so parallel jobs get distinct ids."""
        helper = make_dir(log_folder) + '.expid'
        if not os.path.exists(helper):
            with open(helper, 'w') as f:
                f.write('0')
        with open(helper, 'r+') as f:
            deadline = time.time() + 30
            while True:
                try:
                    fcntl.flock(f, fcntl.LOCK_EX | fcntl.LOCK_NB)
                    break
                except BlockingIOError:
                    if time.time() > deadline:
                        raise TimeoutError('Timeout on accessing log helper file {}'.format(helper))
                    time.sleep(0.1)
            cur = int(f.readline()) + 1
            f.seek(0)
            f.write(str(cur))
            f.truncate()
            fcntl.flock(f, fcntl.LOCK_UN)
        return cur
