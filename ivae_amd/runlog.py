"""Run records for the training loops: per-key running means flushed once per epoch, experiment ids, the
checkpoint file layout.

File contracts kept from the reference so that its tooling reads our output (utils/utils.py:55-64, 145-225):
  * `checkpoint()` writes `<path>/<exp_id>/<exp_id>_ckpt_<it>.pth` = {iteration, model_state_dict,
    optimizer_state_dict, loss, perf};
  * `RunLog.save_to_json()` appends one line {id, <key>: value..., metadata} to `<dir>/log.json`,
    `save_to_npz()` writes `<dir>/<id>.npz`; ids come from the counter file `<dir>/.expid`.
Unlike the reference's Logger, values may be device tensors: `update()` only accumulates on the device and the
single device-to-host copy happens in `log()`, once per epoch, so logging never syncs the training stream.
"""
import fcntl
import json
import os
import time

import numpy as np
import torch


def _dir(path):
    path = path if path.endswith('/') else path + '/'
    os.makedirs(path, exist_ok=True)
    return path


def next_experiment_id(folder, timeout=30.0):
    """Atomically increment and return the integer in `<folder>/.expid` (created at 0)."""
    counter = _dir(folder) + '.expid'
    fd = os.open(counter, os.O_RDWR | os.O_CREAT, 0o644)
    try:
        deadline = time.monotonic() + timeout
        while True:
            try:
                fcntl.flock(fd, fcntl.LOCK_EX | fcntl.LOCK_NB)
                break
            except BlockingIOError:
                if time.monotonic() > deadline:
                    raise TimeoutError('Timeout on accessing log helper file {}'.format(counter))
                time.sleep(0.05)
        raw = os.read(fd, 64).decode().strip()
        new_id = (int(raw) if raw else 0) + 1
        os.lseek(fd, 0, os.SEEK_SET)
        os.ftruncate(fd, 0)
        os.write(fd, str(new_id).encode())
        return new_id
    finally:
        os.close(fd)                      # closing drops the lock


class _Mean:
    """Sum / count of floats or 0-dim tensors; tensors stay on their device until `value()`."""
    __slots__ = ('total', 'n')

    def __init__(self):
        self.total, self.n = 0., 0

    def add(self, v, weight=1):
        self.total = self.total + (v.detach() if isinstance(v, torch.Tensor) else v)
        self.n += weight

    def value(self):
        if self.n == 0:
            return 0.
        t = self.total
        return float(t.item() if isinstance(t, torch.Tensor) else t) / self.n


class RunLog:
    """Drop-in for the reference's `Logger` (utils/utils.py:107-193): add / update / log / get_last / keys /
    add_metadata / save_to_json / save_to_npz, plus `exp_id` and `log_dict`."""

    def __init__(self, log_dir=None, **metadata):
        self.metadata = dict(metadata)
        self.log_dict = {}
        self._means = {}
        self.log_dir = log_dir
        self.exp_id = next_experiment_id(log_dir) if log_dir is not None else None

    def add(self, key):
        self.log_dict[key] = []
        self._means[key] = _Mean()

    def update(self, key, val, weight=1):
        """`val` is one observation, or (with weight=k) the SUM of k observations."""
        self._means[key].add(val, weight)

    def log(self, keys=None):
        """Close the current averaging window (utils/utils.py:150-156).  `keys` (not in the reference): only these keys --
        the training loops log an epoch's loss when the epoch ends and its batch scores one epoch later, when the host has
        worked them out under the next epoch's launches."""
        for key in (self._means if keys is None else keys):
            self.log_dict[key].append(self._means[key].value())
            self._means[key] = _Mean()

    def get_last(self, key):
        return self.log_dict[key][-1]

    def keys(self):
        return self.log_dict.keys()

    def __len__(self):
        return len(self.log_dict)

    def add_metadata(self, **metadata):
        self.metadata.update(metadata)

    def _target_dir(self, log_dir):
        return log_dir if log_dir is not None else self.log_dir

    def save_to_npz(self, log_dir=None):
        folder = self._target_dir(log_dir)
        if folder is None:
            print("Can't save to npz: log path not specified")
            return None
        path = _dir(folder) + '{}.npz'.format(next_experiment_id(folder))
        np.savez_compressed(path, **{k: np.asarray(v) for k, v in self.log_dict.items()})
        print('Log data saved to {}'.format(path))
        return path

    def save_to_json(self, log_dir=None, method='last'):
        folder = self._target_dir(log_dir)
        if folder is None:
            print("Can't save to json: log path not specified")
            return None
        if method not in ('last', 'full'):
            raise ValueError('Incorrect method {}'.format(method))
        record = {'id': next_experiment_id(folder)}
        record.update({k: (v[-1] if method == 'last' else list(v)) for k, v in self.log_dict.items()})
        record['metadata'] = self.metadata
        path = _dir(folder) + 'log.json'
        with open(path, 'a') as f:
            f.write(json.dumps(record) + '\n')
        print('Log saved to {}'.format(path))
        return path


Logger = RunLog          # the name models/wrappers.py:10 imports


def checkpoint(path, exp_id, iteration, model, optimizer, loss, perf):
    """utils/utils.py:55-64."""
    target = '{}{}_ckpt_{}.pth'.format(_dir(path + str(exp_id)), exp_id, iteration)
    print('.. checkpoint at iteration {} ..'.format(iteration))
    torch.save({'iteration': iteration, 'model_state_dict': model.state_dict(),
                'optimizer_state_dict': optimizer.state_dict(), 'loss': loss, 'perf': perf}, target)
    return target
