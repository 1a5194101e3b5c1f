"""Host-side synthetic nonlinear-ICA data: same generator semantics, RNG draw order, cache-file naming
and ``.npz`` keys as the reference's ``data/data.py`` (north_star: "data generation stays on host numpy so
inputs are bit-exact").  Written from the specification of SURVEY.md §3.4 / §8(d):

  draw order under ``np.random.seed(seed)``:  L ~ U(lb,ub) (n_seg,d)  ->  [m ~ U(-5,5) if uncentered]  ->
  sources (randn | laplace | hypsecant) (n,d)  ->  mixing matrices: n_iter_4_cond draws of U(-1,1) (ds,dd) to
  get the 25th-percentile condition threshold, then rejection sampling of column-normalised U(0,2)-1
  matrices  ->  [noise]                                              (data/data.py:134-197, 69-131, 200-304)

``ResidentDataset`` holds a dataset as three whole tensors for the device-resident loops (the reference's torch
``Dataset`` classes feed a DataLoader, which those loops do not use).  No acceleration here: this is plumbing either
side of the hot path.
"""
import os

import numpy as np
import torch


def to_one_hot(x, m=None):
    """List of label vectors -> list of (n, m) one-hot matrices of the labels' dtype (data.py:20-33)."""
    xs = x if isinstance(x, list) else [x]
    if m is None:
        m = max(xi.max() + 1 for xi in xs)
    out = []
    for xi in xs:
        oh = np.zeros((xi.size, int(m)), dtype=xs[0].dtype)
        oh[np.arange(xi.size), xi.astype(int)] = 1
        out.append(oh)
    return out


def lrelu(x, neg_slope):
    """Leaky ReLU as the reference evaluates it (data.py:36-56): element-wise in Python floats, i.e. the
    negative branch is a float64 product that the caller rounds back to the data dtype."""
    assert neg_slope > 0
    x = np.asarray(x)
    x64 = x.astype(np.float64)
    return np.where(x64 > 0, x64, x64 * neg_slope)


def sigmoid(x):
    return 1 / (1 + np.exp(-x))


def _column_normalised(A):
    for i in range(A.shape[1]):
        A[:, i] /= np.sqrt((A[:, i] ** 2).sum())
    return A


def _trial_conds(n_iter, d_sources, d_data, dtype):
    """Condition numbers of n_iter column-normalised uniform matrices: the reference's loop (data.py:100-105: draw,
    normalise, np.linalg.cond, 10^4 times per mixing layer -- 3 s per dataset at d = 10, 12 s at d = 40) as stacked array
    operations with the SAME bits: one uniform draw of the whole stack consumes the generator's stream exactly as the
    n_iter draws do; the column norms reduce over a contiguous last axis, so numpy's pairwise summation sees each
    column as the loop's 1-D `.sum()` does; np.linalg.cond of a stack calls the same LAPACK routine per matrix
    (tests/test_data_cpu.py compares it with the loop)."""
    out = []
    for i0 in range(0, n_iter, 2048):                                  # chunks: bounded memory at large d
        m = min(2048, n_iter - i0)
        A = np.random.uniform(-1, 1, (m, d_sources, d_data)).astype(dtype)
        sq = np.ascontiguousarray((A ** 2).transpose(0, 2, 1))            # (m, column, row): rows contiguous
        A /= np.sqrt(sq.sum(axis=-1))[:, None, :]
        out.append(np.linalg.cond(A))
    return np.concatenate(out)


def generate_mixing_matrix(d_sources, d_data=None, lin_type='uniform', cond_threshold=25, n_iter_4_cond=None,
                           dtype=np.float32, staircase=False):
    """Well-conditioned random mixing matrix (data.py:69-131)."""
    if d_data is None:
        d_data = d_sources
    if lin_type == 'orthogonal':
        return np.linalg.qr(np.random.uniform(-1, 1, (d_sources, d_data)))[0].astype(dtype)
    if lin_type != 'uniform':
        raise ValueError('incorrect method')

    if n_iter_4_cond is None:
        thresh = cond_threshold
    else:
        thresh = np.percentile(_trial_conds(int(n_iter_4_cond), d_sources, d_data, dtype), 25)

    def draw():
        if not staircase:
            return _column_normalised((np.random.uniform(0, 2, (d_sources, d_data)) - 1).astype(dtype))
        first = np.zeros((d_sources, 1))
        first[0, 0] = 1
        rest = np.random.uniform(0, 2, (d_sources, d_data - 1)) - 1
        if d_data > 2:
            rest[0] = 0
        return _column_normalised(np.concatenate([first, rest], axis=1).astype(dtype))

    A = draw()
    while np.linalg.cond(A) > thresh:
        A = draw()
    return A


def generate_nonstationary_sources(n_per_seg, n_seg, d, prior='gauss', var_bounds=np.array([0.5, 3]),
                                   dtype=np.float32, uncentered=False, centers=None, staircase=False):
    """Piecewise-stationary sources: per-segment scale L and mean m (data.py:134-197)."""
    n = n_per_seg * n_seg
    L = np.random.uniform(var_bounds[0], var_bounds[1], (n_seg, d))
    if uncentered:
        if centers is not None:
            assert centers.shape == (n_seg, d)
            m = centers
        else:
            m = np.random.uniform(-5, 5, (n_seg, d))
    else:
        m = np.zeros((n_seg, d))
    if staircase:
        m1 = (3 * np.arange(n_seg).reshape((-1, 1)))[np.random.permutation(n_seg)]
        m2 = np.random.uniform(-1, 1, (n_seg, d - 1)) if uncentered else np.zeros((n_seg, d - 1))
        m = np.concatenate([m1, m2], axis=1)

    if prior == 'lap':
        sources = np.random.laplace(0, 1 / np.sqrt(2), (n, d)).astype(dtype)
    elif prior == 'hs':
        from scipy.stats import hypsecant
        sources = hypsecant.rvs(0, 1, (n, d)).astype(dtype)
    elif prior == 'gauss':
        sources = np.random.randn(n, d).astype(dtype)
    else:
        raise ValueError('incorrect dist')
    labels = np.zeros(n, dtype=dtype)
    for seg in range(n_seg):
        rows = slice(n_per_seg * seg, n_per_seg * (seg + 1))
        sources[rows] *= L[seg]
        sources[rows] += m[seg]
        labels[rows] = seg
    return sources, labels, m, L


def generate_data(n_per_seg, n_seg, d_sources, d_data=None, n_layers=3, prior='gauss', activation='lrelu', batch_size=0,
                  seed=10, slope=.1, var_bounds=np.array([0.5, 3]), lin_type='uniform', n_iter_4_cond=1e4,
                  dtype=np.float32, noisy=0, uncentered=False, centers=None, staircase=False, discrete=False,
                  one_hot_labels=True, repeat_linearity=False):
    """Sources mixed by a random MLP (data.py:200-304) -> (S, X, U, M, L), or lists of batches."""
    if seed is not None:
        np.random.seed(seed)
    if d_data is None:
        d_data = d_sources
    S, U, M, L = generate_nonstationary_sources(n_per_seg, n_seg, d_sources, prior=prior, var_bounds=var_bounds,
                                                dtype=dtype, uncentered=uncentered, centers=centers,
                                                staircase=staircase)
    n = n_per_seg * n_seg
    if activation == 'lrelu':
        act = lambda t: lrelu(t, slope).astype(dtype)
    elif activation == 'sigmoid':
        act = sigmoid
    elif activation == 'xtanh':
        act = lambda t: np.tanh(t) + slope * t
    elif activation == 'none':
        act = lambda t: t
    else:
        raise ValueError('incorrect non linearity: {}'.format(activation))

    if not repeat_linearity:
        X = S.copy()
        for nl in range(n_layers):
            A = generate_mixing_matrix(X.shape[1], d_data, lin_type=lin_type, n_iter_4_cond=n_iter_4_cond, dtype=dtype,
                                       staircase=staircase)
            X = np.dot(X, A) if nl == n_layers - 1 else act(np.dot(X, A))
    else:
        assert n_layers > 1
        A = generate_mixing_matrix(d_sources, d_data, lin_type=lin_type, n_iter_4_cond=n_iter_4_cond, dtype=dtype)
        X = act(np.dot(S, A))
        B = A if d_sources == d_data else generate_mixing_matrix(d_data, lin_type=lin_type,
                                                                n_iter_4_cond=n_iter_4_cond, dtype=dtype)
        for nl in range(1, n_layers):
            X = np.dot(X, B) if nl == n_layers - 1 else act(np.dot(X, B))

    if noisy:
        X += noisy * np.random.randn(*X.shape)
    if discrete:
        X = np.random.binomial(1, sigmoid(X))

    if not batch_size:
        if one_hot_labels:
            U = to_one_hot([U], m=n_seg)[0]
        return S, X, U, M, L
    idx = np.random.permutation(n)
    nb = int(n / batch_size)
    cut = lambda a: [a[idx][c * batch_size:(c + 1) * batch_size] for c in range(nb)]
    Sb, Xb, Ub, Mb, Lb = cut(S), cut(X), cut(U), cut(M), cut(L)
    if one_hot_labels:
        Ub = to_one_hot(Ub, m=n_seg)
    return Sb, Xb, Ub, Mb, Lb


def dataset_path(root, nps, ns, dl, dd, nl, s, p, a, uncentered=False, noisy=False):
    """Cache file name of data.py:358-364."""
    path = root + 'tcl_' + '_'.join([str(nps), str(ns), str(dl), str(dd), str(nl), str(s), p, a])
    if uncentered:
        path += '_u'
    if noisy:
        path += '_noisy'
    return path + '.npz'


def save_data(path, *args, **kwargs):
    """data.py:307-315: generate once, cache as npz with keys s, x, u, m, L."""
    kwargs['batch_size'] = 0
    S, X, U, M, L = generate_data(*args, **kwargs)
    print('Creating dataset {} ...'.format(path))
    d = os.path.dirname(path)
    if d and not os.path.exists(d):
        os.makedirs(d)
    np.savez_compressed(path, s=S, x=X, u=U, m=M, L=L)
    print(' ... done')


class ResidentDataset:
    """The observations x, auxiliary variables u and true sources s of one dataset as three whole tensors.

    This is all the training loops here need of a dataset: they keep it resident in HBM and form batches on the device
    from an index permutation, so there is no per-sample `__getitem__`, no DataLoader and no worker processes (the
    reference's torch Dataset classes, data/data.py:318-413, exist to feed a DataLoader)."""

    def __init__(self, x, u, s=None):
        as_t = lambda a: a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
        self.x, self.u = as_t(x), as_t(u)
        self.s = as_t(s) if s is not None else self.x        # no ground truth: scores are taken against x
        if not (self.x.shape[0] == self.u.shape[0] == self.s.shape[0]):
            raise ValueError('x, u and s hold different numbers of samples')

    @classmethod
    def cached(cls, root, nps, ns, dl, dd, nl, s, p, a, uncentered=False, noisy=False):
        """The dataset `configs/*.yaml` names (keys nps ns dl dd nl s p act), generated once by `generate_data` and
        kept as `<root>tcl_<...>.npz` under the reference's file name and keys, so either code base reads the other's
        cache (data/data.py:357-372).  `s = None` (no seed) always regenerates."""
        path = dataset_path(root, nps, ns, dl, dd, nl, s, p, a, uncentered, noisy)
        if s is None or not os.path.exists(path):
            save_data(path, n_per_seg=nps, n_seg=ns, d_sources=dl, d_data=dd, n_layers=nl, prior=p, activation=a,
                      seed=s, uncentered=uncentered, noisy=noisy, repeat_linearity=True, one_hot_labels=True)
        print('loading data from {}'.format(path))
        with np.load(path) as z:
            out = cls(z['x'], z['u'], z['s'])
            out.mixing_scales, out.segment_means = z['L'], z['m']
        return out

    def __len__(self):
        return self.x.shape[0]

    def get_dims(self):
        """(data_dim, latent_dim, aux_dim)"""
        return self.x.shape[1], self.s.shape[1], self.u.shape[1]

    def to(self, device):
        out = ResidentDataset(self.x.to(device), self.u.to(device), self.s.to(device) if self.s is not self.x else None)
        return out
