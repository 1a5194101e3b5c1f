"""Host-side synthetic nonlinear-ICA data: same generator semantics, RNG draw order, cache-file naming
and ``.npz`` keys as the reference's ``data/data.py`` (north_star: "data generation stays on host numpy so
inputs are bit-exact").  Written from the specification of SURVEY.md §3.4 / §8(d):

  draw order under ``np.random.seed(seed)``:  L ~ U(lb,ub) (n_seg,d)  ->  [m ~ U(-5,5) if uncentered]  ->
  sources (randn | laplace | hypsecant) (n,d)  ->  mixing matrices: n_iter_4_cond draws of U(-1,1) (ds,dd) to
  get the 25th-percentile condition threshold, then rejection sampling of column-normalised U(0,2)-1
  matrices  ->  [noise]                                              (data/data.py:134-197, 69-131, 200-304)

The torch ``Dataset`` wrappers keep the reference's constructor signatures and ``__getitem__`` tuples
(data/data.py:318-413).  No acceleration here: this is plumbing either side of the hot path.
"""
import os

import numpy as np
import torch
from torch.utils.data import Dataset


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
        conds = []
        for _ in range(int(n_iter_4_cond)):
            conds.append(np.linalg.cond(_column_normalised(np.random.uniform(-1, 1, (d_sources, d_data)).astype(dtype))))
        thresh = np.percentile(conds, 25)

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


class SyntheticDataset(Dataset):
    """data.py:318-378."""

    def __init__(self, root, nps, ns, dl, dd, nl, s, p, a, uncentered=False, noisy=False, centers=None, double=False,
                 one_hot_labels=True):
        self.root = root
        data = self.load_tcl_data(root, nps, ns, dl, dd, nl, s, p, a, uncentered, noisy, centers, one_hot_labels)
        self.data = data
        self.s = torch.from_numpy(data['s'])
        self.x = torch.from_numpy(data['x'])
        self.u = torch.from_numpy(data['u'])
        self.l = data['L']
        self.m = data['m']
        self.len = self.x.shape[0]
        self.latent_dim = self.s.shape[1]
        self.aux_dim = self.u.shape[1]
        self.data_dim = self.x.shape[1]
        self.prior, self.activation, self.seed, self.n_layers = p, a, s, nl
        self.uncentered, self.noisy, self.double, self.one_hot_labels = uncentered, noisy, double, one_hot_labels

    def get_dims(self):
        return self.data_dim, self.latent_dim, self.aux_dim

    def __len__(self):
        return self.len

    def __getitem__(self, index):
        if not self.double:
            return self.x[index], self.u[index], self.s[index]
        index2 = np.random.choice(range(len(self)))
        return self.x[index], self.x[index2], self.u[index], self.s[index]

    @staticmethod
    def load_tcl_data(root, nps, ns, dl, dd, nl, s, p, a, uncentered, noisy, centers, one_hot_labels):
        path = dataset_path(root, nps, ns, dl, dd, nl, s, p, a, uncentered, noisy)
        if not os.path.exists(path) or s is None:
            save_data(path, n_per_seg=nps, n_seg=ns, d_sources=dl, d_data=dd, n_layers=nl, prior=p, activation=a,
                      seed=s, batch_size=0, uncentered=uncentered, noisy=noisy, centers=centers, repeat_linearity=True,
                      one_hot_labels=one_hot_labels)
        print('loading data from {}'.format(path))
        return np.load(path)

    def get_test_sample(self, batch_size, seed=None):
        if seed is not None:
            np.random.seed(seed)
        idx = np.random.randint(max(0, self.len - batch_size))
        return self.x[idx:idx + batch_size], self.u[idx:idx + batch_size], self.s[idx:idx + batch_size]


class CustomSyntheticDataset(Dataset):
    """data.py:381-413."""

    def __init__(self, X, U, S=None, device='cpu'):
        self.device = device
        self.x = torch.from_numpy(X).to(self.device)
        self.u = torch.from_numpy(U).to(self.device)
        self.s = torch.from_numpy(S).to(self.device) if S is not None else self.x
        self.len = self.x.shape[0]
        self.latent_dim = self.s.shape[1]
        self.aux_dim = self.u.shape[1]
        self.data_dim = self.x.shape[1]
        self.nps = int(self.len / self.aux_dim)
        print('data loaded on {}'.format(self.x.device))

    def get_dims(self):
        return self.data_dim, self.latent_dim, self.aux_dim

    def __len__(self):
        return self.len

    def __getitem__(self, index):
        return self.x[index], self.u[index], self.s[index]

    def get_metadata(self):
        return {'nps': self.nps, 'ns': self.aux_dim, 'n': self.len, 'latent_dim': self.latent_dim,
                'data_dim': self.data_dim, 'aux_dim': self.aux_dim}
