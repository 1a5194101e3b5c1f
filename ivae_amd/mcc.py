"""Mean correlation coefficient with the correlation matrix computed on the GPU.

Counterpart of ``mean_corr_coef`` / ``mean_corr_coef_np`` (metrics/mcc.py:311-346): the Pearson
cross-correlation block is computed by the ``ivae_corr`` HIP kernel in float64 (numpy promotes the
fp32 inputs the same way); the Hungarian assignment stays on the host (scipy), as in the reference.
"""
import numpy as np
import torch
from scipy.optimize import linear_sum_assignment

from . import ops


def _to_dev(a, device):
    if isinstance(a, np.ndarray):
        a = torch.from_numpy(np.ascontiguousarray(a))
    return a.detach().to(device=device, dtype=torch.float32)


def _ranks(t):
    """Column-wise ranks 1..n (no tie averaging: continuous data), for method='spearman'."""
    order = t.argsort(dim=0)
    r = torch.empty_like(order)
    ar = torch.arange(1, t.shape[0] + 1, device=t.device).unsqueeze(1).expand_as(order)
    r.scatter_(0, order, ar)
    return r.to(torch.float32)


def corr_matrix_gpu(x, y, method='pearson', device=None):
    """|d x d| cross-correlation block as a float64 device tensor (no host sync)."""
    if device is None:
        device = x.device if isinstance(x, torch.Tensor) and x.is_cuda else torch.device('cuda')
    x, y = _to_dev(x, device), _to_dev(y, device)
    if method == 'spearman':
        x, y = _ranks(x), _ranks(y)
    elif method != 'pearson':
        raise ValueError('not a valid method: {}'.format(method))
    return ops.corr(x, y)


def mean_corr_coef_gpu(x, y, method='pearson', device=None):
    """metrics/mcc.py:326-335 with the corrcoef on the GPU."""
    cc = corr_matrix_gpu(x, y, method, device).abs().cpu().numpy()
    return cc[linear_sum_assignment(-1 * cc)].mean()


def mean_corr_coef(x, y, method='pearson'):
    """Dispatcher with the reference's type checks (metrics/mcc.py:338-346)."""
    if type(x) != type(y):
        raise ValueError('inputs are of different types: ({}, {})'.format(type(x), type(y)))
    if isinstance(x, (np.ndarray, torch.Tensor)):
        return mean_corr_coef_gpu(x, y, method)
    raise ValueError('not a supported input type: {}'.format(type(x)))
