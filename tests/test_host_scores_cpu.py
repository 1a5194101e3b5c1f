"""Host side of the batch MCC (metrics/mcc.py:333-335): `runners._assignment_scores` forms, for every correlation block,
`cc[linear_sum_assignment(-cc)].mean()`.  Up to d = 6 it scores all d! assignments of all blocks with one matrix product
instead of calling scipy per block: the results must be the scipy loop's, bit for bit, including the 0 the reference's bare
`except` gives a block that is not finite (runners/ivae.py:85-88)."""
import numpy as np
import pytest
from scipy.optimize import linear_sum_assignment

from ivae_amd.runners import _assignment_scores


def _loop(cc_abs):
    return [float(cc[linear_sum_assignment(-1 * cc)].mean()) if np.isfinite(cc).all() else 0. for cc in cc_abs]


@pytest.mark.parametrize("d", [1, 2, 3, 4, 5, 6, 7, 10])
def test_scores_equal_scipy_loop(d):
    r = np.random.RandomState(d)
    cc = np.abs(np.clip(r.randn(3000, d, d) * .4, -1, 1))
    cc[5, 0, 0] = np.nan                      # a collapsed latent: the kernels hand nan through
    cc[9] = np.inf
    cc[11] = 1.                               # every assignment ties
    cc[12] = np.eye(d)
    got = _assignment_scores(cc)
    assert got == _loop(cc)
    assert got[5] == 0. and got[9] == 0. and got[11] == 1. and got[12] == 1.


def test_scores_of_no_blocks():
    assert _assignment_scores(np.zeros((0, 5, 5))) == []
