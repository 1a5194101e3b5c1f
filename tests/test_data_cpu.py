"""Host data generator (ivae_amd/data.py) vs the reference generator's outputs recorded in
tests/golden/datasets.json (tools/make_golden.py datasets): same legacy np.random stream, same bytes.
Bit-exactness holds on the numpy/BLAS build the fixture was made with (this image); elsewhere the
leading values and sums are compared at fp32 tolerance."""
import hashlib
import json
import os

import numpy as np
import pytest

from ivae_amd import data as D

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "datasets.json")))


@pytest.mark.parametrize("i", range(len(GOLD)))
def test_generator_matches_reference(i):
    rec = GOLD[i]
    S, X, U, M, L = D.generate_data(batch_size=0, repeat_linearity=True, **rec["kwargs"])
    exact = True
    for k, a in (("s", S), ("x", X), ("u", U), ("m", M), ("L", L)):
        a = np.ascontiguousarray(a)
        assert list(a.shape) == rec[k]["shape"] and str(a.dtype) == rec[k]["dtype"], k
        np.testing.assert_allclose(a.reshape(-1)[:6], rec[k]["head"], rtol=1e-5, atol=1e-6)
        assert abs(float(a.astype(np.float64).sum()) - rec[k]["sum"]) <= 1e-4 * max(1.0, abs(rec[k]["sum"])), k
        exact &= hashlib.sha256(a.tobytes()).hexdigest() == rec[k]["sha256"]
    assert exact, "generator output is numerically close to the reference but not bit-identical"


def test_dataset_cache_name_and_keys(tmp_path):
    root = str(tmp_path) + "/"
    ds = D.ResidentDataset.cached(root, 20, 3, 2, 2, 3, 1, 'gauss', 'xtanh')
    assert os.path.exists(root + "tcl_20_3_2_2_3_1_gauss_xtanh.npz")       # the reference's cache name (data.py:358-364)
    z = np.load(root + "tcl_20_3_2_2_3_1_gauss_xtanh.npz")
    assert sorted(z.files) == ["L", "m", "s", "u", "x"]                     # ... and keys (data.py:314)
    assert ds.get_dims() == (2, 2, 3) and len(ds) == 60
    assert ds.x.shape == (60, 2) and ds.u.shape == (60, 3) and bool((ds.u.sum(1) == 1).all())
    again = D.ResidentDataset.cached(root, 20, 3, 2, 2, 3, 1, 'gauss', 'xtanh')     # second call reads the cache
    assert bool((again.x == ds.x).all())
    assert D.dataset_path("r/", 1, 2, 3, 4, 5, 6, 'lap', 'lrelu', True, True) == "r/tcl_1_2_3_4_5_6_lap_lrelu_u_noisy.npz"
    cd = D.ResidentDataset(ds.x.numpy(), ds.u.numpy())                      # no ground truth: s falls back to x
    assert cd.get_dims() == (2, 2, 3) and cd.s is cd.x


@pytest.mark.parametrize("ds,dd", [(2, 2), (5, 5), (8, 8), (10, 10), (5, 8), (20, 20), (40, 40)])
def test_condition_number_trials_are_the_reference_loop_bit_for_bit(ds, dd):
    """The stacked form of the generator's 10^4 condition-number trials (data.py:100-105) against the loop it replaces:
    same condition numbers, same generator state afterwards (what keeps every later draw -- the datasets -- bit-exact)."""
    from ivae_amd import data

    n = 600
    np.random.seed(100 + ds)
    want = np.array([np.linalg.cond(data._column_normalised(np.random.uniform(-1, 1, (ds, dd)).astype(np.float32)))
                     for _ in range(n)])
    after_loop = np.random.uniform()
    np.random.seed(100 + ds)
    got = data._trial_conds(n, ds, dd, np.float32)
    assert got.dtype == want.dtype and np.array_equal(got, want)
    assert np.random.uniform() == after_loop
    assert np.percentile(list(want), 25) == np.percentile(got, 25)


def test_sweep_dataset_cache_round_trip(tmp_path):
    """`python -m ivae_amd.sweep --data-path DIR`: datasets are left in the reference's cache format (file name and keys of
    data/data.py:357-372) and loaded on the next run -- the same arrays as a fresh generation, bit for bit."""
    from ivae_amd.data import dataset_path
    from ivae_amd.sweep import host_datasets_numpy
    jobs = [(3, 0), (3, 1), (4, 0)]
    root = str(tmp_path) + os.sep
    made = host_datasets_numpy(5, 100, jobs, cache=root)
    for d, sd in jobs:
        path = dataset_path(root, 100, 5, d, d, 3, sd, "gauss", "lrelu")
        with np.load(path) as z:
            assert sorted(z.files) == ["L", "m", "s", "u", "x"]
    again = host_datasets_numpy(5, 100, jobs, cache=root)          # loaded
    fresh = host_datasets_numpy(5, 100, jobs)                       # generated
    for k in jobs:
        for a, b, c in zip(made[k], again[k], fresh[k]):
            assert a.dtype == np.float32 and np.array_equal(a, b) and np.array_equal(a, c)
