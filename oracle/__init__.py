"""CPU oracle for the iVAE ELBO hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``ivae_amd/`` may import this package:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` call it, and there only as the checker or the timed CPU baseline,
never as the thing shipped.

Two independent restatements of the reference algorithm live here:

* ``np_ref``    -- numpy, analytic backward, dtype-selectable (fp32 to mimic the
                   reference arithmetic, fp64 as a tight ground truth).
* ``torch_ref`` -- stock PyTorch CPU ops + autograd + ``torch.optim.Adam``, i.e.
                   the same third-party arithmetic the reference runs on.

Parity pinning: the reference ships no tests or golden vectors (SURVEY.md §4,
§8c), so both restatements are pinned against fixtures under ``tests/golden/``
that ``tools/make_golden.py`` generated in the development container by
importing the reference itself (torch 2.10.0).
"""
