"""CPU test of the numpy model of k_dc_chain_spec's start-value iteration (tools/dc_iteration_model.py): the
integer recurrence z' = z + e + [z < 0] evaluated lane-parallel from speculated start values must end at the sequential
trajectory, and the affine step with secant slopes must need far fewer rounds than the plain sums where the estimate is
pinned to its threshold."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import dc_iteration_model as m  # noqa: E402


def _case(rng, sigma, lanes):
    z0 = int(m.truth(np.rint(rng.normal(-0.5, sigma, 2000)).astype(np.int64), 0))
    return np.rint(rng.normal(-0.5, sigma, lanes * 16)).astype(np.int64).reshape(lanes, 16), z0


def test_a_step_that_stands_is_the_sequential_recurrence():
    rng = np.random.default_rng(1)
    for sigma in (60, 8, 1.2):
        E, z0 = _case(rng, sigma, 128)
        # starts that satisfy "every lane starts where its predecessor ended" are the sequential trajectory's
        zs = np.empty(128, np.int64)
        z = z0
        for l in range(128):
            zs[l] = z
            z = m.truth(E[l], z)
        end, _, _ = m.evaluate(E, zs)
        assert np.array_equal(end[:-1], zs[1:]) and end[-1] == z
        for scheme in ("plain", "secant", "final"):
            assert m.rounds(E, z0, scheme, limit=400) < 400, (sigma, scheme)


def test_affine_step_needs_few_rounds_where_the_plain_sums_do_not_settle():
    rng = np.random.default_rng(2)
    plain, final = [], []
    for _ in range(6):
        E, z0 = _case(rng, 1.2, 256)  # an offset many times the noise: the estimate is pinned to its threshold
        plain.append(m.rounds(E, z0, "plain", limit=60))
        final.append(m.rounds(E, z0, "final", limit=60))
    assert max(final) <= 8 and min(plain) >= 30, (plain, final)
