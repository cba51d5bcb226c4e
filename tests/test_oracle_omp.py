"""The C++/OpenMP leg of the CPU oracle (oracle/ttm_oracle_omp.cpp, the all-core `cpu_baseline` of bench.py) against
the reference's outputs in tests/golden/ and against the NumPy oracle.  CPU only."""
import numpy as np
import pytest

from tests import util
from oracle import omp


@pytest.mark.parametrize('name', ['c2b_sep', 'c3_sep', 'c5_sep'])
def test_forward_matches_reference_golden(name):
    npz, desc = util.load_case(name)
    om = util.make_oracle(name, npz, desc)
    X = util.case_X(name, npz)
    n = npz['Z'].shape[0]
    Z = omp.OmpMap(om).map(X[:n], threads=2)
    assert util.relerr(Z, npz['Z']) < 1e-12


@pytest.mark.parametrize('name', ['c2b_sep', 'c3_sep', 'c5_sep'])
def test_table_inverse_matches_reference_golden(name):
    npz, desc = util.load_case(name)
    om = util.make_oracle(name, npz, desc)
    Xi = omp.OmpMap(om).inverse_map(npz['inv_Z'], threads=2)
    assert util.relerr(Xi, npz['inv_X_table']) < 1e-10


def test_matches_numpy_oracle_on_a_larger_sample_and_is_thread_invariant():
    npz, desc = util.load_case('c5_sep')
    om = util.make_oracle('c5_sep', npz, desc)
    from triangular_transport_toolbox_amd import specs
    X = specs.sample_mixture(3000, seed=99)
    m = omp.OmpMap(om)
    Z1, Z4 = m.map(X, threads=1), m.map(X, threads=4)
    assert np.array_equal(Z1, Z4)
    assert util.relerr(Z1, om.map(X)) < 1e-12
    Xi = m.inverse_map(Z1, threads=3)
    assert util.relerr(Xi, om.inverse_map(Z1)) < 1e-10


def test_usable_cores_reports_affinity_and_quota():
    n, info = omp.usable_cores()
    assert 1 <= n <= info['affinity']
