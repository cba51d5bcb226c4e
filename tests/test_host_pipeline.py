"""map() / inverse_map() on large host arrays: chunks of rows pipelined over PCIe (transport_map._host_pipeline) give the very
same bits as the one-copy path, for ensemble sizes that do not divide into the chunks."""
import numpy as np
import pytest

from tests.util import coeff_lists, load_case


@pytest.mark.gpu
@pytest.mark.parametrize('N', [262144 + 4097, 700001])
def test_pipelined_host_boundary_equals_the_plain_path(N):
    from triangular_transport_toolbox_amd import specs
    from triangular_transport_toolbox_amd.transport_map import transport_map
    cfg = specs.config('C5')
    X = cfg['sampler'](N)
    npz, desc = load_case('c5_sep')
    tm = transport_map(X=X[:20000], monotone=cfg['monotone'], nonmonotone=cfg['nonmonotone'], verbose=False, **cfg['kwargs'])
    tm.coeffs_mon, tm.coeffs_nonmon = coeff_lists(npz, tm.D)
    assert tm._pipe_ok(N)
    Zp = tm.map(X)
    Xp = tm.inverse_map(Zp)
    tm.host_pipeline = False
    assert not tm._pipe_ok(N)
    Z = tm.map(X)
    Xi = tm.inverse_map(Z)
    assert Zp.shape == Z.shape == (N, tm.D) and np.array_equal(Zp, Z)
    assert Xp.shape == Xi.shape and np.array_equal(Xp, Xi)
    # results are fresh, writable arrays; inputs are never mutated
    X0 = X[:64].copy()
    assert Zp.flags.writeable
    Zp[:] = 0.0
    assert np.array_equal(X[:64], X0)
