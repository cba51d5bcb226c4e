"""np.quantile('linear') restated on top of exact order statistics is bit-identical to NumPy."""
import numpy as np

from triangular_transport_toolbox_amd import quantile


def test_bit_identical_to_numpy():
    rng = np.random.default_rng(0)
    for n in (1, 2, 3, 10, 501, 1000, 4097):
        x = rng.standard_normal(n) * 3 + 1
        if n > 10:
            x[::7] = x[3]           # ties
        xs = np.sort(x)
        for q in ([0.5], [0.2, 0.4, 0.6, 0.8], list(np.arange(1, 10) / 10), [0.0, 1.0, 0.8413447460685429, 0.15865525393145707],
                  list(np.arange(1, 5) / 5)):
            got = quantile.quantile_from_order_statistics(n, q, lambda r: xs[r])
            assert np.array_equal(got, np.quantile(x, q)), (n, q)
