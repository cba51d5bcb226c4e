#!/usr/bin/env python3
"""ms per cycle of the device-resident filter (N = 1e5, 300 cycles), BASELINE configs[3]."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from triangular_transport_toolbox_amd import entf
N = 100000
rng = np.random.default_rng(0)
ens = rng.standard_normal((N, 3)) * [8, 9, 8] + [0, 0, 25]
flt = entf.Filter(N, seed=0)
for rep in range(3):
    r = flt.benchmark(ens, np.array([1.0, 1.0, 25.0]), int(os.environ.get("TTM_FILTER_CYCLES", 300)))
    print('ms per cycle %.3f  rmse %.3f' % (r['ms_per_cycle'], r['rmse_last']), flush=True)
