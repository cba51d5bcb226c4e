#!/usr/bin/env python3
"""cProfile of the device-resident filter's cycles (N = 1e5): where the host time of an update goes."""
import cProfile, os, pstats, sys, io
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from triangular_transport_toolbox_amd import entf
import torch
N = 100000
rng = np.random.default_rng(0)
ens = rng.standard_normal((N, 3)) * [8, 9, 8] + [0, 0, 25]
flt = entf.Filter(N, seed=0)
r = flt.benchmark(ens, np.array([1.0, 1.0, 25.0]), 100)
print('ms per cycle', r['ms_per_cycle'])
pr = cProfile.Profile()
pr.enable()
r = flt.benchmark(ens, np.array([1.0, 1.0, 25.0]), 200, warmup=0)
pr.disable()
print('profiled ms per cycle', r['ms_per_cycle'])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45)
print(s.getvalue()[:9000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(30)
print(s.getvalue()[:6000])
