#!/usr/bin/env python3
"""optimize() of the device-resident filter's map (N = 1e5): cProfile of the host side, by internal time."""
import cProfile, os, pstats, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from triangular_transport_toolbox_amd import entf  # noqa: E402
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
rng = np.random.default_rng(0)
ens = rng.standard_normal((N, 3)) * [8, 9, 8] + [0, 0, 25]
flt = entf.Filter(N, seed=0)
flt.set_ensemble(ens)
obs = np.array([1.0, 2.0, 25.0])
for _ in range(5):
    flt.forecast(); flt.assimilate(obs)
torch.cuda.synchronize()
tm = flt.tm
pr = cProfile.Profile()
tot = 0.0
for rep in range(40):
    tm.reset_device(flt._inp, N)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pr.enable()
    tm.optimize()
    pr.disable()
    torch.cuda.synchronize()
    tot += time.perf_counter() - t0
print('optimize: %.3f ms per call (under the profiler)' % (1e3 * tot / 40))
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
