#!/usr/bin/env python3
"""Where one update of the device-resident filter (entf.Filter, N = 1e5) spends its time: phases with synchronisation
between them (so the sum exceeds the pipelined cycle), then a cProfile of unsynchronised cycles."""
import cProfile
import os
import pstats
import sys
import time

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
sync = torch.cuda.synchronize


def timed(f):
    sync(); t0 = time.perf_counter(); r = f(); sync()
    return 1e3 * (time.perf_counter() - t0), r


acc = {}
for rep in range(10):
    flt.forecast()
    inp, ensd = flt._inp, flt.ens
    for idx, perm in enumerate(entf.PERMUTATIONS):
        import ctypes
        t, _ = timed(lambda: (tm.map_columns([-1] + list(perm), 4, N, source=ensd, out=inp),
                              entf._check(tm._lib.ttm_perturb(ctypes.c_void_p(ensd.data_ptr() + 8 * idx * ensd.shape[1]), None, 2.0, 0, rep * 3 + idx + 100, 0, N, tm._ptr(inp), tm._stream()))))
        acc['assemble'] = acc.get('assemble', 0) + t
        t, _ = timed(lambda: tm.reset_device(inp, N)); acc['reset_device'] = acc.get('reset_device', 0) + t
        t, _ = timed(lambda: tm.optimize()); acc['optimize'] = acc.get('optimize', 0) + t
        t, Z = timed(lambda: tm.forward_device(tm._Xs, N)); acc['pack+forward'] = acc.get('pack+forward', 0) + t
        ystar = (obs[idx] - tm.X_mean[0]) / tm.X_std[0]
        t, Xc = timed(lambda: tm.map_columns([-1, -1, -1, -1], 4, N, shift=[ystar, 0.0, 0.0, 0.0])); acc['xc'] = acc.get('xc', 0) + t
        t, _ = timed(lambda: tm.inverse_device(Z, N, X=Xc)); acc['inverse'] = acc.get('inverse', 0) + t
        src = [1 + p for p in perm]
        t, _ = timed(lambda: tm.map_columns(src, 3, N, source=Xc, scale=[tm.X_std[c] for c in src], shift=[tm.X_mean[c] for c in src], out=ensd)); acc['back'] = acc.get('back', 0) + t
print('per update (ms, synchronised phases):', {k: round(v / 30, 3) for k, v in acc.items()}, 'sum %.3f' % (sum(acc.values()) / 30))
print('evaluations last update', getattr(tm, 'last_optimize_evaluations', None))
t0 = time.perf_counter()
for _ in range(50):
    flt.forecast(); flt.assimilate(obs)
sync()
print('pipelined: %.3f ms per cycle' % (1e3 * (time.perf_counter() - t0) / 50))
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    flt.forecast(); flt.assimilate(obs)
sync(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(30)
pstats.Stats(pr).sort_stats('tottime').print_stats(28)
