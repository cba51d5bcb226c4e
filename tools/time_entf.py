#!/usr/bin/env python3
"""Where one EnTF update (reset -> optimize -> map -> inverse_map, Example-06 map, N = 1e5) spends its time."""
import os
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
tm = entf.make_filter_map(N)


def timed(f):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = f()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, r


for rep in range(3):
    Yt = ens[:, 0] + 2 * rng.standard_normal(N)
    mi = np.column_stack((Yt[:, None], ens))
    t_reset, _ = timed(lambda: tm.reset(mi.copy()))
    t_opt, _ = timed(lambda: tm.optimize())
    t_map, Z = timed(lambda: tm.map(mi))
    Ys = np.full((N, 1), 1.0)
    t_inv, _ = timed(lambda: tm.inverse_map(X_star=Ys, Z=Z))
    print('rep %d: reset %.2f ms, optimize %.2f ms, map %.2f ms, inverse %.2f ms' % (rep, 1e3 * t_reset, 1e3 * t_opt, 1e3 * t_map, 1e3 * t_inv),
          'u_enabled', tm._cm.u_enabled, 'hot', tm._cm.u_h_cls, flush=True)
import cProfile
import pstats
pr = cProfile.Profile()
pr.enable()
tm.reset(mi.copy()); tm.optimize(); Z = tm.map(mi); tm.inverse_map(X_star=Ys, Z=Z)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(25)
