#!/usr/bin/env python3
"""Wall-clock timeline of the filter update WITHOUT added synchronisation: the host methods of one update are wrapped with
perf_counter accumulators (inclusive times; the blocking host visits - .cpu() reads, the optimiser's polls - are inside
the methods that make them), so the numbers add up to the pipelined cycle.
    python tools/filter_timeline.py [N] [cycles]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from triangular_transport_toolbox_amd import entf, termtable, transport_map as T  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rng = np.random.default_rng(0)
ens = rng.standard_normal((N, 3)) * [8, 9, 8] + [0, 0, 25]
flt = entf.Filter(N, seed=0)
flt.set_ensemble(ens)
obs = np.array([1.0, 2.0, 25.0])
for _ in range(5):
    flt.forecast(); flt.assimilate(obs)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(cycles):
    flt.forecast(); flt.assimilate(obs)
torch.cuda.synchronize()
print('plain: %.3f ms per cycle' % (1e3 * (time.perf_counter() - t0) / cycles))

acc, cnt = {}, {}


def wrap(obj, name, label=None):
    f = getattr(obj, name)
    label = label or name

    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[label] = acc.get(label, 0.0) + time.perf_counter() - t
            cnt[label] = cnt.get(label, 0) + 1
    setattr(obj, name, g)


tm = flt.tm
for n in ('map_columns', 'reset_device', '_export', 'standardize', '_import', 'determine_special_term_locations', '_device_quantile',
          '_refresh_uform', 'optimize', '_optimize_separable_batch', '_gram_many', 'separable_setup', 'forward_device', '_pack_coeffs',
          '_fold_staged', 'inverse_device', '_order_statistics', '_prefetch_order_statistics', '_launch_select', '_standardize_cols',
          '_launch_default_tables'):
    wrap(tm, n)
wrap(termtable, 'place_special_terms')
wrap(tm._cm, 'fill_special_terms')
wrap(flt, 'forecast')
wrap(tm._lib, 'ttm_optimize_separable_batch', 'native chain')
wrap(tm._lib, 'ttm_basis')
t0 = time.perf_counter()
for _ in range(cycles):
    flt.forecast(); flt.assimilate(obs)
torch.cuda.synchronize()
tot = 1e3 * (time.perf_counter() - t0) / cycles
print('wrapped: %.3f ms per cycle' % tot)
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print('%-36s %8.1f us per update   %5.1f calls per update' % (k, 1e6 * v / (3 * cycles), cnt[k] / (3.0 * cycles)))
