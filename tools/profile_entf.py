#!/usr/bin/env python3
"""Where an EnTF cycle (C4, N = 1e5, device-resident filter) spends its host time: cProfile of 50 cycles."""
import cProfile, io, os, pstats, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402

res = bench.entf_config(torch, N=100000, cycles=30)
print({k: res[k] for k in ('ms_per_cycle', 'rmse_last')})
pr = cProfile.Profile()
pr.enable()
res = bench.entf_config(torch, N=100000, cycles=50)
pr.disable()
print({k: res[k] for k in ('ms_per_cycle', 'rmse_last')})
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45)
print('\n'.join(l for l in s.getvalue().splitlines() if 'transport_map.py' in l or 'entf.py' in l or 'termtable' in l or 'method' in l or 'quantile' in l)[:6000])
