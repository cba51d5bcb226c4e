#!/usr/bin/env python3
"""tests/test_random_few.py over a range of seeds (GPU): random banded maps of a few components through k_band_few /
k_band_few_inverse / the density variant against the oracle.
    python tools/fuzz_few.py SEED_LO SEED_HI"""
import os, sys, traceback, warnings
warnings.filterwarnings('ignore')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_random_few import one  # noqa: E402

lo, hi = int(sys.argv[1]), int(sys.argv[2])
fails, seen = 0, {}
for seed in range(lo, hi):
    try:
        banded, kf, ki, info = one(seed)
        key = (kf, ki, 'skip' if info[1] else 'noskip', info[2], info[4] != 'hermite function')
        seen[key] = seen.get(key, 0) + 1
    except Exception as e:       # noqa: BLE001
        fails += 1
        fr = traceback.extract_tb(e.__traceback__)[-1]
        print('seed', seed, type(e).__name__, str(e)[:120], 'line', fr.lineno, flush=True)
print('seeds', lo, hi, 'fails', fails, 'kernels', seen)
