#!/usr/bin/env python3
"""One-off sweep of the random-specification parity tests (tests/test_random_maps.py) over many seeds on the GPU."""
import os
import sys
import traceback
import warnings

warnings.filterwarnings('ignore')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tests.test_random_maps as t  # noqa: E402

lo, hi = int(sys.argv[1]), int(sys.argv[2])
fails = 0
for seed in range(lo, hi):
    for fn in (t.test_random_separable_maps, t.test_random_integrated_maps):
        try:
            fn('hip', seed)
        except Exception as e:       # noqa: BLE001
            fails += 1
            fr = [f for f in traceback.extract_tb(e.__traceback__) if 'test_random_maps' in f.filename][-1]
            print(fn.__name__, seed, type(e).__name__, 'line', fr.lineno, (fr.line or '')[:100], flush=True)
print('seeds', lo, hi, 'fails', fails)
