#!/usr/bin/env python3
"""Host boundary of map() / inverse_map() at C5 (NumPy in -> NumPy out): pipelined against plain, by chunk count."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import torch

tm, X, cfg = bench.build_map('C5', 0)
N = X.shape[0]
def best(fn, n=4):
    b = 1e9; r = None
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); b = min(b, time.perf_counter() - t0)
    return 1e3 * b, r
src = torch.from_numpy(X); dev = torch.empty((N, 40), dtype=torch.float64, device='cuda')
print('pageable H2D whole %.2f ms' % best(lambda: dev.copy_(src))[0])
def chunked(nc):
    rows = -(-N // nc)
    for r0 in range(0, N, rows):
        dev[r0:r0 + rows].copy_(src[r0:r0 + rows], non_blocking=True)
for nc in (4, 8, 16):
    print('pageable H2D in %d chunks %.2f ms' % (nc, best(lambda: chunked(nc))[0]))
for nc in (2, 4, 8, 16):
    tm.PIPE_CHUNKS = nc
    ms, Z = best(lambda: tm.map(X))
    msi, Xi = best(lambda: tm.inverse_map(Z))
    print('chunks %2d: map %.2f ms  inverse_map %.2f ms' % (nc, ms, msi))
tm.host_pipeline = False
print('plain: map %.2f ms  inverse_map %.2f ms' % (best(lambda: tm.map(X), 2)[0], best(lambda: tm.inverse_map(Z), 2)[0]))
