#!/usr/bin/env python3
"""What the host boundary of map() / inverse_map() can cost: PCIe and host-copy rates of this box for a 320 MB fp64 matrix
(pageable and pinned copies, page-locking in place, multi-threaded host copies)."""
import ctypes
import os
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch


def t(fn, n=3):
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


N, d = 1000000, 40
X = np.random.default_rng(0).standard_normal((N, d))
mb = X.nbytes / 1e6
dev = torch.empty((N, d), dtype=torch.float64, device='cuda')
pin = torch.empty((N, d), dtype=torch.float64).pin_memory()
print('cpus', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
print('pageable H2D   %.1f ms' % (1e3 * t(lambda: dev.copy_(torch.from_numpy(X)))))
print('pinned   H2D   %.1f ms' % (1e3 * t(lambda: dev.copy_(pin, non_blocking=True))))
print('pinned   D2H   %.1f ms' % (1e3 * t(lambda: pin.copy_(dev, non_blocking=True))))
print('pageable D2H   %.1f ms' % (1e3 * t(lambda: dev.cpu())))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
dev2 = torch.empty_like(dev); pin2 = torch.empty((N, d), dtype=torch.float64).pin_memory()
def duplex():
    with torch.cuda.stream(s1):
        dev.copy_(pin, non_blocking=True)
    with torch.cuda.stream(s2):
        pin2.copy_(dev2, non_blocking=True)
print('pinned duplex  %.1f ms (H2D and D2H of %d MB each at once)' % (1e3 * t(duplex), mb))
print('host memcpy 1 thread  %.1f ms' % (1e3 * t(lambda: np.copyto(pin.numpy(), X))))
for nt in (2, 4, 8, 16):
    pool = ThreadPoolExecutor(nt)
    rows = np.array_split(np.arange(N), nt)
    def par():
        list(pool.map(lambda r: np.copyto(pin.numpy()[r[0]:r[-1] + 1], X[r[0]:r[-1] + 1]), rows))
    print('host memcpy %2d threads %.1f ms' % (nt, 1e3 * t(par)))
rt = torch.cuda.cudart()
def reg():
    rc = rt.cudaHostRegister(X.ctypes.data, X.nbytes, 0)
    assert int(rc) == 0, rc
def unreg():
    rt.cudaHostUnregister(X.ctypes.data)
t0 = time.perf_counter(); reg(); t1 = time.perf_counter()
print('hostRegister %d MB %.1f ms' % (mb, 1e3 * (t1 - t0)))
print('registered H2D %.1f ms' % (1e3 * t(lambda: dev.copy_(torch.from_numpy(X), non_blocking=True))))
t0 = time.perf_counter(); unreg(); t1 = time.perf_counter()
print('hostUnregister %.1f ms' % (1e3 * (t1 - t0)))
t0 = time.perf_counter(); reg(); t1 = time.perf_counter()
print('hostRegister again %.1f ms' % (1e3 * (t1 - t0)))
unreg()
t0 = time.perf_counter(); p3 = torch.empty((N, d), dtype=torch.float64).pin_memory(); t1 = time.perf_counter()
print('fresh pinned allocation of %d MB (torch, copy of pageable zeros) %.1f ms' % (mb, 1e3 * (t1 - t0)))
t0 = time.perf_counter(); p4 = torch.empty((N, d), dtype=torch.float64, pin_memory=True); t1 = time.perf_counter()
print('torch.empty(pin_memory=True) %.1f ms' % (1e3 * (t1 - t0)))
del p4
t0 = time.perf_counter(); p5 = torch.empty((N, d), dtype=torch.float64, pin_memory=True); t1 = time.perf_counter()
print('torch.empty(pin_memory=True) after a free (cached) %.1f ms' % (1e3 * (t1 - t0)))
