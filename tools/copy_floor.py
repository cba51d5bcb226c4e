#!/usr/bin/env python3
"""What a plain device copy achieves on buffers of the C5 step's shape (40 x 1e6 fp64), alternating X -> Z, Z -> Xi as the
bench's forward + inverse do: the memory floor of a step on this card."""
import json
import time
import torch
N, D = 1000000, 40
X = torch.randn(D, N, dtype=torch.float64, device='cuda')
Z = torch.empty_like(X); Xi = torch.empty_like(X)


def run(fn, n=100):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t_end = time.time() + 1.0
    while time.time() < t_end:
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def step():
    Z.copy_(X); Xi.copy_(Z)


def step_ew():
    torch.abs(X, out=Z); torch.abs(Z, out=Xi)


out = {'alternating_pair_ms': run(step), 'same_buffer_copy_ms': run(lambda: Z.copy_(X)),
       'alternating_pair_abs_ms': run(step_ew), 'single_abs_ms': run(lambda: torch.abs(X, out=Z)),
       'single_neg_inplace_ms': run(lambda: torch.neg_(Z))}
out['pair_abs_TBps'] = 4 * X.numel() * 8 / (out['alternating_pair_abs_ms'] * 1e-3) / 1e12
out['single_abs_TBps'] = 2 * X.numel() * 8 / (out['single_abs_ms'] * 1e-3) / 1e12
out['pair_TBps'] = 4 * X.numel() * 8 / (out['alternating_pair_ms'] * 1e-3) / 1e12
out['single_TBps'] = 2 * X.numel() * 8 / (out['same_buffer_copy_ms'] * 1e-3) / 1e12
print(json.dumps(out))
