import os, sys, ctypes, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tests.test_random_few as t
from tests.test_uform import _synthetic_separable
from triangular_transport_toolbox_amd.transport_map import transport_map
lib = t._lib()
rng = np.random.default_rng(327)
D, skip, mon, non, family = t.random_spec(rng)
d = D + skip
n = int(rng.choice([257, 2049, 5003]))
X = rng.standard_normal((n, d)) @ (np.tril(rng.standard_normal((d, d)) * 0.4) + np.eye(d)).T + 0.3 * rng.standard_normal((n, d)) ** 2
tm = transport_map(X=X, monotone=mon, nonmonotone=non, verbose=False, monotonicity='separable monotonicity', polynomial_type=family)
for k in range(D):
    tm.coeffs_mon[k] = 0.2 + 0.5 * rng.random(len(tm.coeffs_mon[k]))
    tm.coeffs_nonmon[k] = 0.3 * rng.standard_normal(len(tm.coeffs_nonmon[k])) / (1 + np.arange(len(tm.coeffs_nonmon[k])))


def unwritten(tm, n, label):
    for ul, b in ((1, 0), (0, 0), (1, 1)):
        lib.ttm_reset_options(); lib.ttm_set_option(b'u_loader', ul); lib.ttm_set_option(b'band_fwd', b)
        Z = tm._cols(tm.D, n)
        Z.fill_(float('nan'))
        tm.forward_device(tm._Xs, n, Z=Z)
        kname = lib.ttm_last_kernel().decode()
        bad = torch.isnan(Z[:, :n]).any(0).nonzero().flatten().cpu().numpy()
        print(label, 'u_loader', ul, 'band', b, kname, 'rows left NaN:', len(bad), bad[:12])


unwritten(tm, n, 'seed327 D=2 skip=1 n=%d' % n)
rng = np.random.default_rng(0)
for D2 in (2, 5, 6):
    for n2 in (5003, 70001):
        mon, non = _synthetic_separable(D2, 2, 3, 1, 2)
        X = rng.standard_normal((n2, D2))
        tm2 = transport_map(X=X, monotone=mon, nonmonotone=non, verbose=False, monotonicity='separable monotonicity')
        for k in range(D2):
            tm2.coeffs_mon[k] = 0.2 + 0.5 * rng.random(len(tm2.coeffs_mon[k]))
            tm2.coeffs_nonmon[k] = 0.3 * rng.standard_normal(len(tm2.coeffs_nonmon[k]))
        unwritten(tm2, n2, 'synthetic D=%d n=%d' % (D2, n2))
print('----')
lib.ttm_reset_options(); lib.ttm_set_option(b'u_loader', 1); lib.ttm_set_option(b'band_fwd', 0)
Z = tm._cols(tm.D, n)
ld = Z.shape[1]
print('Z ptr', hex(Z.data_ptr()), 'Xs ptr', hex(tm._Xs.data_ptr()), 'Xs bytes', tm._Xs.numel() * 8, 'ld', ld, 'gap', Z.data_ptr() - tm._Xs.data_ptr())
def run(fill):
    Z.fill_(7.0); fill(Z); tm.forward_device(tm._Xs, n, Z=Z); torch.cuda.synchronize()
    Zh = Z[:, :n]
    return torch.isnan(Zh).any(0).nonzero().flatten().cpu().numpy()
print('all NaN ->', run(lambda Z: Z.fill_(float('nan'))))
for c in range(tm.D):
    print('only comp', c, 'NaN ->', run(lambda Z: Z[c].fill_(float('nan'))))
print('only padding (cols >= n) NaN ->', run(lambda Z: Z[:, n:].fill_(float('nan'))))
lo, hi = 0, ld
c = 0
import functools
def f(a, b, Z): Z[:, a:b] = float('nan')
while hi - lo > 1:
    mid = (lo + hi) // 2
    r = run(functools.partial(f, lo, mid))
    if len(r): hi = mid
    else: lo = mid
print('smallest NaN column range that leaks:', lo, hi, '->', run(functools.partial(f, lo, hi)))
