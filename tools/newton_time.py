import sys, time
sys.path.insert(0,'/root/repo')
import numpy as np, torch, bench
for rf in ('reference','newton'):
    tm, X, cfg = bench.build_map('C2a', 0, root_finder=rf)
    N, D, d = tm._N, tm.D, tm._cm.d_cols
    coef = tm._pack_coeffs(); Xs = tm._Xs; Z = tm._cols(D, N); Xinv = tm._cols(d, N, zero=True)
    tm.forward_device(Xs, N, coef=coef, Z=Z)
    for _ in range(3): tm.inverse_device(Z, N, coef=coef, X=Xinv)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(5): tm.inverse_device(Z, N, coef=coef, X=Xinv)
    torch.cuda.synchronize(); t=(time.perf_counter()-t0)/5
    err=(Xinv[:, :N]-Xs[:, :N]).abs()
    print(rf, 'inverse ms', 1e3*t, 'roundtrip max', float(err.max()), 'median', float(err.median()))
