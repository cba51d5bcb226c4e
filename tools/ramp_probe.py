import sys, time, json
sys.path.insert(0,'/root/repo')
import numpy as np, torch
import bench
tm, X, cfg = bench.build_map('C5', 0)
N, D, d = tm._N, tm.D, tm._cm.d_cols
coef = tm._pack_coeffs(); Xs = tm._Xs; Z = tm._cols(D, N); Xinv = tm._cols(d, N, zero=True)
def step():
    tm.forward_device(Xs, N, coef=coef, Z=Z); tm.inverse_device(Z, N, coef=coef, X=Xinv)
for _ in range(2000): step()
torch.cuda.synchronize()
for trial, gap in enumerate([0.0, 0.0, 0.001, 0.01, 0.1]):
    time.sleep(gap)
    K=60
    ev=[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    t0=time.perf_counter()
    for a,b in ev:
        a.record(); tm.forward_device(Xs, N, coef=coef, Z=Z); b.record(); tm.inverse_device(Z, N, coef=coef, X=Xinv)
    t1=time.perf_counter()
    torch.cuda.synchronize()
    t2=time.perf_counter()
    ts=[a.elapsed_time(b) for a,b in ev]
    print('gap',gap,'enqueue ms',round(1e3*(t1-t0),3),'total ms',round(1e3*(t2-t0),3),'fwd first10',[round(x,3) for x in ts[:10]],'mid',[round(x,3) for x in ts[25:30]],'last',[round(x,3) for x in ts[-5:]])
