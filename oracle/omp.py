"""
CPU ORACLE, C++/OpenMP leg - TEST INFRASTRUCTURE ONLY (see oracle/ttm_oracle_omp.cpp).

Builds ``oracle/_omp/libttm_oracle_omp.so`` with g++ -fopenmp and drives it from a NumPy ``OracleMap``: the term
plans the NumPy oracle resolved from the specification (which tests/test_oracle_golden.py pins against the
reference) are flattened into the arrays the compiled routines walk.  Only ``bench.py``'s ``cpu_baseline`` leg,
``__graft_entry__.build()`` (compile only) and ``tests/`` may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'ttm_oracle_omp.cpp')
OUT = os.path.join(HERE, '_omp')
LIB = os.path.join(OUT, 'libttm_oracle_omp.so')

_KIND = {'let': 3, 'ret': 4, 'rbf': 5, 'irbf': 6}
_lib = None


def build():
    if os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    os.makedirs(OUT, exist_ok=True)
    tmp = '%s.tmp.%d' % (LIB, os.getpid())
    subprocess.run(['g++', '-O2', '-std=c++17', '-fopenmp', '-ffp-contract=off', '-fPIC', '-shared', '-o', tmp, SRC],
                   check=True)
    os.replace(tmp, LIB)
    return LIB


def lib():
    global _lib
    if _lib is None:
        l = ctypes.CDLL(build())
        vp, i32, i64, dbl = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_double
        l.ttmo_max_threads.restype = ctypes.c_int
        l.ttmo_forward.restype = ctypes.c_int
        l.ttmo_forward.argtypes = [i32, i32, i32, vp, vp, vp, vp, vp, vp, i64, vp, i32]
        l.ttmo_inverse_table.restype = ctypes.c_int
        l.ttmo_inverse_table.argtypes = [i32, i32, i32, vp, vp, vp, vp, vp, vp, i32, vp, i64, i32, i32, dbl, i32]
        _lib = l
    return _lib


def usable_cores():
    """Host cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (a container that
    sees 256 CPUs but is throttled to a 16-CPU share runs 64 busy processes at a quarter of a core each - which is
    what made one-process-per-os.cpu_count() baselines slower per process)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    info = {'os_cpu_count': os.cpu_count(), 'affinity': n, 'cgroup_quota': None}
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    info['cgroup_quota'] = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    info['cgroup_quota'] = q / float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            break
        except (OSError, ValueError, IndexError):
            continue
    if info['cgroup_quota']:
        n = max(1, min(n, int(info['cgroup_quota'] + 0.5)))
    info['usable'] = n
    return n, info


class OmpMap:
    """The compiled leg for one separable Hermite-function OracleMap (coefficients are read when a routine runs)."""

    def __init__(self, om):
        if om.monotonicity.lower() != 'separable monotonicity' or not om.hf_allowed:
            raise NotImplementedError('the OpenMP leg restates the separable Hermite-function path only')
        self.om = om
        tstart, foff, fi, fd = [0], [0], [], []
        for k in range(om.D):
            kc = k + om.skip_dimensions
            for plans in (om.plan_nonmon[k], om.plan_mon[k]):
                for term in plans:
                    for f in term:
                        if f[0] == 'const':
                            fi += [0, 0, 0]
                            fd += [0.0, 0.0]
                        elif f[0] == 'poly':
                            _, var, order, hf, lin = f
                            if lin:
                                raise NotImplementedError("'LIN' factors")
                            fi += [2 if hf else 1, var, order]
                            fd += [om._hfconst(order) if hf else 1.0, 0.0]
                        else:
                            _, kind, var, cross, index = f
                            mu, sc = om._st_params(kc, var, cross, index)
                            fi += [_KIND[kind], var, 0]
                            fd += [float(mu), float(sc)]
                    foff.append(len(fi) // 3)
                tstart.append(len(foff) - 1)
        self.tstart = np.asarray(tstart, dtype=np.int32)
        self.foff = np.asarray(foff, dtype=np.int32)
        self.fi = np.asarray(fi, dtype=np.int32)
        self.fd = np.asarray(fd, dtype=np.float64)
        self.d = om.skip_dimensions + om.D

    def _coef(self):
        om = self.om
        parts = []
        for k in range(om.D):
            parts += [np.asarray(om.coeffs_nonmon[k], dtype=float).ravel(), np.asarray(om.coeffs_mon[k], dtype=float).ravel()]
        c = np.ascontiguousarray(np.concatenate(parts))
        assert len(c) == len(self.foff) - 1
        return c

    @staticmethod
    def _p(a):
        return ctypes.c_void_p(a.ctypes.data)

    def forward_std(self, Xs, threads):
        """Z = S(x) for standardised row-major samples."""
        Xs = np.ascontiguousarray(Xs, dtype=np.float64)
        N = Xs.shape[0]
        Z = np.empty((N, self.om.D))
        c = self._coef()
        rc = lib().ttmo_forward(self.om.D, self.d, self.om.skip_dimensions, self._p(self.tstart), self._p(self.foff), self._p(self.fi),
                                self._p(self.fd), self._p(c), self._p(Xs), N, self._p(Z), int(threads))
        assert rc == 0
        return Z

    def inverse_std(self, Z, threads, Xs_cond=None, k0=0):
        """Standardised X = S^{-1}(z) (table inverse); Xs_cond: N x d with the conditioning columns filled in."""
        Z = np.ascontiguousarray(Z, dtype=np.float64)
        N = Z.shape[0]
        X = np.zeros((N, self.d)) if Xs_cond is None else np.ascontiguousarray(Xs_cond, dtype=np.float64).copy()
        c = self._coef()
        rc = lib().ttmo_inverse_table(self.om.D, self.d, self.om.skip_dimensions, self._p(self.tstart), self._p(self.foff),
                                      self._p(self.fi), self._p(self.fd), self._p(c), self._p(Z), int(k0), self._p(X), N,
                                      1 if self.om.root_search_truncation else 0, 1001, 10.0, int(threads))
        assert rc == 0
        return X

    # the reference's public semantics (raw samples in / out), for the parity tests
    def map(self, X, threads=1):
        om = self.om
        Xs = np.array(X, dtype=float, copy=True)
        if om.standardize_samples:
            Xs -= om.X_mean
            Xs /= om.X_std
        return self.forward_std(Xs, threads)

    def inverse_map(self, Z, threads=1):
        om = self.om
        X = self.inverse_std(Z, threads)
        if om.standardize_samples:
            X *= om.X_std
            X += om.X_mean
        return X[:, om.skip_dimensions:]
