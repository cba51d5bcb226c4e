"""
TEST INFRASTRUCTURE: a host test double of libttm.so.

tests/hostemu/ttm_hostemu.cpp exports the C ABI of include/ttm.h with host
pointers (the kernels' per-sample bodies compiled for the CPU).  `install()`
injects it into the product's ctypes layer and points the `transport_map` class
at torch CPU tensors, so the host logic can be tested without a GPU.  The
product never does this by itself: without an injected double it loads
libttm.so and requires a HIP device.
"""
import contextlib
import ctypes
import os
import subprocess

import numpy as np

from triangular_transport_toolbox_amd import _capi, termtable

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'ttm_hostemu.cpp')
# (TTM_HOSTEMU_LIB: a build of the test double made elsewhere - the sanitizer builds of tools/sanitize_hostemu.sh - used as it is)
LIB = os.environ.get('TTM_HOSTEMU_LIB') or os.path.join(HERE, 'libttm_hostemu.so')
CSRC = os.path.join(HERE, '..', '..', 'triangular_transport_toolbox_amd', 'csrc')
DEPS = [SRC, os.path.join(CSRC, 'ttm_eval.h'), os.path.join(CSRC, 'ttm_math.h'), os.path.join(CSRC, 'ttm_vec.h'),
        os.path.join(CSRC, 'ttm_erf_table.h'), os.path.join(CSRC, 'ttm_dense.h'), os.path.join(CSRC, 'ttm_xprog.h'), os.path.join(CSRC, 'ttm_dense_table.h'), os.path.join(CSRC, 'ttm_uform.h'), os.path.join(CSRC, 'ttm_cheb_table.h'), os.path.join(CSRC, 'ttm_lbfgsb.h'), os.path.join(CSRC, 'ttm_bfgs.h'), os.path.join(CSRC, 'ttm_rng.h'), os.path.join(CSRC, 'ttm_optim.cpp'),
        os.path.join(HERE, '..', '..', 'include', 'ttm.h')]

_lib = None

_ALLREDUCE_CB = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32)


@_ALLREDUCE_CB
def _allreduce_cb(buf, count, is_f64, op):
    import torch
    import torch.distributed as dist
    ctype = ctypes.c_double if is_f64 else ctypes.c_int32
    arr = np.ctypeslib.as_array(ctypes.cast(buf, ctypes.POINTER(ctype)), shape=(int(count),))
    dist.all_reduce(torch.from_numpy(arr), op=dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX)


def build():
    if os.environ.get('TTM_HOSTEMU_LIB'):
        return LIB
    if os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in DEPS):
        return LIB
    tmp = '%s.tmp.%d' % (LIB, os.getpid())
    subprocess.run(['g++', '-O2', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared', '-pthread', '-o', tmp, SRC], check=True)
    os.replace(tmp, LIB)
    return LIB


def lib():
    global _lib
    if _lib is None:
        l = ctypes.CDLL(build())
        for name, (res, args) in _capi._SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        # the collective: performed by torch.distributed (gloo) on the host buffer the double hands over
        l.ttm_hostemu_set_allreduce.restype = None
        l.ttm_hostemu_set_allreduce.argtypes = [_ALLREDUCE_CB]
        l.ttm_hostemu_set_allreduce(_allreduce_cb)
        l.ttm_hostemu_allreduce_calls.restype = ctypes.c_int64
        _lib = l
    return _lib


@contextlib.contextmanager
def install():
    """Route the transport_map class through the host test double."""
    from triangular_transport_toolbox_amd.transport_map import transport_map
    saved = (_capi._lib, transport_map._DEVICE)
    _capi._lib = lib()
    transport_map._DEVICE = 'cpu'
    try:
        yield
    finally:
        _capi._lib, transport_map._DEVICE = saved


def ptr(a):
    return ctypes.c_void_p(a.ctypes.data) if a is not None else None


class EmuMap:
    """Low-level driver: host-pointer program + helpers taking standardised N x d samples."""

    def __init__(self, cm, monotonicity, rectifier='exponential', delta=1e-8, quad_order=100):
        self.cm = cm
        self.itab = np.ascontiguousarray(cm.itab)
        self.dpar = np.ascontiguousarray(cm.dpar)
        self.ftab = np.ascontiguousarray(cm.ftab)
        self.offsets = np.ascontiguousarray(cm.offsets)
        self.fdesc = np.ascontiguousarray(cm.fdesc)
        self.fints = np.ascontiguousarray(cm.fints)
        self.qx, self.qw = termtable.gauss_legendre(quad_order)
        self.qx, self.qw = np.ascontiguousarray(self.qx), np.ascontiguousarray(self.qw)
        self.prog = _capi.make_program(cm, self.itab.ctypes.data, self.ftab.ctypes.data, self.fdesc.ctypes.data,
                                       self.fints.ctypes.data, self.dpar.ctypes.data, self.qx.ctypes.data,
                                       self.qw.ctypes.data, self.offsets.ctypes.data, len(self.qx), monotonicity, rectifier, delta)
        self.pp = ctypes.byref(self.prog)

    def pack(self, coeffs_nonmon, coeffs_mon):
        coef = np.ascontiguousarray(np.concatenate([np.concatenate((np.asarray(n, float), np.asarray(m, float)))
                                                    for n, m in zip(coeffs_nonmon, coeffs_mon)]))
        self.attach_uform()
        self.fold = np.zeros(int(lib().ttm_fold_size(self.pp)))
        rc = lib().ttm_fold(self.pp, ptr(coef), ptr(self.fold), None)
        assert rc == 0
        return coef

    def attach_uform(self):
        """U-form tables for the current special-term constants (what transport_map._refresh_uform does)."""
        cm = self.cm
        if not cm.u_static:
            return
        termtable.uform_geometry(cm)
        if getattr(self, 'no_uform', False):
            cm.u_enabled = False
        self.ucomp = np.ascontiguousarray(cm.ucomp, dtype=np.int32)
        self.ugrp = np.ascontiguousarray(cm.ugrp, dtype=np.int32)
        self.umono = np.ascontiguousarray(cm.umono)
        self.ugeo = np.ascontiguousarray(cm.ugeo)
        _capi.set_uform(self.prog, cm, self.ucomp.ctypes.data, self.ugrp.ctypes.data, self.umono.ctypes.data,
                        self.ugeo.ctypes.data)
        self.prog.h_ucomp = self.ucomp.ctypes.data
        self.prog.h_ugrp = self.ugrp.ctypes.data

    def uform_errors(self):
        off = int(lib().ttm_uform_offset(self.pp))
        if off < 0:
            return None
        off += int(self.cm.u_err_off)
        return self.fold[off:off + 2 * self.cm.D].reshape(-1, 2).copy()

    @staticmethod
    def soa(X):
        return np.ascontiguousarray(np.asarray(X, dtype=float).T)

    def forward(self, coef, Xs, k0=0, k1=None, sigma=None):
        k1 = self.cm.D if k1 is None else k1
        X = self.soa(Xs)
        N = X.shape[1]
        Z = np.zeros((k1 - k0, N))
        ld = np.zeros(N)
        sg = None if sigma is None else np.ascontiguousarray(sigma, dtype=float)
        lib().ttm_forward(self.pp, ptr(coef), ptr(self.fold), ptr(X), N, N, k0, k1, ptr(Z), N, ptr(ld), ptr(sg), None, None)
        return Z.T.copy(), ld

    def basis(self, k, which, Xs):
        X = self.soa(Xs)
        N = X.shape[1]
        m = int(self.cm.n_nm[k] if which == 0 else self.cm.n_mon[k])
        out = np.zeros((max(m, 1), N))
        lib().ttm_basis(self.pp, k, which, ptr(X), N, N, ptr(out), N, None)
        return out[:m].T.copy()

    def objective(self, k, coef_k, Xs, separable):
        X = self.soa(Xs)
        N = X.shape[1]
        nacc = 1 + int(self.cm.n_mon[k]) + (0 if separable else int(self.cm.n_nm[k]))
        out = np.zeros(nacc)
        ck = np.ascontiguousarray(coef_k, dtype=float)
        lib().ttm_objective(self.pp, k, ptr(ck), ptr(X), N, N, None, ptr(out), None)
        return out

    def gram(self, k, Xs):
        X = self.soa(Xs)
        N = X.shape[1]
        m = int(self.cm.n_nm[k] + self.cm.n_mon[k])
        out = np.zeros(m * m)
        lib().ttm_gram(self.pp, k, ptr(X), N, N, None, ptr(out), None)
        return out.reshape(m, m)

    def table_build(self, coef, k, pts):
        out = np.zeros(len(pts))
        pts = np.ascontiguousarray(pts)
        lib().ttm_inverse_table_build(self.pp, ptr(coef), ptr(self.fold), k, k + 1, ptr(pts), len(pts), ptr(out), None)
        return out

    def inverse_table(self, coef, k0, k1, Z, Xinit, tab_x, tab_y, tmin, tmax, truncate=True, nb=64):
        X = self.soa(Xinit)
        Zs = self.soa(Z)
        N = X.shape[1]
        tab_x, tab_y = np.ascontiguousarray(tab_x), np.ascontiguousarray(tab_y)
        tmin, tmax = np.ascontiguousarray(tmin), np.ascontiguousarray(tmax)
        T = tab_x.shape[1]
        edges = tmin[:, None] + np.arange(nb + 1)[None, :] * ((tmax - tmin) / nb)[:, None]
        bkt = np.stack([np.searchsorted(tab_x[i], edges[i], side='left') for i in range(k1 - k0)]).astype(np.int32)
        bkt[:, 0], bkt[:, -1] = 0, T
        lib().ttm_inverse_table(self.pp, ptr(coef), ptr(self.fold), k0, k1, ptr(Zs), N, ptr(X), N, N, ptr(tab_x), ptr(tab_y),
                                T, T, None, ptr(tmin), ptr(tmax), ptr(bkt), nb, int(truncate), None, 0, None)
        return X.T.copy()

    def inverse_bisect(self, coef, k0, k1, Z, Xinit, cap=None):
        X = self.soa(Xinit)
        Zs = self.soa(Z)
        N = X.shape[1]
        iters = np.zeros(k1 - k0, dtype=np.int32)
        capa = None if cap is None else np.ascontiguousarray(cap, dtype=np.int32)
        lib().ttm_inverse_bisect(self.pp, ptr(coef), ptr(self.fold), k0, k1, ptr(Zs), N, ptr(X), N, N, ptr(iters), ptr(capa), None)
        return X.T.copy(), iters
