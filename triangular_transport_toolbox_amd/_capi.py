"""
ctypes binding of the C ABI in include/ttm.h (libttm.so).

This is the thin layer north_star asks for: Python host code -> C ABI -> HIP
kernels.  Device memory is owned by the caller (PyTorch-ROCm tensors, passed
as raw pointers); no torch type crosses the boundary.
"""
import ctypes
import os

import numpy as np

from . import build as _build

c_i32, c_i64, c_dbl, c_vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_double, ctypes.c_void_p


OBJECTIVE_CB = ctypes.CFUNCTYPE(c_i32, c_i32, ctypes.POINTER(c_dbl), ctypes.POINTER(c_dbl), ctypes.POINTER(c_dbl), c_vp)


class ttm_sep_task(ctypes.Structure):
    """Mirror of `struct ttm_sep_task` (include/ttm.h: one component problem of ttm_optimize_separable_batch)."""
    _fields_ = [('dPsi', c_vp), ('ldp', c_i64), ('m', c_i32), ('rc', c_i32), ('A', c_vp), ('b', c_vp), ('lb', c_vp), ('ub', c_vp),
                ('x', c_vp), ('work', c_vp), ('counter', c_vp), ('sums_host', c_vp), ('result', c_dbl * 5),
                ('xk', c_vp), ('kinds', c_vp), ('pars', c_vp), ('armed', c_i32)]


class ttm_int_task(ctypes.Structure):
    """Mirror of `struct ttm_int_task` (include/ttm.h: one component problem of ttm_optimize_integrated_batch)."""
    _fields_ = [('k', c_i32), ('m', c_i32), ('regularization', c_i32), ('rc', c_i32), ('lam', c_vp), ('x', c_vp), ('work', c_vp),
                ('counter', c_vp), ('sums_host', c_vp), ('result', c_dbl * 5)]


class ttm_program(ctypes.Structure):
    """Mirror of `struct ttm_program` (include/ttm.h)."""
    _fields_ = [('itab', c_vp), ('ftab', c_vp), ('fdesc', c_vp), ('fints', c_vp), ('dpar', c_vp), ('quad_x', c_vp), ('quad_w', c_vp),
                ('h_comp_off', c_vp), ('h_dpar_off', c_vp), ('h_coef_off', c_vp), ('h_nslots', c_vp), ('h_n_nm', c_vp), ('h_fold_off', c_vp), ('h_ftab_off', c_vp), ('h_nb1', c_vp), ('h_complex', c_vp), ('d_offsets', c_vp),
                ('D', c_i32), ('d_cols', c_i32), ('family', c_i32), ('monotonicity', c_i32), ('rectifier', c_i32),
                ('Q', c_i32), ('plan_ways', c_i32), ('u_enabled', c_i32), ('delta', c_dbl),
                ('ucomp', c_vp), ('ugrp', c_vp), ('umono', c_vp), ('ugeo', c_vp), ('h_ucomp', c_vp), ('h_ugrp', c_vp),
                ('u_size', c_i64), ('u_err_off', c_i64), ('u_h_off', c_i64), ('u_h_cls', c_i32), ('u_h_ng', c_i32),
                ('u_p_off', c_i64), ('u_p_lag', c_i32), ('u_p_stride', c_i32)]


MONO = {'integrated rectifier': 0, 'separable monotonicity': 1}
RECT = {'exponential': 0, 'softplus': 1, 'squared': 2, 'expneg': 3, 'explinearunit': 4}

_SIGNATURES = {
    'ttm_last_error_string': (ctypes.c_char_p, []),
    'ttm_set_error_string': (ctypes.c_int, [ctypes.c_char_p]),
    'ttm_version': (ctypes.c_int, []),
    'ttm_last_kernel': (ctypes.c_char_p, []),
    'ttm_set_option': (ctypes.c_int, [ctypes.c_char_p, c_i32]),
    'ttm_reset_options': (ctypes.c_int, []),
    'ttm_program_sizeof': (c_i64, []),
    'ttm_device_count': (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    'ttm_colstats_work_size': (c_i64, [c_i64, c_i32]),
    'ttm_colstats': (ctypes.c_int, [c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp]),
    'ttm_colstats_cols': (ctypes.c_int, [c_vp, c_i64, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp]),
    'ttm_standardize_cols': (ctypes.c_int, [c_vp, c_i64, c_i64, c_i32, c_vp, c_vp, c_vp, c_i64, c_vp]),
    'ttm_import': (ctypes.c_int, [c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_i64, c_vp]),
    'ttm_export': (ctypes.c_int, [c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    'ttm_select_work_size': (c_i64, [c_i32]),
    'ttm_order_statistics': (ctypes.c_int, [c_vp, c_i64, c_vp, c_i32, c_vp, c_vp, c_vp]),
    'ttm_order_statistics_dist': (ctypes.c_int, [c_vp, c_i64, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp]),
    'ttm_fold_size': (c_i64, [ctypes.POINTER(ttm_program)]),
    'ttm_uform_offset': (c_i64, [ctypes.POINTER(ttm_program)]),
    'ttm_fold': (ctypes.c_int, [ctypes.POINTER(ttm_program), c_vp, c_vp, c_vp]),
    'ttm_fold_staged': (ctypes.c_int, [ctypes.POINTER(ttm_program), c_vp, c_vp, c_vp, c_vp, c_vp]),
    'ttm_forward': (ctypes.c_int, [ctypes.POINTER(ttm_program), c_vp, c_vp, c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_i64,
                                   c_vp, c_vp, c_vp, c_vp]),
    'ttm_basis': (ctypes.c_int, [ctypes.POINTER(ttm_program), c_i32, c_i32, c_vp, c_i64, c_i64, c_vp, c_i64, c_vp]),
    'ttm_inverse_table_build': (ctypes.c_int, [ctypes.POINTER(ttm_program), c_vp, c_vp, c_i32, c_i32, c_vp, c_i32, c_vp, c_vp]),
    'ttm_inverse_table_index': (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'ttm_inverse_table_build_index': (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'ttm_inverse_table_image_doubles': (c_i64, [ctypes.POINTER(ttm_program), c_i32, c_i32, c_i32, c_i32]),
    'ttm_setup_staged': (ctypes.c_int, [ctypes.POINTER(ttm_program), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'ttm_inverse_table': (ctypes.c_int, [ctypes.POINTER(ttm_program), c_vp, c_vp, c_i32, c_i32, c_vp, c_i64, c_vp, c_i64,
                                         c_i64, c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_i64, c_vp]),
    'ttm_inverse_bisect': (ctypes.c_int, [ctypes.POINTER(ttm_program), c_vp, c_vp, c_i32, c_i32, c_vp, c_i64, c_vp, c_i64,
                                          c_i64, c_vp, c_vp, c_vp]),
    'ttm_inverse_newton': (ctypes.c_int, [ctypes.POINTER(ttm_program), c_vp, c_vp, c_i32, c_i32, c_vp, c_i64, c_vp, c_i64,
                                          c_i64, c_vp, c_vp]),
    'ttm_reduce_work_size': (c_i64, [c_i32]),
    'ttm_objective': (ctypes.c_int, [ctypes.POINTER(ttm_program), c_i32, c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp]),
    'ttm_objective_host': (ctypes.c_int, [ctypes.POINTER(ttm_program), c_i32, c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp]),
    'ttm_objective_sep_cached': (ctypes.c_int, [c_vp, c_i64, c_i64, c_i32, c_vp, c_dbl, c_vp, c_vp, c_vp, c_vp]),
    'ttm_objective_host_marked': (ctypes.c_int, [ctypes.POINTER(ttm_program), c_i32, c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_dbl, c_vp]),
    'ttm_stream_synchronize': (ctypes.c_int, [c_vp]),
    'ttm_roundtrip': (ctypes.c_int, [ctypes.POINTER(ttm_program), c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp,
                                     c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp]),
    'ttm_mailbox_acquire': (c_vp, []),
    'ttm_mailbox_release': (None, [c_vp]),
    'ttm_objective_sep_server_start': (ctypes.c_int, [c_vp, c_i64, c_i64, c_i32, c_dbl, c_vp, c_vp, c_vp, ctypes.c_uint32, c_vp]),
    'ttm_sentinel_fill': (ctypes.c_int, [c_vp, c_i32, c_i64, c_vp]),
    'ttm_objective_sep_cached_sent': (ctypes.c_int, [c_vp, c_i64, c_i64, c_i32, c_vp, c_dbl, c_vp, c_vp, c_vp]),
    'ttm_objective_sep_direct_sent': (ctypes.c_int, [c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_dbl, c_vp, c_vp, c_vp]),
    'ttm_objective_sep_cached_marked': (ctypes.c_int, [c_vp, c_i64, c_i64, c_i32, c_vp, c_dbl, c_vp, c_vp, c_vp, c_vp, c_dbl, c_vp]),
    'ttm_objective_sep_direct_marked': (ctypes.c_int, [c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_dbl, c_vp, c_vp, c_vp, c_vp, c_dbl, c_vp]),
    'ttm_gram': (ctypes.c_int, [ctypes.POINTER(ttm_program), c_i32, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp]),
    'ttm_gram_many': (ctypes.c_int, [ctypes.POINTER(ttm_program), c_vp, c_i32, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp]),
    'ttm_lorenz63_rk4': (ctypes.c_int, [c_vp, c_i64, c_i64, c_dbl, c_i32, c_vp]),
    'ttm_perturb': (ctypes.c_int, [c_vp, c_vp, c_dbl, ctypes.c_uint64, ctypes.c_uint32, c_i64, c_i64, c_vp, c_vp]),
    'ttm_map_columns': (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_i32, c_i64, c_vp, c_i64, c_vp]),
    'ttm_signal': (ctypes.c_int, [c_vp, c_dbl, c_vp]),
    'ttm_lbfgsb_minimize': (ctypes.c_int, [c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp]),
    'ttm_optimize_separable': (ctypes.c_int, [c_vp, c_i64, c_i64, c_i32, c_vp, c_vp, c_dbl, c_dbl, c_vp, c_vp, c_vp, c_vp, c_vp,
                                              c_vp, c_vp, c_vp, c_vp, c_i32, c_vp]),
    'ttm_optimize_separable_batch': (ctypes.c_int, [ctypes.POINTER(ttm_sep_task), c_i32, c_i64, c_dbl, c_dbl, c_i32, c_vp, c_i32]),
    'ttm_separable_reduce_l2': (ctypes.c_int, [c_vp, c_i32, c_i32, c_dbl, c_vp, c_vp]),
    'ttm_optimize_integrated_batch': (ctypes.c_int, [ctypes.POINTER(ttm_program), ctypes.POINTER(ttm_int_task), c_i32, c_vp, c_i64, c_i64,
                                                     c_dbl, c_i32, c_vp, c_i32]),
    'ttm_bfgs_minimize': (ctypes.c_int, [c_i32, c_vp, c_vp, c_vp, c_i32, c_vp]),
    'ttm_optimize_integrated': (ctypes.c_int, [ctypes.POINTER(ttm_program), c_i32, c_i32, c_vp, c_i64, c_i64, c_dbl, c_i32, c_vp, c_vp,
                                               c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp]),
    'ttm_comm_last_error': (ctypes.c_char_p, []),
    'ttm_comm_unique_id': (ctypes.c_int, [c_vp]),
    'ttm_comm_create': (ctypes.c_int, [c_vp, c_i32, c_i32, ctypes.POINTER(c_vp)]),
    'ttm_comm_destroy': (ctypes.c_int, [c_vp]),
    'ttm_comm_size': (ctypes.c_int, [c_vp, ctypes.POINTER(c_i32), ctypes.POINTER(c_i32)]),
    'ttm_allreduce_f64': (ctypes.c_int, [c_vp, c_vp, c_i64, c_i32, c_vp]),
    'ttm_allreduce_i32': (ctypes.c_int, [c_vp, c_vp, c_i64, c_i32, c_vp]),
}

EXPORTED_SYMBOLS = sorted(_SIGNATURES)

_lib = None


def library_path():
    return _build.LIB


def load():
    """Load libttm.so.  A missing or out-of-date library is (re)built in-tree with hipcc first; if that is
    impossible the call raises - there is no fallback implementation."""
    global _lib
    if _lib is None:
        path = library_path()
        if _build.is_stale():
            # never run an out-of-date binary silently: a stale library whose rebuild fails is an error
            try:
                _build.build_lib()
            except Exception as exc:          # no hipcc / compile error
                raise RuntimeError('libttm.so (%s) is %s and could not be (re)built: %s'
                                   % (path, 'out of date' if os.path.exists(path) else 'not built', exc))
        # PyTorch-ROCm bundles its own HIP runtime (same SONAME as /opt/rocm's).  It has to be
        # loaded first so that libttm.so binds to that one runtime: two runtimes in a process
        # cannot share streams / allocations (and the second one finds no GPU).
        import torch  # noqa: F401
        lib = ctypes.CDLL(path)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.ttm_program_sizeof() != ctypes.sizeof(ttm_program):
            raise RuntimeError('libttm.so was built from a different include/ttm.h (struct ttm_program: %d bytes, '
                               'binding: %d)' % (lib.ttm_program_sizeof(), ctypes.sizeof(ttm_program)))
        _lib = lib
    return _lib


class TTMError(RuntimeError):
    pass


TTM_E_UNSUPPORTED = -4           # (include/ttm.h: "not for this shape / map - take the general path")


def check(rc):
    if rc != 0:
        raise TTMError('libttm error %d: %s' % (rc, load().ttm_last_error_string().decode()))


def set_option(name, value):
    """Pin a launch-planning option of the loaded library (include/ttm.h: ttm_set_option)."""
    check(load().ttm_set_option(name.encode(), int(value)))


def reset_options():
    check(load().ttm_reset_options())


def device_count():
    n = ctypes.c_int(0)
    load().ttm_device_count(ctypes.byref(n))
    return n.value


def require_device():
    """The product has no CPU path: fail loudly when no MI355X/HIP device is visible."""
    lib = load()
    n = ctypes.c_int(0)
    rc = lib.ttm_device_count(ctypes.byref(n))
    if rc != 0 or n.value < 1:
        raise RuntimeError('triangular_transport_toolbox_amd needs a HIP device (MI355X); none is visible: '
                           + lib.ttm_last_error_string().decode())
    return n.value


def make_program(cm, itab_ptr, ftab_ptr, fdesc_ptr, fints_ptr, dpar_ptr, qx_ptr, qw_ptr, offsets_ptr, Q, monotonicity,
                 rectifier, delta):
    """Fill a ttm_program from a CompiledMap and raw table pointers.  The
    returned object keeps the host offset arrays alive."""
    p = ttm_program()
    p._keep = [np.ascontiguousarray(a, dtype=np.int32) for a in (cm.comp_off, cm.dpar_off, cm.coef_off, cm.nslots, cm.n_nm, cm.fold_off, cm.ftab_off,
                                                                 cm.nb1, cm.complex)]
    p.itab, p.ftab, p.fdesc, p.fints = itab_ptr, ftab_ptr, fdesc_ptr, fints_ptr
    p.dpar, p.quad_x, p.quad_w = dpar_ptr, qx_ptr, qw_ptr
    (p.h_comp_off, p.h_dpar_off, p.h_coef_off, p.h_nslots, p.h_n_nm, p.h_fold_off, p.h_ftab_off, p.h_nb1, p.h_complex) = \
        [a.ctypes.data for a in p._keep]
    p.d_offsets = offsets_ptr
    p.D, p.d_cols, p.family = int(cm.D), int(cm.d_cols), int(cm.family)
    p.monotonicity = MONO[monotonicity.lower()]
    p.rectifier = RECT[rectifier]
    p.Q = int(Q)
    p.plan_ways = int(cm.plan_ways)
    p.delta = float(delta)
    p.u_enabled = 0
    return p


def set_uform(p, cm, ucomp_ptr, ugrp_ptr, umono_ptr, ugeo_ptr):
    """Attach / refresh the U-form tables of a program (device pointers; the host copy of ucomp is kept alive
    by the program object).  Called again whenever the spline geometry changes (termtable.uform_geometry)."""
    p._keep_u = np.ascontiguousarray(cm.ucomp, dtype=np.int32)
    p.ucomp, p.ugrp, p.umono, p.ugeo = ucomp_ptr, ugrp_ptr, umono_ptr, ugeo_ptr
    p._keep_g = np.ascontiguousarray(cm.ugrp, dtype=np.int32)
    p.h_ucomp = p._keep_u.ctypes.data
    p.h_ugrp = p._keep_g.ctypes.data
    p.u_size, p.u_err_off = int(cm.u_size), int(cm.u_err_off)
    p.u_h_off, p.u_h_cls, p.u_h_ng = int(cm.u_h_off), int(cm.u_h_cls if cm.u_enabled else 0), int(cm.u_h_ng)
    p.u_p_off, p.u_p_stride = int(cm.u_p_off), int(cm.u_p_stride)
    p.u_p_lag = int(cm.u_p_lag if (cm.u_enabled and cm.u_h_cls) else 0)
    p.u_enabled = 1 if cm.u_enabled else 0
    return p
