"""
MI355X-native drop-in for the reference class ``transport_map``
(MaxRamgraber/Triangular-Transport-Toolbox, transport_map.py - "TM" below).

Same constructor keywords, methods, attributes and error behaviour as the
reference for the hot path (TM:12-39 ctor, TM:710 reset, TM:2391 map, TM:2439 s,
TM:2569/2646 densities, TM:2714 optimize, TM:3300/3435 objective, TM:3639
inverse_map).  The host code stays Python; every per-sample computation runs in
the HIP kernels of libttm.so through the C ABI of include/ttm.h (ctypes,
``_capi``).  PyTorch-ROCm is used for device memory, streams and (optionally)
``torch.distributed`` only - no torch op touches sample data.

There is no CPU path: constructing a map without a visible HIP device or
without libttm.so raises.

What is deliberately different from the reference:
* no source generation / ``exec``: the specification lists are compiled into
  term tables (``termtable.compile_map``);
* ``workers`` is accepted and ignored (the process pool of TM:2789-2874 is
  replaced by the GPU; components / samples shard over ranks instead);
* ``adaptation`` (``adapt_map``: 'separable' TM:373-636, 'cross-terms'
  TM:4575-4950) re-specifies the map on the resident ensemble instead of
  regenerating source; ``linearization`` / ``'LIN'`` are supported for the
  integrated rectifier (see ``_linearization_thresholds``), with separable
  monotonicity they raise (reference defect TM:2063-2080);
* reference defects that are only reachable through invalid specifications are
  rejected instead of replicated (see ``termtable.compile_map``).
Reference quirks on valid inputs are reproduced: the
sample-0 bisection guard, the un-standardised derivative in the densities, the
missing 1/N of the L2-regularised separable objective, ``map(X)`` ignoring X
when ``standardize_samples`` is False.
"""

import copy
import ctypes
import os

import numpy as np

from . import _capi, comm, lbfgsb, quantile, termtable

__all__ = ['transport_map']


def _torch():
    import torch
    return torch


_COPY_POOL = None


def _parallel_copy(dst, src, threads=8):
    """dst[...] = src for two host matrices with the same number of rows, split over a few threads (NumPy's copy releases the
    GIL; one thread moves ~23 GB/s on the GPU boxes, eight ~110 GB/s)."""
    global _COPY_POOL
    n = dst.shape[0]
    if n < 8192:
        np.copyto(dst, src)
        return
    if _COPY_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        try:
            usable = len(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            usable = os.cpu_count() or 1
        _COPY_POOL = ThreadPoolExecutor(max(1, min(threads, usable)))
    parts = _COPY_POOL._max_workers
    step = -(-n // parts)
    list(_COPY_POOL.map(lambda r0: np.copyto(dst[r0:r0 + step], src[r0:r0 + step]), range(0, n, step)))


HOSTCOEF_MAX = 128           # coefficients of a component that travel as kernel arguments (csrc/ttm_dev.h: TTM_HOSTCOEF_MAX)

class _Result:
    """What an optimiser loop returns (the fields of scipy's OptimizeResult the class reads)."""
    pass


class transport_map():

    # the device the class allocates on; tests that inject the host test double
    # of the C ABI (tests/hostemu) set this to 'cpu'
    _DEVICE = 'cuda'

    def __init__(self,
                 X,
                 monotone=None,
                 nonmonotone=None,
                 polynomial_type='hermite function',
                 monotonicity='integrated rectifier',
                 standardize_samples=True,
                 standardization='standard',
                 workers=1,
                 ST_scale_factor=1.0,
                 ST_scale_mode='dynamic',
                 coeffs_init=0.,
                 alternate_root_finding=True,
                 root_search_truncation=True,
                 verbose=True,
                 linearization=None,
                 linearization_specified_as_quantiles=True,
                 linearization_increment=1E-6,
                 regularization=None,
                 regularization_lambda=0.1,
                 quadrature_input={},
                 rectifier_type='exponential',
                 delta=1E-8,
                 adaptation=False,
                 adaptation_map_type="cross-terms",
                 adaptation_max_order=10,
                 adaptation_skip_dimensions=0,
                 adaptation_max_iterations=25,
                 shard_samples=False,
                 shard_components=False,
                 root_finder='reference'):

        self._lib = _capi.load()
        if self._DEVICE == 'cuda':
            _capi.require_device()
        torch = _torch()
        self._dev = torch.device(self._DEVICE, torch.cuda.current_device()) if self._DEVICE == 'cuda' \
            else torch.device(self._DEVICE)

        if adaptation and adaptation_map_type.lower() not in ('separable', 'cross-terms'):
            raise Exception("Currently, only adaptation_map_type = 'cross-terms' is implemented.")          # (TM:648, sic)
        if linearization is not None and monotonicity.lower() == 'separable monotonicity':
            # TM:2063-2080: with a linearisation the reference's derivative functions (the only consumers in
            # separable mode) overwrite their input with min(x, lower threshold) in every column (inverted masks)
            raise NotImplementedError('linearization with separable monotonicity is a defect of the reference '
                                      '(TM:2063-2080) and is not reproduced')
        self.adaptation_map_type = adaptation_map_type.lower()
        self.adaptation_max_order = adaptation_max_order
        self.adaptation_skip_dimensions = adaptation_skip_dimensions
        self.adaptation_max_iterations = adaptation_max_iterations
        if adaptation:
            # TM:327-340: a dummy map (one constant term in both lists of every component) until adapt_map() runs
            n_comp = np.asarray(X).shape[-1] - adaptation_skip_dimensions
            monotone = [[[]] for _ in range(n_comp)]
            nonmonotone = [[[]] for _ in range(n_comp)]
        if monotone is None or nonmonotone is None:
            raise ValueError("'monotone' and 'nonmonotone' must be specified (or adaptation = True)")

        self.monotone = copy.deepcopy(monotone)
        self.nonmonotone = copy.deepcopy(nonmonotone)
        self.workers = workers
        self.rectifier_type = rectifier_type
        self.delta = delta
        if rectifier_type not in _capi.RECT:
            raise ValueError("rectifier_type '" + str(rectifier_type) + "' not understood")

        # quadrature rule, TM:196-225 (the user's dict is copied, not mutated)
        self.quadrature_input = dict(quadrature_input)
        if 'xis' not in self.quadrature_input and 'Ws' not in self.quadrature_input:
            order = self.quadrature_input.get('order', 100)
            xis, Ws = termtable.gauss_legendre(order)
            self.quadrature_input['xis'] = copy.copy(xis)
            self.quadrature_input['Ws'] = copy.copy(Ws)
        if self.quadrature_input.get('adaptive', False):
            raise NotImplementedError('adaptive quadrature (TM:4322-4353) is not supported')

        self.ST_scale_factor = ST_scale_factor
        self.ST_scale_mode = ST_scale_mode
        if self.ST_scale_mode not in ['dynamic', 'static']:
            raise ValueError("'ST_scale_mode' must be either 'dynamic' or 'static'.")
        self.standardization = standardization
        self.coeffs_init = coeffs_init
        self.alternate_root_finding = alternate_root_finding
        self.root_search_truncation = root_search_truncation
        self.verbose = verbose
        self.regularization = regularization
        self.regularization_lambda = regularization_lambda
        self.linearization = linearization
        self.linearization_specified_as_quantiles = linearization_specified_as_quantiles
        self.linearization_increment = linearization_increment
        self.monotonicity = monotonicity
        if self.monotonicity.lower() not in ['integrated rectifier', 'separable monotonicity']:
            raise ValueError("'monotonicity' type " + str(self.monotonicity) + " not understood. " +
                             "Must be either 'integrated rectifier' or 'separable monotonicity'.")
        self.polynomial_type = polynomial_type
        if polynomial_type.lower() not in termtable.FAMILIES:
            raise Exception("Polynomial type not understood. The variable polynomial_type should be either "
                            "'power series', 'hermite', 'hermite_e', 'chebyshev', 'laguerre', or 'legendre'.")
        if polynomial_type.lower() in ('hermite function', 'hermite_function', 'hermite functions'):
            self.polynomial_type = 'hermite function'
        self.standardize_samples = standardize_samples
        self.adaptation = adaptation
        self.shard_samples = bool(shard_samples)
        self._pack_shape = None
        self.shard_components = bool(shard_components)
        if root_finder not in ('reference', 'newton'):
            raise ValueError("root_finder must be 'reference' (TM:3798-3985, the default) or 'newton'")
        self.root_finder = root_finder
        self.objective_total = None

        X = np.asarray(X)
        if X.ndim != 2:
            raise Exception('X should be a two-dimensional array of shape (N,D), N = number of samples, '
                            'D = number of dimensions. Current shape of X is ' + str(X.shape))
        self.D = len(monotone)
        self.skip_dimensions = X.shape[-1] - self.D
        self._build_program(X.shape[-1])

        # ---- samples: upload, standardise, place special terms -----------------
        self._load_samples(X)

    def _build_program(self, d_cols):
        """Compile self.monotone / self.nonmonotone into term tables and the device program (what the reference's
        function_constructor_alternative does with generated source, TM:1263-2134); coefficients back to coeffs_init."""
        torch = _torch()
        # ---- compile the specification into term tables -----------------------
        self._cm = termtable.compile_map(self.monotone, self.nonmonotone, d_cols, self.polynomial_type,
                                         self.monotonicity, linearization=self.linearization)
        self.special_terms = termtable.count_special_terms(self.monotone, self.nonmonotone, self.skip_dimensions)
        self.coeffs_mon = [np.ones(int(n)) * self.coeffs_init for n in self._cm.n_mon]
        self.coeffs_nonmon = [np.ones(int(n)) * self.coeffs_init for n in self._cm.n_nm]
        if self.monotonicity.lower() == 'separable monotonicity':
            self.optimization_constraints_lb = [np.asarray([b[0] for b in bk]) for bk in self._cm.bounds]
            self.optimization_constraints_ub = [np.asarray([b[1] for b in bk]) for bk in self._cm.bounds]

        self._itab_d = self._to_dev(self._cm.itab, dtype=torch.int32)
        self._ftab_d = self._to_dev(self._cm.ftab, dtype=torch.int32)
        self._off_d = self._to_dev(self._cm.offsets, dtype=torch.int32)
        self._fdesc_d = self._to_dev(self._cm.fdesc, dtype=torch.int32)
        self._fints_d = self._to_dev(self._cm.fints, dtype=torch.int32)
        self._dpar_d = self._to_dev(self._cm.dpar)
        self._qx_d = self._to_dev(np.asarray(self.quadrature_input['xis'], dtype=float))
        self._qw_d = self._to_dev(np.asarray(self.quadrature_input['Ws'], dtype=float))
        self._prog = _capi.make_program(self._cm, self._itab_d.data_ptr(), self._ftab_d.data_ptr(),
                                        self._fdesc_d.data_ptr(), self._fints_d.data_ptr(), self._dpar_d.data_ptr(),
                                        self._qx_d.data_ptr(), self._qw_d.data_ptr(), self._off_d.data_ptr(),
                                        self._qx_d.numel(),
                                        self.monotonicity, self.rectifier_type, self.delta)
        self._pp = ctypes.byref(self._prog)
        self._u_checked = None
        self._u_rejected = False
        self._epoch = getattr(self, '_epoch', 0) + 1      # packed coefficient vectors of an older program are stale
        self._ugrp_d = None                      # (static U-form tables belong to the specification just compiled)
        self._refresh_uform()
        self._work = None
        self._obj_cache = None

    # ------------------------------------------------------------------------
    # device plumbing
    # ------------------------------------------------------------------------

    def _to_dev(self, a, dtype=None):
        torch = _torch()
        a = np.ascontiguousarray(a)
        if not a.flags.writeable:
            a = a.copy()
        t = torch.from_numpy(a)
        if dtype is not None:
            t = t.to(dtype)
        return t.to(self._dev)

    def _empty(self, *shape, dtype=None):
        torch = _torch()
        return torch.empty(shape, dtype=dtype or torch.float64, device=self._dev)

    def _zeros(self, *shape, dtype=None):
        torch = _torch()
        return torch.zeros(shape, dtype=dtype or torch.float64, device=self._dev)

    @staticmethod
    def _ld(N):
        """Leading dimension of column-major sample matrices: N rounded up to even, so that every column starts
        16-byte aligned and can be read in pairs (what the loader-wave kernels need, include/ttm.h)."""
        return int(N) + (int(N) & 1)

    def _cols(self, ncols, N, zero=False):
        """Column-major ncols x N device matrix with padded leading dimension (a (ncols, ld) tensor)."""
        return (self._zeros if zero else self._empty)(int(ncols), self._ld(N))

    def _stream(self):
        if self._dev.type != 'cuda':
            return None
        # (the raw handle of torch's current stream: torch.cuda.current_stream().cuda_stream builds a Stream object
        # and resolves the device on every call - 3 us of a 12 us launch call)
        torch = _torch()
        idx = getattr(self, '_dev_index', None)
        if idx is None:
            idx = self._dev_index = self._dev.index if self._dev.index is not None else torch.cuda.current_device()
        getter = getattr(torch._C, '_cuda_getCurrentRawStream', None)
        return ctypes.c_void_p(getter(idx) if getter is not None else torch.cuda.current_stream(idx).cuda_stream)

    def _stream_obj(self):
        """torch's current stream as a Stream object, kept while the raw handle stays the same (torch.cuda.current_stream()
        resolves the device through four layers of Python on every call: 12 us, five times per filter update)."""
        torch = _torch()
        raw = self._stream().value
        memo = getattr(self, '_stream_memo', None)
        if memo is None or memo[0] != raw:
            memo = self._stream_memo = (raw, torch.cuda.current_stream(self._dev_index))
        return memo[1]

    def _sync_stream(self):
        """Wait for the work queued on the current stream (hipStreamSynchronize through the library: no Stream object)."""
        if self._dev.type == 'cuda':
            _capi.check(self._lib.ttm_stream_synchronize(self._stream()))

    def _record_event(self):
        ev = _torch().cuda.Event()
        ev.record(self._stream_obj())
        return ev

    @staticmethod
    def _ptr(t, offset=0):
        if t is None:
            return None
        return ctypes.c_void_p(t.data_ptr() + 8 * offset)

    def _workspace(self, n):
        if self._work is None or self._work.numel() < n:
            self._work = self._empty(int(n))
        return self._work

    def _dist(self):
        """torch.distributed handle when samples are sharded over ranks."""
        if not self.shard_samples:
            return None
        import torch.distributed as dist
        return dist if (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1) else None

    def _allreduce(self, t, op='sum'):
        """In-place all-reduce of a short device vector over the ranks that share the SAMPLES (no-op otherwise)."""
        if self._dist() is not None:
            self._allreduce_world(t, op)
        return t

    def _allreduce_world(self, t, op='sum'):
        """The collective itself: ttm_allreduce_f64 / _i32 of the C ABI (RCCL over xGMI) on the current stream; with
        a process group that is not RCCL (gloo rehearsals sharing one GPU) the same reduction through torch.distributed."""
        torch = _torch()
        handle = comm.get(self._lib, force=self._dev.type != 'cuda')
        if handle is not None and t.dtype in (torch.float64, torch.int32) and t.is_contiguous():
            fn = self._lib.ttm_allreduce_f64 if t.dtype == torch.float64 else self._lib.ttm_allreduce_i32
            rc = fn(handle, ctypes.c_void_p(t.data_ptr()), t.numel(), 0 if op == 'sum' else 1, self._stream())
            if rc != 0:
                raise _capi.TTMError('ttm_allreduce: %d: %s' % (rc, self._lib.ttm_comm_last_error().decode()))
            return t
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.SUM if op == 'sum' else dist.ReduceOp.MAX)
        return t

    def _refresh_uform(self):
        """(Re)attach the U-form tables (include/ttm.h "U-form"): called at construction and after every
        special-term placement, which changes the spline geometry."""
        torch = _torch()
        cm = self._cm
        if not cm.u_static:
            return
        termtable.uform_geometry(cm)
        if self._u_rejected:
            cm.u_enabled = False
        # (the component table changes when a placement changes the number of spline intervals - not in most updates of a
        # filter; the geometry always does: one short copy through the page-locked ring)
        self._u_layout = hash(cm.ucomp.tobytes())
        if self._dev.type == 'cuda':
            # geometry (fp64) and component table (int32) in ONE copy through the page-locked ring: the table rides along as the
            # bit pattern of ceil(n / 2) doubles
            uc = np.ascontiguousarray(cm.ucomp, dtype=np.int32)
            if uc.shape[0] % 2:
                uc = np.concatenate((uc, np.zeros(1, dtype=np.int32)))
            ng = int(cm.ugeo.shape[0])
            both = self._to_dev_staged(np.concatenate((np.ascontiguousarray(cm.ugeo, dtype=float), uc.view(np.float64))))
            self._ugeo_d = both[:ng]
            self._ucomp_d = both[ng:].view(torch.int32)
        else:
            self._ucomp_d = self._to_dev(cm.ucomp, dtype=torch.int32)
            self._ugeo_d = self._to_dev(cm.ugeo)
        if getattr(self, '_ugrp_d', None) is None:
            self._ugrp_d = self._to_dev(cm.ugrp, dtype=torch.int32)
            self._umono_d = self._to_dev(cm.umono)
        _capi.set_uform(self._prog, cm, self._ucomp_d.data_ptr(), self._ugrp_d.data_ptr(), self._umono_d.data_ptr(),
                        self._ugeo_d.data_ptr())
        self._u_checked = None
        self._epoch = getattr(self, '_epoch', 0) + 1      # new spline geometry: folds / tables kept with a vector are stale

    def _check_uform(self, fold, err=None):
        """The special-term splines are verified when they are built (fit error against the direct evaluation,
        csrc/ttm_uform.h).  The errors are read back with every packed coefficient vector (2 D doubles); a spline
        outside the tolerance disables the U-form for this map until the next special-term placement (the direct
        kernels run instead) and the fold is redone without it."""
        cm = self._cm
        if not cm.u_enabled:
            return True
        if err is None:
            off = int(self._lib.ttm_uform_offset(self._pp)) + int(cm.u_err_off)
            err = fold[off:off + 2 * cm.D].cpu().numpy()
        err = np.array(err, dtype=float).reshape(cm.D, 2)
        self.uform_fit_error = err
        ok = bool(np.all(err[:, 0] <= termtable.U_TOL_VALUE) and np.all(err[:, 1] <= termtable.U_TOL_DERIV))
        self._u_checked = ok
        if not ok:
            self._u_rejected = True
            cm.u_enabled = False
            self._prog.u_enabled = 0
            self._epoch += 1
        return ok

    # ------------------------------------------------------------------------
    # samples
    # ------------------------------------------------------------------------

    def _import(self, X, standardize):
        """Row-major host array -> standardised column-major device matrix (d x N)."""
        if isinstance(X, _torch().Tensor):
            Xrow = X.contiguous()
        else:
            X = np.ascontiguousarray(X, dtype=np.float64)
            Xrow = self._to_dev(X)
        N, d = Xrow.shape
        Xs = self._cols(d, N, zero=True)
        mean = self._mean_d if standardize else None
        sd = self._std_d if standardize else None
        _capi.check(self._lib.ttm_import(self._ptr(Xrow), N, d, self._ptr(mean), self._ptr(sd), self._ptr(Xs), Xs.shape[1],
                                         self._stream()))
        return Xs

    def _export(self, Xs, N, j0, dout, destandardize, to_host=True):
        out = self._empty(N, dout)
        mean = self._mean_d if destandardize else None
        sd = self._std_d if destandardize else None
        _capi.check(self._lib.ttm_export(self._ptr(Xs), Xs.shape[1], N, j0, dout, self._ptr(mean), self._ptr(sd),
                                         self._ptr(out), self._stream()))
        if not to_host:
            return out
        if self._dev.type == 'cuda' and out.numel() >= (1 << 17):
            # a fresh pageable result pays a page fault per 4 KB on its first touch (37 ms for 320 MB against 5.6 ms on the
            # link): the result is a page-locked buffer of torch's caching host allocator, handed out as the NumPy array
            torch = _torch()
            host = torch.empty(out.shape, dtype=out.dtype, pin_memory=True)
            host.copy_(out, non_blocking=True)
            torch.cuda.current_stream(self._dev).synchronize()
            return host.numpy()
        return out.cpu().numpy()

    # ---- pipelined host boundary of map() / inverse_map() (TM:2391-2437, 3639-3796 take and return NumPy arrays) -------------
    # A row-major host matrix is cut into chunks of rows; chunk c+1 crosses PCIe (side stream) while chunk c runs through the
    # layout change, the map kernels and the layout change back (the caller's stream) and chunk c-1 returns (second side
    # stream) into ONE page-locked result buffer, handed out as the NumPy array - PCIe is full duplex, so a call costs about
    # one crossing of its larger operand instead of two crossings plus 37 ms of first-touch page faults for a fresh pageable
    # 320 MB result (tools/pcie_probe.py: 5.6 ms per direction, 6.6 ms for both at once).
    PIPE_MIN_ROWS = 1 << 18          # below this the plain path (one copy in, one copy out)
    PIPE_CHUNKS = 8

    def _pipe_ok(self, N):
        return self._dev.type == 'cuda' and self.host_pipeline and N >= self.PIPE_MIN_ROWS and self._dist() is None

    def _host_pipeline(self, Xhost, dout, stage):
        """out[r0:r0+n] = stage(rows (n x d_in row-major device tensor), r0, n) for every chunk of rows of the host matrix
        Xhost; stage launches on the current stream and returns an (n x dout) row-major device tensor.  Returns the
        (N x dout) result as a NumPy array over page-locked memory."""
        torch = _torch()
        N, d_in = Xhost.shape
        streams = getattr(self, '_pipe_streams', None)
        if streams is None:
            streams = self._pipe_streams = (torch.cuda.Stream(device=self._dev), torch.cuda.Stream(device=self._dev))
        s_in, s_out = streams
        cur = torch.cuda.current_stream(self._dev)
        nchunk = max(1, min(self.PIPE_CHUNKS, N // (self.PIPE_MIN_ROWS // 2)))
        rows = -(-N // nchunk)
        rows = -(-rows // 4096) * 4096                      # (whole tiles: every chunk starts on an even, 32 KB aligned row)
        bounds = [(r0, min(N, r0 + rows)) for r0 in range(0, N, rows)]
        rows_dev = self._empty(N, d_in)
        out_pin = torch.empty((N, dout), dtype=torch.float64, pin_memory=True)
        # a pageable source does not overlap with anything (the runtime stages it synchronously: H2D + D2H took their sum): the
        # chunk is copied into a ring of page-locked staging buffers by a few host threads (320 MB: 2.9 ms with 8 threads
        # against 5.6 ms on the link) and crosses asynchronously from there
        ring = getattr(self, '_pipe_ring', None)
        if ring is None or ring[0][0].numel() < rows * d_in:
            ring = self._pipe_ring = [[torch.empty(rows * d_in, dtype=torch.float64, pin_memory=True), None] for _ in range(3)]
        s_in.wait_stream(cur)
        ev_in = []
        keep = []
        for i, (r0, r1) in enumerate(bounds):
            slot = ring[i % len(ring)]
            if slot[1] is not None:
                slot[1].synchronize()                     # (the crossing that last read this buffer is done)
            stage_in = slot[0][:(r1 - r0) * d_in].view(r1 - r0, d_in)
            _parallel_copy(stage_in.numpy(), Xhost[r0:r1])
            with torch.cuda.stream(s_in):
                rows_dev[r0:r1].copy_(stage_in, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(s_in)
            slot[1] = ev
            ev_in.append(ev)
            if i >= 1:                                        # (chunk i - 1 computes while chunk i crosses)
                self._pipe_step(bounds[i - 1], ev_in[i - 1], rows_dev, stage, out_pin, cur, s_out, keep)
        self._pipe_step(bounds[-1], ev_in[-1], rows_dev, stage, out_pin, cur, s_out, keep)
        s_out.synchronize()
        cur.synchronize()
        del keep
        return out_pin.numpy()

    def _pipe_step(self, bound, ev_in, rows_dev, stage, out_pin, cur, s_out, keep):
        torch = _torch()
        r0, r1 = bound
        cur.wait_event(ev_in)
        out_dev = stage(rows_dev[r0:r1], r0, r1 - r0)
        ev = torch.cuda.Event()
        ev.record(cur)
        keep.append(out_dev)
        with torch.cuda.stream(s_out):
            s_out.wait_event(ev)
            out_pin[r0:r1].copy_(out_dev, non_blocking=True)

    def _load_samples(self, X):
        torch = _torch()
        X = np.ascontiguousarray(X, dtype=np.float64)
        N, d = X.shape
        self._N = N
        self._Nglobal = N
        dist = self._dist()
        if dist is not None:
            n = torch.tensor([N], dtype=torch.int64, device=self._dev)
            dist.all_reduce(n)
            self._Nglobal = int(n.item())
        self._X_host = None
        if self.standardize_samples:
            self.standardize(X)
        self._Xs = self._import(X, self.standardize_samples)
        self._obj_cache = None
        self.determine_special_term_locations()

    @property
    def X(self):
        """Standardised training samples (host copy, fetched on demand)."""
        if self._X_host is None:
            self._X_host = self._export(self._Xs, self._N, 0, self._Xs.shape[0], False)
        return self._X_host

    # X_mean / X_std (TM:760-781): host copies of the standardisation constants.  After a reset of device-resident samples
    # they are still on their way: the first reader fetches them (the filter's update reads them together with its order statistics)
    def _resolve_moments(self):
        pend = getattr(self, '_moments_pending', None)
        if pend is not None:
            self._moments_pending = None
            both = _torch().stack(pend).cpu().numpy()
            self._X_mean, self._X_std = both[0].copy(), both[1].copy()

    @property
    def X_mean(self):
        self._resolve_moments()
        return self._X_mean

    @X_mean.setter
    def X_mean(self, v):
        self._resolve_moments()
        self._X_mean = v

    @property
    def X_std(self):
        self._resolve_moments()
        return self._X_std

    @X_std.setter
    def X_std(self, v):
        self._resolve_moments()
        self._X_std = v

    def _visit_buffers(self, d):
        """(device, page-locked host) vectors of one host visit of reset_device: [mean (d) | std (d) | 16 order statistics per
        column (16 d)]."""
        torch = _torch()
        vb = getattr(self, '_visit', None)
        if vb is None or vb[0].numel() < 18 * d:
            dev = torch.zeros(18 * d, dtype=torch.float64, device=self._dev)
            host = torch.zeros(18 * d, dtype=torch.float64, pin_memory=self._dev.type == 'cuda')
            vb = self._visit = (dev, host)
        return vb

    def standardize(self, X=None):
        """TM:750-787.  'standard': mean / ddof-0 std per column (device reduction);
        'quantiles': median / quantile spread."""
        torch = _torch()
        if X is None:
            raise ValueError('standardize() needs the raw samples')
        N, d = X.shape
        on_device = isinstance(X, torch.Tensor)          # (reset_device: the raw samples are already there, row-major)
        if self.standardization.lower() == 'standard':
            Xrow = X if on_device else self._to_dev(X)
            visit = self._visit_buffers(d)[0]
            mean, sd = (visit[:d], visit[d:2 * d]) if self._dist() is None else (self._empty(d), self._empty(d))
            work = self._workspace(self._lib.ttm_colstats_work_size(N, d))
            _capi.check(self._lib.ttm_colstats(self._ptr(Xrow), N, d, self._ptr(mean), self._ptr(sd), self._ptr(work),
                                               self._stream()))
            dist = self._dist()
            if dist is not None:
                # combine per-rank moments: mean = sum n_r m_r / n, var = sum n_r (v_r + (m_r - mean)^2) / n
                n_r = float(N)
                s1 = self._allreduce(mean * n_r)
                gmean = s1 / self._Nglobal
                s2 = self._allreduce((sd * sd + (mean - gmean) ** 2) * n_r)
                mean, sd = gmean, torch.sqrt(s2 / self._Nglobal)
            # (the device copies the layout kernels read are the reduction's own output; the host copies are read when
            # somebody asks for them - X_mean / X_std below -, or together with the order statistics of the special-term
            # placement: ONE host visit per reset instead of one per vector)
            self._mean_d, self._std_d = mean, sd
            self._moments_pending = (mean, sd)
            return
        elif self.standardization.lower() in ('quantile', 'quantiles'):
            # median / quantile spread per column (TM:775-778) from device order statistics; the
            # quantiles of X - median are the quantiles of X minus the median (monotone shift)
            Xraw = self._cols(d, N, zero=True)
            Xrow = X if on_device else self._to_dev(X)
            _capi.check(self._lib.ttm_import(self._ptr(Xrow), N, d, None, None, self._ptr(Xraw), Xraw.shape[1], self._stream()))
            med, spread = np.empty(d), np.empty(d)
            for j in range(d):
                med[j] = self._device_quantile(Xraw[j, :N], [0.5])[0]
                qs = self._device_quantile(Xraw[j, :N], [0.8413447460685429, 0.15865525393145707], shift=med[j])
                spread[j] = (qs[0] - qs[1]) / 2
            self.X_mean, self.X_std = med, spread
        else:
            raise ValueError("'standardization' must be either 'standard' or 'quantiles'.")
        self._mean_d = self._to_dev(self.X_mean)
        self._std_d = self._to_dev(self.X_std)

    def _order_statistics(self, col, ranks):
        """Exact order statistics of a device column (K9 radix select).  Sharded ensembles: the same select over the
        shards, bin counts all-reduced per pass (ttm_order_statistics_dist) - no element of the column moves; without an
        RCCL / test-double communicator (gloo rehearsals) the column is gathered first (rank order = sample order)."""
        torch = _torch()
        dist = self._dist()
        handle = None
        n_total = col.numel()
        if dist is not None:
            t = torch.tensor([n_total], dtype=torch.int64, device=self._dev)
            dist.all_reduce(t)
            n_total = int(t.item())
            if n_total >= 2 ** 31:
                # (the distributed select sums its 256 bin counts per pass as int32)
                raise ValueError('order statistics of a sharded column: the global ensemble must stay below 2^31 samples')
            handle = comm.get(self._lib, force=self._dev.type != 'cuda')
            if handle is None:
                sizes = torch.zeros(dist.get_world_size(), dtype=torch.int64, device=self._dev)
                sizes[dist.get_rank()] = col.numel()
                dist.all_reduce(sizes)
                nmax = int(sizes.max().item())
                pad = torch.zeros(nmax, dtype=col.dtype, device=self._dev)
                pad[:col.numel()] = col
                parts = [torch.empty_like(pad) for _ in range(dist.get_world_size())]
                dist.all_gather(parts, pad)
                col = torch.cat([p_[:int(n_)] for p_, n_ in zip(parts, sizes.tolist())])
        col = col.contiguous()
        ranks = np.asarray(ranks, dtype=np.int64)
        out = np.empty(len(ranks))
        # (scratch and the rank vectors - functions of N and the requested quantiles only - stay on the device between calls:
        # the filter asks for the same order statistics in every update: _launch_select)
        for i in range(0, len(ranks), 16):
            o = self._empty(len(ranks[i:i + 16]))
            self._launch_select(col, ranks[i:i + 16], o, handle)
            out[i:i + 16] = o.cpu().numpy()
        return out, n_total

    def _launch_select(self, col, ranks, out, handle=None):
        """ttm_order_statistics of <= 16 ranks of a device column into the device vector `out` (no host visit)."""
        torch = _torch()
        work = getattr(self, '_select_work', None)
        if work is None:
            work = self._select_work = torch.zeros(int(self._lib.ttm_select_work_size(16)), dtype=torch.uint8, device=self._dev)
        rcache = getattr(self, '_select_ranks', None)
        if rcache is None or len(rcache) > 64:
            rcache = self._select_ranks = {}
        rkey = ranks.tobytes()
        r = rcache.get(rkey)
        if r is None:
            r = rcache[rkey] = self._to_dev(np.ascontiguousarray(ranks, dtype=np.int64))
        if handle is not None:
            _capi.check(self._lib.ttm_order_statistics_dist(self._ptr(col), col.numel(), ctypes.c_void_p(r.data_ptr()), r.numel(),
                                                            self._ptr(out), ctypes.c_void_p(work.data_ptr()), handle, self._stream()))
        else:
            _capi.check(self._lib.ttm_order_statistics(self._ptr(col), col.numel(), ctypes.c_void_p(r.data_ptr()),
                                                       r.numel(), self._ptr(out), ctypes.c_void_p(work.data_ptr()),
                                                       self._stream()))

    def _device_quantile(self, col, q, shift=None):
        """np.quantile(col - shift, q) (method 'linear'), bit-identical, from device order statistics."""
        dist = self._dist()
        n = col.numel()
        if dist is not None:
            t = _torch().tensor([n], dtype=_torch().int64, device=self._dev)
            dist.all_reduce(t)
            n = int(t.item())
        return quantile.quantile_from_order_statistics(n, q, lambda ranks: self._order_statistics(col, ranks)[0], shift)

    def determine_special_term_locations(self, k=None):
        """TM:2219-2389: centres = quantiles of the standardised training columns."""
        self._linearization_thresholds()
        req = termtable.quantile_requests(self.special_terms)
        if len(req) == 0:
            return
        memo = {}
        lut = self._prefetch_order_statistics(req)

        def column_quantiles(var, q):
            key = (var, tuple(np.asarray(q, dtype=float).tolist()))
            if key not in memo:
                if lut is not None and var in lut:
                    memo[key] = quantile.quantile_from_order_statistics(self._N, q, lambda ranks: [lut[var][int(r)] for r in ranks])
                else:
                    memo[key] = self._device_quantile(self._Xs[var, :self._N], q)
            return memo[key]
        termtable.place_special_terms(self.special_terms, column_quantiles, self.ST_scale_factor, self.ST_scale_mode)
        self._cm.fill_special_terms(self.special_terms)
        if self._dev.type == 'cuda':
            self._to_dev_staged(self._cm.dpar, out=self._dpar_d)        # (page-locked staging: no synchronous copy in the runtime)
        else:
            self._dpar_d.copy_(_torch().from_numpy(self._cm.dpar))
        self._dpar_version = getattr(self, '_dpar_version', 0) + 1
        self._gram_ahead()
        self._epoch += 1                         # (the folded special-term records depend on centres and scales)
        self._u_rejected = False
        self._refresh_uform()

    def _prefetch_order_statistics(self, req):
        """Every order statistic the placement will interpolate between (req: {column: quantiles}), selected column by column
        WITHOUT a host visit in between and read in one copy - together with the column moments of a reset that are still on
        the device.  {column: {rank: value}}, or None when this does not apply (sharded samples, more than 16 ranks of a
        column): the placement then asks column by column."""
        torch = _torch()
        if self._dist() is not None:
            return None
        d = self._cm.d_cols
        plans = {}
        for var, qs in req.items():
            ranks = quantile._plan_lists(self._N, qs)[3].astype(np.int64, copy=False)
            if len(ranks) > 16 or not 0 <= int(var) < d:
                return None
            plans[int(var)] = ranks
        dev, host = self._visit_buffers(d)
        for var, ranks in plans.items():
            self._launch_select(self._Xs[var, :self._N], ranks, dev[2 * d + 16 * var:2 * d + 16 * var + len(ranks)])
        host.copy_(dev, non_blocking=True)
        self._sync_stream()
        h = host.numpy()
        if getattr(self, '_moments_pending', None) is not None and self._moments_pending[0].data_ptr() == dev.data_ptr():
            self._moments_pending = None
            self._X_mean, self._X_std = h[:d].copy(), h[d:2 * d].copy()
        return {var: dict(zip(ranks.tolist(), h[2 * d + 16 * var:2 * d + 16 * var + len(ranks)].tolist()))
                for var, ranks in plans.items()}

    def _linearization_thresholds(self):
        """TM:2364-2389: per column the quantiles (linearization, 1 - linearization) of the standardised training
        samples, or +-linearization.  The 'LIN' modifier itself leaves the map unchanged: as the reference's
        generated code executes (TM:1375-1385, "__x__" is already replaced when the clipped / extended variables are
        substituted) both sides of its blend evaluate the factor at the unclipped x, P(x)(1 - v/inc) + P(x) v/inc,
        i.e. P(x) up to rounding noise of relative size |v|/inc * 1e-16; the factor is evaluated once here."""
        if self.linearization is None:
            return
        d = self._cm.d_cols
        thr = np.zeros((d, 2))
        for j in range(d):
            if self.linearization_specified_as_quantiles:
                thr[j, 0] = self._device_quantile(self._Xs[j, :self._N], [self.linearization])[0]
                thr[j, 1] = self._device_quantile(self._Xs[j, :self._N], [1 - self.linearization])[0]
            else:
                thr[j, 0], thr[j, 1] = -self.linearization, +self.linearization
        self.linearization_threshold = thr

    def reset(self, X):
        """TM:710-748: new samples, coefficients back to coeffs_init."""
        X = np.asarray(X)
        if len(X.shape) != 2:
            raise Exception('X should be a two-dimensional array of shape (N,D), N = number of samples, D = number '
                            'of dimensions. Current shape of X is ' + str(X.shape))
        if X.shape[-1] != self._cm.d_cols:
            raise Exception('X has ' + str(X.shape[-1]) + ' columns, the map was built for ' + str(self._cm.d_cols))
        for k in range(self.D):
            self.coeffs_mon[k] = np.asarray(self.coeffs_mon[k], dtype=float) * 0 + self.coeffs_init
            self.coeffs_nonmon[k] = np.asarray(self.coeffs_nonmon[k], dtype=float) * 0 + self.coeffs_init
        self._load_samples(X)

    def reset_device(self, Xcols, N):
        """reset() (TM:710-748) for samples that are already on the device: Xcols is a raw (un-standardised)
        column-major matrix (d x ld tensor, column j = sample column j); nothing of size N crosses the host boundary -
        only the 2 d column moments and the few order statistics the special-term placement interpolates.
        Same kernels, hence the same numbers, as reset() of the same samples."""
        torch = _torch()
        d = self._cm.d_cols
        if Xcols.shape[0] != d or Xcols.shape[1] < N:
            raise Exception('Xcols has shape ' + str(tuple(Xcols.shape)) + ', the map was built for ' + str(d) + ' columns')
        for k in range(self.D):
            self.coeffs_mon[k] = np.asarray(self.coeffs_mon[k], dtype=float) * 0 + self.coeffs_init
            self.coeffs_nonmon[k] = np.asarray(self.coeffs_nonmon[k], dtype=float) * 0 + self.coeffs_init
        self._N = N
        self._Nglobal = N
        dist = self._dist()
        if dist is not None:
            n = torch.tensor([N], dtype=torch.int64, device=self._dev)
            dist.all_reduce(n)
            self._Nglobal = int(n.item())
        self._X_host = None
        self._obj_cache = None
        if not (dist is None and self.standardize_samples and self._standardize_cols(Xcols, N)):
            Xrow = self._export(Xcols, N, 0, d, False, to_host=False)      # row-major copy for the moment / layout kernels
            if self.standardize_samples:
                self.standardize(Xrow)
            self._Xs = self._import(Xrow, self.standardize_samples)
        self.determine_special_term_locations()

    def _standardize_cols(self, Xcols, N):
        """'standard' standardisation (TM:760-771) of a column-major device matrix without a row-major copy: moments in one
        launch (ttm_colstats_cols: the same sums, hence the same bits, as ttm_colstats of the exported rows), one
        elementwise pass.  False: not for this shape / setting (the caller exports)."""
        if self.standardization.lower() != 'standard' or not hasattr(self._lib, 'ttm_colstats_cols'):
            return False
        d = self._cm.d_cols
        visit = self._visit_buffers(d)[0]
        mean, sd = visit[:d], visit[d:2 * d]
        work = self._workspace(self._lib.ttm_colstats_work_size(N, d))
        st = self._stream()
        rc = self._lib.ttm_colstats_cols(self._ptr(Xcols), Xcols.shape[1], N, d, self._ptr(mean), self._ptr(sd), self._ptr(work), st)
        if rc == _capi.TTM_E_UNSUPPORTED:
            return False
        _capi.check(rc)
        # (the matrix of the reset before is taken over when it has the same shape: its padding beyond N is zero already)
        Xs = getattr(self, '_Xs', None)
        if Xs is None or not getattr(Xs, '_ttm_reset_cols', False) or tuple(Xs.shape) != (d, self._ld(N)):
            Xs = self._cols(d, N, zero=True)
            Xs._ttm_reset_cols = True
        _capi.check(self._lib.ttm_standardize_cols(self._ptr(Xcols), Xcols.shape[1], N, d, self._ptr(mean), self._ptr(sd),
                                                   self._ptr(Xs), Xs.shape[1], st))
        self._mean_d, self._std_d = mean, sd
        self._moments_pending = (mean, sd)
        self._Xs = Xs
        return True

    def map_columns(self, src, ncols_out, N, source=None, scale=None, shift=None, out=None):
        """out[j] = source[src[j]] * scale[j] + shift[j] on column-major device matrices (ttm_map_columns): column
        gather / permutation, constants (src[j] < 0) and (de)standardisation in one pass."""
        src = np.ascontiguousarray(src, dtype=np.int32)
        scale = None if scale is None else np.ascontiguousarray(scale, dtype=float)
        shift = None if shift is None else np.ascontiguousarray(shift, dtype=float)
        out = self._cols(ncols_out, N) if out is None else out
        p = lambda a: None if a is None else ctypes.c_void_p(a.ctypes.data)      # noqa: E731
        _capi.check(self._lib.ttm_map_columns(self._ptr(source), 0 if source is None else source.shape[1], p(src), p(scale), p(shift),
                                              len(src), N, self._ptr(out), out.shape[1], self._stream()))
        return out

    # ------------------------------------------------------------------------
    # coefficients
    # ------------------------------------------------------------------------

    def _pack_coeffs(self, override_k=None, coeffs_nonmon=None, coeffs_mon=None):
        host = None
        if override_k is None:
            # the common case - every entry already a 1-D float64 array of the right length - is ONE concatenate of the
            # interleaved lists (9 instead of 31 us at D = 40); anything else takes the checked path below
            try:
                parts = [None] * (2 * self.D)
                parts[0::2], parts[1::2] = self.coeffs_nonmon, self.coeffs_mon
                if self._pack_shape is None:
                    self._pack_shape = [int(n) for k in range(self.D) for n in (self._cm.n_nm[k], self._cm.n_mon[k])]
                if all(type(a) is np.ndarray and a.ndim == 1 and a.dtype == np.float64 for a in parts) and \
                        [a.shape[0] for a in parts] == self._pack_shape:
                    host = np.concatenate(parts)
            except Exception:                           # noqa: BLE001
                host = None
        if host is None:
            parts = []
            for k in range(self.D):
                cn = self.coeffs_nonmon[k] if (k != override_k or coeffs_nonmon is None) else coeffs_nonmon
                cm = self.coeffs_mon[k] if (k != override_k or coeffs_mon is None) else coeffs_mon
                cn, cm = np.asarray(cn, dtype=float).ravel(), np.asarray(cm, dtype=float).ravel()
                if len(cn) != self._cm.n_nm[k] or len(cm) != self._cm.n_mon[k]:
                    raise ValueError('component %d expects %d nonmonotone and %d monotone coefficients, got %d and %d'
                                     % (k, self._cm.n_nm[k], self._cm.n_mon[k], len(cn), len(cm)))
                parts += [cn, cm]
            host = np.concatenate(parts)
        # the public methods pack on every call: an unchanged coefficient vector (the common case - map() and
        # inverse_map() of one fitted map) reuses the packed, folded vector and the inverse tables kept with it
        memo = getattr(self, '_pack_memo', None)
        if override_k is None and memo is not None and memo[0] == self._epoch and np.array_equal(memo[1], host):
            return memo[2]
        coef = self._fold_staged(host) if (self._dev.type == 'cuda' and self._cm.u_enabled) else None
        if coef is None:
            coef = self._fold(self._to_dev_staged(host))
        if override_k is None:
            self._pack_memo = (coef._ttm_epoch, host, coef)
        return coef

    def _to_dev_staged(self, host, out=None):
        """Host -> device copy of a short fp64 vector through a ring of pinned staging buffers (asynchronous on the current
        stream; a pageable source costs a synchronous staging copy inside the runtime: 17 against 8 us for 3.4 KB)."""
        torch = _torch()
        if self._dev.type != 'cuda':
            return self._to_dev(host)
        n = int(host.shape[0])
        ring = getattr(self, '_stage_ring', None)
        if ring is None or ring[0][0].numel() < n:
            ring = self._stage_ring = [[torch.empty(max(n, 64), dtype=torch.float64).pin_memory(), None] for _ in range(8)]
            self._stage_next = 0
        slot = ring[self._stage_next]
        self._stage_next = (self._stage_next + 1) % len(ring)
        if slot[1] is not None:
            slot[1].synchronize()                       # (the copy that last used this buffer - eight uploads ago - is done)
        slot[0][:n].numpy()[:] = host
        dev = torch.empty(n, dtype=torch.float64, device=self._dev) if out is None else out
        dev.copy_(slot[0][:n], non_blocking=True)
        slot[1] = self._record_event()
        return dev

    def _fold_staged(self, host):
        """A new coefficient vector of a U-form map with NO copy in the launch stream (an H2D copy between two kernels sits
        between two engine switches: 3.7 us + an 11 us gap, tools/trace_uncached.sh): the packed vector is written into a
        page-locked ring slot, the fold kernel reads it from there and fills the device vector itself (ttm_fold_staged); the
        spline fit errors and the sortedness flags of the default inverse tables come back the same way - written to
        page-locked memory by the kernels, read behind an event (eager mode: behind one stream synchronisation).
        None: not applicable (the caller takes the copying path)."""
        torch = _torch()
        n = int(host.shape[0])
        D = self._cm.D
        ring = getattr(self, '_fold_ring', None)
        if ring is None or ring[0][0].numel() < n or ring[0][1].numel() < 2 * D:
            ring = self._fold_ring = [[torch.empty(max(n, 64), dtype=torch.float64, pin_memory=True),
                                       torch.empty(2 * D, dtype=torch.float64, pin_memory=True),
                                       torch.empty(max(D, 16), dtype=torch.int32, pin_memory=True), None, None, None, None]
                                      for _ in range(8)]
            self._fold_ring_next = 0
        slot = ring[self._fold_ring_next]
        self._fold_ring_next = (self._fold_ring_next + 1) % len(ring)
        if slot[3] is not None:
            slot[3].synchronize()                       # (the kernels that last used this slot - eight vectors ago - are done)
            prev = slot[4]() if slot[4] is not None else None
            if prev is not None and getattr(prev, '_ttm_pending', None) is not None and not self.validate(prev):
                prev._ttm_failed = True                 # (its deferred checks live in this slot: read before the slot is reused)
        slot[0][:n].numpy()[:] = host
        coef = torch.empty(n, dtype=torch.float64, device=self._dev)
        # The fold buffer of the vector that used this slot eight vectors ago is taken over when that vector is gone and the
        # layout is still the same (`_epoch`): the set of slots a fold writes is a function of the layout alone, so every
        # slot the old fold wrote is written again and the others still hold the zeros of the first allocation - no
        # zero-fill kernel (5 us) in front of the fold.  (Stream order protects the old contents' readers.)
        nfold = int(self._lib.ttm_fold_size(self._pp))
        prev = slot[4]() if slot[4] is not None else None
        # (`_u_layout`: the component table of the U section - a new special-term placement moves constants, not slots, unless it
        # changes the number of spline intervals)
        lkey = (id(self._cm), getattr(self, '_u_layout', self._epoch), nfold)
        if prev is None and slot[5] is not None and slot[6] == lkey:
            fold = slot[5]
        else:
            fold = self._zeros(nfold)
        slot[5], slot[6] = fold, lkey
        st = self._stream()
        pending = None
        rc = -1
        if self._eager_tables():
            # fold + U section and the default inverse tables as ONE launch (include/ttm.h: ttm_setup_staged - the table workgroups
            # fold for themselves into a scratch copy that belongs to this map: launches on one stream run one after the other)
            scratch = getattr(self, '_fold2', None)
            skey = (lkey, st.value if hasattr(st, 'value') else st)
            if scratch is None or scratch[0] != skey:
                scratch = self._fold2 = (skey, self._zeros(nfold))
            coef._ttm_fold = fold
            pending = self._launch_default_tables(coef, h_unsorted=slot[2], staged=(slot[0], slot[1], scratch[1]))
            rc = 0 if pending is not None else -1
        if rc != 0:
            rc = self._lib.ttm_fold_staged(self._pp, ctypes.c_void_p(slot[0].data_ptr()), self._ptr(coef), self._ptr(fold),
                                           ctypes.c_void_p(slot[1].data_ptr()), st)
            if rc != 0:
                slot[3] = None
                return None
        coef._ttm_fold = fold
        coef._ttm_tables = {}
        if pending is None and self._eager_tables():
            pending = self._launch_default_tables(coef, h_unsorted=slot[2])
        done = self._record_event()
        slot[3] = done
        import weakref
        slot[4] = weakref.ref(coef)
        errs = slot[1][:2 * D]
        flags = slot[2][:D] if pending is not None else None
        coef._ttm_epoch = self._epoch
        if getattr(self, 'deferred_checks', False):
            if pending is not None:
                tkey, entry = pending
                coef._ttm_tables[tkey] = entry[:4] + (True, entry[5])
            # (the ring slot is reused eight vectors on: validate() must have read it by then - it copies at once)
            coef._ttm_pending = (fold, pending, errs, flags, done)
            return coef
        done.synchronize()
        if not self._check_uform(fold, errs.numpy().copy()):
            return self._fold(coef)                     # (U-form rejected: folded again without it, the plain path)
        if pending is not None:
            tkey, entry = pending
            coef._ttm_tables[tkey] = entry[:4] + (int(flags.numpy().max()) == 0, entry[5])
        return coef

    def _fold(self, coef):
        """Folded coefficients of a packed coefficient vector (device pre-pass, include/ttm.h "Folded coefficients"),
        kept with the vector together with the state of the map they were made for: special-term placement, U-form
        geometry and its accept / reject decision (`_epoch`).  A vector that outlived that state - a caller of
        forward_device / inverse_device holding it across reset() - is folded again on its next use (`_current`)."""
        fold = self._zeros(int(self._lib.ttm_fold_size(self._pp)))
        _capi.check(self._lib.ttm_fold(self._pp, self._ptr(coef), self._ptr(fold), self._stream()))
        coef._ttm_fold = fold
        coef._ttm_tables = {}
        # The default inverse tables of a separable map are launched right behind the fold, BEFORE the one host visit of
        # a new coefficient vector: the fit errors of the splines and the sortedness flags of the tables then come back
        # behind a single synchronisation (two before: here and in the first inverse_map)
        pending = self._launch_default_tables(coef) if self._eager_tables() else None
        if getattr(self, 'deferred_checks', False) and self._dev.type == 'cuda':
            # no host visit now (`deferred_checks`, validate()): the kernels behind this vector run on the assumption that
            # the splines fit and the tables are sorted - what the checks find for every map seen so far - and the flags
            # are read at the caller's next synchronisation point
            torch = _torch()
            errs = flags = None
            if self._cm.u_enabled:                          # (copies into pinned memory, in stream order behind the kernels)
                off = int(self._lib.ttm_uform_offset(self._pp)) + int(self._cm.u_err_off)
                errs = torch.empty(2 * self._cm.D, dtype=torch.float64, pin_memory=True)
                errs.copy_(fold[off:off + 2 * self._cm.D], non_blocking=True)
            if pending is not None:
                tkey, entry = pending
                coef._ttm_tables[tkey] = entry[:4] + (True, entry[5])
                flags = torch.empty(entry[4].numel(), dtype=torch.int32, pin_memory=True)
                flags.copy_(entry[4], non_blocking=True)
            done = torch.cuda.Event()
            done.record()
            coef._ttm_pending = (fold, pending, errs, flags, done)
            coef._ttm_epoch = self._epoch
            return coef
        if not self._check_uform(fold):
            fold = self._zeros(int(self._lib.ttm_fold_size(self._pp)))
            _capi.check(self._lib.ttm_fold(self._pp, self._ptr(coef), self._ptr(fold), self._stream()))
            coef._ttm_fold = fold
            coef._ttm_tables = {}
            pending = self._launch_default_tables(coef) if self._eager_tables() else None
        if pending is not None:
            tkey, entry = pending
            coef._ttm_tables[tkey] = entry[:4] + (int(entry[4].cpu().max().item()) == 0, entry[5])
        coef._ttm_epoch = self._epoch
        return coef

    def validate(self, coef=None):
        """With `deferred_checks = True` the device entry points (forward_device, inverse_device, density_device) do not
        wait for the two checks every new coefficient vector gets - fit errors of the special-term splines, sortedness of
        the default inverse tables - before they launch.  validate() reads them (ONE synchronisation, at a point of the
        caller's choosing: it must come before results computed with this vector are used) and returns True when
        both held.  False: the state is repaired (U-form off and the vector folded again / the tables marked unsorted)
        and everything computed with `coef` since it was packed has to be computed again."""
        coef = self._pack_memo[2] if coef is None and getattr(self, '_pack_memo', None) is not None else coef
        pend = getattr(coef, '_ttm_pending', None) if coef is not None else None
        if pend is None:
            if coef is not None and getattr(coef, '_ttm_failed', False):
                coef._ttm_failed = False              # (a check that failed when it was consumed early, see _current)
                return False
            return True
        fold, pending, errs, flags, done = pend
        coef._ttm_pending = None
        done.synchronize()                                   # (the copies behind the fold / index kernels, not the stream)
        ok = True
        if pending is not None:
            tkey, entry = pending
            if int(flags.numpy().max()) != 0:
                coef._ttm_tables[tkey] = entry[:4] + (False, entry[5])
                ok = False
        if coef._ttm_epoch == self._epoch and errs is not None and not self._check_uform(fold, errs.numpy()):
            saved, self.deferred_checks = self.deferred_checks, False
            try:
                self._fold(coef)
            finally:
                self.deferred_checks = saved
            ok = False
        return ok

    def _eager_tables(self):
        # the default inverse tables ride along with the fold of a new coefficient vector (their flags come back behind the same
        # synchronisation) - once this map HAS been inverted by table: a caller that only runs forward / density passes over
        # many coefficient vectors pays neither the launch nor the allocations
        return (self.alternate_root_finding and self.monotonicity.lower() == 'separable monotonicity' and self._cm.u_enabled and
                getattr(self, '_cm', None) is not None and self._cm.u_h_cls > 0 and getattr(self, '_inverse_seen', False))

    def _launch_default_tables(self, coef, resolution=1001, start_distance=10, h_unsorted=None, staged=None):
        """Build + index the inverse tables of all components for the default table geometry (TM:4047-4058), no host visit:
        returns (cache key, (tables, tmin, tmax, bucket index, unsorted flags on the device)).  staged = (page-locked packed
        vector, page-locked fit errors, fold scratch): the fold of the vector rides in the same launch (ttm_setup_staged);
        None is returned when the library declines (nothing launched)."""
        torch = _torch()
        nb = self._inv_nb()
        self._ensure_pts(resolution, start_distance)
        ncomp = self.D
        st = self._stream()
        out_d = self._empty(ncomp, resolution)
        tmin_d, tmax_d = self._empty(ncomp), self._empty(ncomp)
        bkt_d = self._empty(ncomp, nb + 1, dtype=torch.int32)
        uns_d = self._empty(ncomp, dtype=torch.int32)
        img_d = self._table_images(0, ncomp, resolution, nb)
        if staged is not None:
            h_coef, h_err, fold2 = staged
            rc = self._lib.ttm_setup_staged(self._pp, ctypes.c_void_p(h_coef.data_ptr()), self._ptr(coef), self._ptr(coef._ttm_fold),
                                            self._ptr(fold2), ctypes.c_void_p(h_err.data_ptr()), self._ptr(self._pts_d), resolution, nb,
                                            self._ptr(out_d), self._ptr(tmin_d), self._ptr(tmax_d), ctypes.c_void_p(bkt_d.data_ptr()),
                                            ctypes.c_void_p(uns_d.data_ptr()),
                                            None if h_unsorted is None else ctypes.c_void_p(h_unsorted.data_ptr()), self._ptr(img_d), st)
            if rc != 0:
                return None
            return (0, ncomp, resolution, start_distance, nb), (out_d, tmin_d, tmax_d, bkt_d, uns_d, img_d)
        _capi.check(self._lib.ttm_inverse_table_build_index(self._pp, self._ptr(coef), self._ptr(coef._ttm_fold), 0, ncomp,
                                                            self._ptr(self._pts_d), resolution, nb, self._ptr(out_d),
                                                            self._ptr(tmin_d), self._ptr(tmax_d), ctypes.c_void_p(bkt_d.data_ptr()),
                                                            ctypes.c_void_p(uns_d.data_ptr()),
                                                            None if h_unsorted is None else ctypes.c_void_p(h_unsorted.data_ptr()),
                                                            self._ptr(img_d), st))
        return (0, ncomp, resolution, start_distance, nb), (out_d, tmin_d, tmax_d, bkt_d, uns_d, img_d)

    def _table_images(self, k0, k1, resolution, nb):
        """Room for the resident-table images of the components [k0, k1) (include/ttm.h: ttm_inverse_table_build_index writes them,
        the lookup kernel of banded maps copies them into LDS by DMA), or None when this map / table geometry has no such kernel."""
        per = int(self._lib.ttm_inverse_table_image_doubles(self._pp, k0, k1, resolution, nb))      # (asked every time: a function of the launch options too)
        return self._empty((k1 - k0) * per) if per > 0 else None

    def _inv_nb(self):
        """Buckets of the table index (~ table points; nb + 1 int32 per row = whole 16-byte units); TTM_INV_NB is read once."""
        nb = getattr(transport_map, '_INV_NB', None)
        if nb is None:
            import os
            nb = transport_map._INV_NB = int(os.environ.get('TTM_INV_NB', 1023))
        return nb

    def _ensure_pts(self, resolution, start_distance):
        key = (resolution, start_distance)
        if getattr(self, '_pts_key', None) != key:
            self._pts = np.linspace(-start_distance, start_distance, resolution)
            self._pts_d = self._to_dev(self._pts)
            self._pts_key = key
            # np.linspace is i*step + start with the end point forced: let the kernel compute it when the
            # restatement reproduces every bit
            step = (2.0 * start_distance) / (resolution - 1)
            regen = np.arange(0, resolution) * step + (-float(start_distance))
            regen[-1] = float(start_distance)
            self._pts_affine = ((ctypes.c_double * 3)(-float(start_distance), step, float(start_distance))
                                if np.array_equal(regen, self._pts) else None)

    def _current(self, coef):
        if coef is None:
            return self._pack_coeffs()
        pend = getattr(coef, '_ttm_pending', None)
        if pend is not None and pend[4].query():
            # deferred checks whose copies have landed are read at the next use of the vector, not only at an explicit validate():
            # a failed one repairs the state now and is reported by the validate() the caller still owes
            if not self.validate(coef):
                coef._ttm_failed = True
        if getattr(coef, '_ttm_epoch', None) != self._epoch:
            if coef.numel() != int(self._cm.coef_off[-1]):
                raise ValueError('packed coefficient vector belongs to a different specification of the map')
            self._fold(coef)
        return coef

    # ------------------------------------------------------------------------
    # forward map
    # ------------------------------------------------------------------------

    def _samples_for(self, X):
        """TM:2410-2422: a user X is only used when standardize_samples is True."""
        if X is not None and self.standardize_samples:
            X = np.asarray(X)
            if X.ndim != 2 or X.shape[1] != self._cm.d_cols:
                raise ValueError('X must have shape (N, %d)' % self._cm.d_cols)
            return self._import(X, True), X.shape[0]
        return self._Xs, self._N

    def map(self, X=None):
        """TM:2391-2437: Z[:, k] = S_k(x) for all map components."""
        if X is not None and self.standardize_samples and not isinstance(X, _torch().Tensor) and self._pipe_ok(np.shape(X)[0]):
            X = np.ascontiguousarray(X, dtype=np.float64)
            if X.ndim != 2 or X.shape[1] != self._cm.d_cols:
                raise ValueError('X must have shape (N, %d)' % self._cm.d_cols)
            coef = self._pack_coeffs()
            N, d, D = X.shape[0], self._cm.d_cols, self.D
            Xs, Z = self._cols(d, N), self._cols(D, N)
            st = self._stream()

            def stage(rows, r0, n):
                _capi.check(self._lib.ttm_import(self._ptr(rows), n, d, self._ptr(self._mean_d), self._ptr(self._std_d),
                                                 self._ptr(Xs, r0), Xs.shape[1], st))
                _capi.check(self._lib.ttm_forward(self._pp, self._ptr(coef), self._ptr(coef._ttm_fold), self._ptr(Xs, r0), Xs.shape[1], n,
                                                  0, D, self._ptr(Z, r0), Z.shape[1], None, None, None, st))
                out = self._empty(n, D)
                _capi.check(self._lib.ttm_export(self._ptr(Z, r0), Z.shape[1], n, 0, D, None, None, self._ptr(out), st))
                return out
            return self._host_pipeline(X, D, stage)
        Xs, N = self._samples_for(X)
        coef = self._pack_coeffs()
        Z = self._cols(self.D, N)
        _capi.check(self._lib.ttm_forward(self._pp, self._ptr(coef), self._ptr(coef._ttm_fold), self._ptr(Xs), Xs.shape[1], N, 0, self.D,
                                          self._ptr(Z), Z.shape[1], None, None, None, self._stream()))
        return self._export(Z, N, 0, self.D, False)

    # device-resident entry points (column-major tensors in, column-major tensors out; no PCIe traffic)
    def forward_device(self, Xs, N, coef=None, Z=None, logdet=None, sigma=None, sumsq=None):
        """S(x) for a standardised column-major device matrix Xs (d x N) -> Z (D x N)."""
        coef = self._current(coef)
        Z = self._cols(self.D, N) if Z is None else Z
        _capi.check(self._lib.ttm_forward(self._pp, self._ptr(coef), self._ptr(coef._ttm_fold), self._ptr(Xs), Xs.shape[1], N, 0, self.D,
                                          self._ptr(Z), Z.shape[1], self._ptr(logdet), self._ptr(sigma), self._ptr(sumsq),
                                          self._stream()))
        return Z

    def density_device(self, Xs, N, coef=None, logdet=None, sigma=None, sumsq=None):
        """The density pass WITHOUT the map values: sum_k log(dS_k/dx_k / sigma_k) and / or sum_k S_k^2 per sample of a
        standardised column-major device matrix (what evaluate_pullback_density needs, TM:2646-2712: 8 N (d + 1) bytes)."""
        coef = self._current(coef)
        _capi.check(self._lib.ttm_forward(self._pp, self._ptr(coef), self._ptr(coef._ttm_fold), self._ptr(Xs), Xs.shape[1], N, 0, self.D,
                                          None, N, self._ptr(logdet), self._ptr(sigma), self._ptr(sumsq), self._stream()))
        return logdet, sumsq

    def roundtrip_device(self, Xs, N, coef=None, Z=None, Xr=None, logdet=None, sigma=None, sumsq=None):
        """Z = S(x) and S^-1(Z) of a standardised column-major device matrix in ONE pass (ttm_roundtrip: maps of at most four
        components, N >= 65 536 - what BASELINE configs[1] times as a step; with logdet / sumsq also the terms of the pullback
        density).  The same bits as forward_device followed by inverse_device on the default tables - which is what runs when
        the fused kernel does not apply.  Returns (Z, Xr); the conditioning columns of Xr are copies of Xs's."""
        coef = self._current(coef)
        Z = self._cols(self.D, N) if Z is None else Z
        Xr = self._cols(self._cm.d_cols, N, zero=True) if Xr is None else Xr
        skip = self._cm.d_cols - self.D
        if skip > 0:
            Xr[:skip, :N].copy_(Xs[:skip, :N])
        table = self.alternate_root_finding and self.monotonicity.lower() == 'separable monotonicity' and self.root_search_truncation
        rc = _capi.TTM_E_UNSUPPORTED
        if table and self._dev.type == 'cuda':
            resolution, start_distance, nb = 1001, 10, self._inv_nb()
            self._inverse_table(coef, 0, self.D, None, None, 0, resolution, start_distance)      # (the tables, if this vector has none yet)
            out_d, tmin_d, tmax_d, bkt_d, is_sorted, _ = coef._ttm_tables[(0, self.D, resolution, start_distance, nb)]
            if is_sorted:
                rc = self._lib.ttm_roundtrip(self._pp, self._ptr(coef), self._ptr(coef._ttm_fold), self._ptr(Xs), Xs.shape[1], N,
                                             self._ptr(Z), Z.shape[1], self._ptr(Xr), Xr.shape[1], self._ptr(logdet), self._ptr(sigma),
                                             self._ptr(sumsq), self._ptr(out_d), resolution, self._pts_affine, self._ptr(tmin_d),
                                             self._ptr(tmax_d), ctypes.c_void_p(bkt_d.data_ptr()), nb, self._stream())
        if rc == _capi.TTM_E_UNSUPPORTED:
            self.forward_device(Xs, N, coef, Z=Z, logdet=logdet, sigma=sigma, sumsq=sumsq)
            self.inverse_device(Z, N, coef, X=Xr)
        else:
            _capi.check(rc)
        return Z, Xr

    def inverse_device(self, Zs, N, coef=None, X=None, table=None):
        """S^{-1}(z) for a column-major device matrix Zs (D x N) -> standardised X (d x N);
        conditioning columns (if any) must already be in X."""
        coef = self._current(coef)
        X = self._cols(self._cm.d_cols, N, zero=True) if X is None else X
        if table is None:
            table = self.alternate_root_finding and self.monotonicity.lower() == 'separable monotonicity'
        if table:
            self._inverse_table(coef, 0, self.D, Zs, X, N)
        else:
            self._inverse_bisect(coef, 0, self.D, Zs, X, N)
        return X

    def s(self, x, k, coeffs_nonmon=None, coeffs_mon=None):
        """TM:2439-2567: k-th map component on already-standardised samples x
        (None = the training samples)."""
        if x is None:
            Xs, N = self._Xs, self._N
        else:
            x = np.asarray(x)
            Xs, N = self._import(x, False), x.shape[0]
        coef = self._pack_coeffs(k, coeffs_nonmon, coeffs_mon)
        Z = self._cols(1, N)
        _capi.check(self._lib.ttm_forward(self._pp, self._ptr(coef), self._ptr(coef._ttm_fold), self._ptr(Xs), Xs.shape[1], N, int(k), int(k) + 1,
                                          self._ptr(Z), Z.shape[1], None, None, None, self._stream()))
        return Z[0, :N].cpu().numpy()

    def basis(self, k, which, x=None):
        """Basis matrices of component k on standardised samples (None = training):
        which 'nonmon' | 'mon' | 'der_mon'  (the reference's fun_nonmon / fun_mon / der_fun_mon)."""
        sel = {'nonmon': 0, 'mon': 1, 'der_mon': 2}[which]
        if x is None:
            Xs, N = self._Xs, self._N
        else:
            Xs, N = self._import(np.asarray(x), False), np.asarray(x).shape[0]
        m = int(self._cm.n_nm[k] if sel == 0 else self._cm.n_mon[k])
        if m == 0:
            return None
        out = self._empty(m, N)
        _capi.check(self._lib.ttm_basis(self._pp, int(k), sel, self._ptr(Xs), Xs.shape[1], N, self._ptr(out), N,
                                        self._stream()))
        return self._export(out, N, 0, m, False)

    # ------------------------------------------------------------------------
    # densities (separable maps only, as the reference)
    # ------------------------------------------------------------------------

    def _log_determinant_raw(self, X_full, skip_in_std):
        """sum_k log( dPsi_mon,k(x_raw) c_k / sigma ) with the derivative basis
        evaluated on the UN-standardised samples (reference behaviour,
        TM:2627 / TM:2695) and sigma = X_std[k (+ skip)] (TM:2638 vs TM:2706)."""
        N = X_full.shape[0]
        Xraw = self._import(X_full, False)
        off = self.skip_dimensions if skip_in_std else 0
        sigma = self._to_dev(np.asarray(self.X_std[off:off + self.D], dtype=float))
        coef = self._pack_coeffs()
        ld = self._empty(N)
        _capi.check(self._lib.ttm_forward(self._pp, self._ptr(coef), self._ptr(coef._ttm_fold), self._ptr(Xraw), Xraw.shape[1], N, 0, self.D, None, N,
                                          self._ptr(ld), self._ptr(sigma), None, self._stream()))
        return ld

    def evaluate_pullback_density(self, X, X_star=None):
        """TM:2646-2712."""
        assert self.monotonicity == "separable monotonicity", \
            "evaluate_pushforward_density is currently only implemented for monotonicity = 'separable monotonicity'."
        if X_star is not None:
            X = np.column_stack((X_star, X))
        X = np.asarray(X, dtype=float)
        Xs, N = self._samples_for(X)
        coef = self._pack_coeffs()
        ss = self._empty(N)
        _capi.check(self._lib.ttm_forward(self._pp, self._ptr(coef), self._ptr(coef._ttm_fold), self._ptr(Xs), Xs.shape[1], N, 0, self.D,
                                          None, N, None, None, self._ptr(ss), self._stream()))
        ld = self._log_determinant_raw(X, skip_in_std=False)
        log_ref = -0.5 * (self.D * np.log(2 * np.pi) + ss.cpu().numpy())
        return np.exp(log_ref + ld.cpu().numpy())

    def evaluate_pushforward_density(self, Z, log_target_pdf, X_star=None):
        """TM:2569-2644."""
        assert self.monotonicity == "separable monotonicity", \
            "evaluate_pushforward_density is currently only implemented for monotonicity = 'separable monotonicity'."
        X = self.inverse_map(Z, X_star)
        log_target_densities = log_target_pdf(X)
        if X_star is not None:
            X = np.column_stack((X_star, X))
        ld = self._log_determinant_raw(np.asarray(X, dtype=float), skip_in_std=True)
        return np.exp(log_target_densities - ld.cpu().numpy())

    # ------------------------------------------------------------------------
    # inverse map
    # ------------------------------------------------------------------------

    def inverse_map(self, Z, X_star=None):
        """TM:3639-3796: sequential inversion of the components; the three
        conditioning shapes of the reference."""
        torch = _torch()
        Z = np.asarray(Z, dtype=float)                   # (never written to: no copy)
        N = Z.shape[0]
        d = self._cm.d_cols
        skip = self.skip_dimensions
        if X_star is None:
            k0, E, Xstar_cols = 0, 0, None
        else:
            X_star = np.asarray(X_star, dtype=float)
            if X_star.shape[-1] == skip:
                k0, E, Xstar_cols = 0, skip, X_star
            elif skip == 0:
                E = X_star.shape[-1]
                k0, Xstar_cols = E, X_star
                if E + Z.shape[-1] != self.D:
                    raise ValueError('X_star and Z must together have %d columns' % self.D)
            else:
                raise UnboundLocalError("local variable 'X' referenced before assignment")   # as the reference
        k1 = self.D
        ncomp = k1 - k0
        if Z.shape[-1] < ncomp:
            raise IndexError('Z has %d columns, %d are needed' % (Z.shape[-1], ncomp))
        coef = self._pack_coeffs()
        table = self.alternate_root_finding and self.monotonicity.lower() == 'separable monotonicity'
        if table and Xstar_cols is None and Z.shape[1] == ncomp and Z.flags.c_contiguous and self._pipe_ok(N):
            # rows are independent in the table inverse (no sample-0 guard): chunks of rows stream through
            # layout change -> lookups -> layout change back while their neighbours cross PCIe
            st = self._stream()
            Xs, Zs = self._cols(d, N, zero=True), self._cols(ncomp, N)
            self._inverse_table(coef, k0, k1, Zs, Xs, 0)            # (tables of this coefficient vector: built once, no rows)
            mean = self._mean_d if self.standardize_samples else None
            sd = self._std_d if self.standardize_samples else None

            def stage(rows, r0, n):
                _capi.check(self._lib.ttm_import(self._ptr(rows), n, ncomp, None, None, self._ptr(Zs, r0), Zs.shape[1], st))
                self._inverse_table(coef, k0, k1, Zs, Xs, n, row0=r0)
                out = self._empty(n, d)
                _capi.check(self._lib.ttm_export(self._ptr(Xs, r0), Xs.shape[1], n, 0, d, self._ptr(mean), self._ptr(sd), self._ptr(out), st))
                return out
            return self._host_pipeline(Z, d, stage)[:, skip:]
        Xs = self._cols(d, N, zero=True)
        if Xstar_cols is not None and E > 0:
            cols = np.array(Xstar_cols, dtype=float, copy=True)
            if self.standardize_samples:
                cols -= self.X_mean[:E]
                cols /= self.X_std[:E]
            Xs[:E, :N].copy_(torch.from_numpy(np.ascontiguousarray(cols.T)))
        # the layout change runs on the device, into a matrix with a padded (even) leading dimension: what the large-ensemble
        # kernels need (a host transpose gave odd ensemble sizes an odd leading dimension, i.e. the generic kernel)
        Zrow = self._to_dev(np.ascontiguousarray(Z[:, :ncomp]))
        Zs = self._cols(ncomp, N)
        _capi.check(self._lib.ttm_import(self._ptr(Zrow), N, ncomp, None, None, self._ptr(Zs), Zs.shape[1], self._stream()))
        if table:
            self._inverse_table(coef, k0, k1, Zs, Xs, N)
        else:
            self._inverse_bisect(coef, k0, k1, Zs, Xs, N)
        X = self._export(Xs, N, 0, d, self.standardize_samples)
        return X[:, skip:]

    def _inverse_table(self, coef, k0, k1, Zs, Xs, N, resolution=1001, start_distance=10, row0=0):
        """TM:3987-4084 for all components: tabulate, index and look up on the device.  interp1d sorts its
        abscissae (stable) first; monotone tables are already sorted, which the index kernel verifies -
        only an unsorted table (flat, noisy tails) takes the host detour that applies the sort."""
        torch = _torch()
        ncomp = k1 - k0
        nb = self._inv_nb()
        self._inverse_seen = True
        self._ensure_pts(resolution, start_distance)
        st = self._stream()
        trunc = 1 if self.root_search_truncation else 0
        # the tables depend on the coefficients only: they are built once per packed coefficient vector and kept
        # with it (like its folded coefficients), so repeated inversions with the same map are lookups only
        cache = getattr(coef, '_ttm_tables', None)
        if cache is None:
            cache = coef._ttm_tables = {}
        tkey = (k0, k1, resolution, start_distance, nb)
        if tkey not in cache:
            out_d = self._empty(ncomp, resolution)
            tmin_d, tmax_d = self._empty(ncomp), self._empty(ncomp)
            bkt_d = self._empty(ncomp, nb + 1, dtype=torch.int32)
            uns_d = self._empty(ncomp, dtype=torch.int32)
            img_d = None
            if resolution <= 2048:
                # table + its index (range, sortedness, bucket index, resident-table images) in one launch
                img_d = self._table_images(k0, k1, resolution, nb)
                _capi.check(self._lib.ttm_inverse_table_build_index(self._pp, self._ptr(coef), self._ptr(coef._ttm_fold), k0, k1,
                                                                    self._ptr(self._pts_d), resolution, nb, self._ptr(out_d),
                                                                    self._ptr(tmin_d), self._ptr(tmax_d),
                                                                    ctypes.c_void_p(bkt_d.data_ptr()), ctypes.c_void_p(uns_d.data_ptr()), None,
                                                                    self._ptr(img_d), st))
            else:
                _capi.check(self._lib.ttm_inverse_table_build(self._pp, self._ptr(coef), self._ptr(coef._ttm_fold), k0, k1,
                                                              self._ptr(self._pts_d), resolution, self._ptr(out_d), st))
                _capi.check(self._lib.ttm_inverse_table_index(self._ptr(out_d), ncomp, resolution, nb, self._ptr(tmin_d),
                                                              self._ptr(tmax_d), ctypes.c_void_p(bkt_d.data_ptr()),
                                                              ctypes.c_void_p(uns_d.data_ptr()), st))
            cache[tkey] = (out_d, tmin_d, tmax_d, bkt_d, int(uns_d.cpu().max().item()) == 0, img_d)    # (one copy of D flags, no reduction launch)
        out_d, tmin_d, tmax_d, bkt_d, is_sorted, img_d = cache[tkey]
        if N == 0:
            return                                      # (tables only)
        if is_sorted:
            _capi.check(self._lib.ttm_inverse_table(self._pp, self._ptr(coef), self._ptr(coef._ttm_fold), k0, k1,
                                                    self._ptr(Zs, row0), Zs.shape[1], self._ptr(Xs, row0), Xs.shape[1], N, self._ptr(out_d),
                                                    self._ptr(self._pts_d), 0, resolution, self._pts_affine, self._ptr(tmin_d),
                                                    self._ptr(tmax_d), ctypes.c_void_p(bkt_d.data_ptr()), nb, trunc, self._ptr(img_d),
                                                    0 if img_d is None else img_d.numel() // ncomp, st))
            return
        # rare: some table is not non-decreasing -> reproduce interp1d's stable sort on the host and redo
        out = out_d.cpu().numpy()
        order = np.argsort(out, axis=1, kind='mergesort')
        tab_x = np.take_along_axis(out, order, axis=1)
        tmin, tmax = np.min(out, axis=1), np.max(out, axis=1)
        with np.errstate(invalid='ignore'):
            edges = tmin[:, None] + np.arange(nb + 1)[None, :] * ((tmax - tmin) / nb)[:, None]
        bkt = np.stack([np.searchsorted(tab_x[i], edges[i], side='left') for i in range(ncomp)]).astype(np.int32)
        bkt[:, 0], bkt[:, -1] = 0, resolution
        tab_x_d, tab_y_d = self._to_dev(tab_x), self._to_dev(np.ascontiguousarray(self._pts[order]))
        bkt_d = self._to_dev(bkt, dtype=torch.int32)
        tmin_d, tmax_d = self._to_dev(tmin), self._to_dev(tmax)
        _capi.check(self._lib.ttm_inverse_table(self._pp, self._ptr(coef), self._ptr(coef._ttm_fold), k0, k1,
                                                self._ptr(Zs, row0), Zs.shape[1], self._ptr(Xs, row0), Xs.shape[1], N, self._ptr(tab_x_d),
                                                self._ptr(tab_y_d), resolution, resolution, None, self._ptr(tmin_d),
                                                self._ptr(tmax_d), ctypes.c_void_p(bkt_d.data_ptr()), nb, trunc, None, 0, st))

    def _inverse_bisect(self, coef, k0, k1, Zs, Xs, N):
        """TM:3798-3985.  Samples 1..N-1 run to convergence and record the largest
        midpoint-iteration count per component; global sample 0 is then replayed
        with that count as its cap, which is what the reference's
        ``while np.sum(indices) > 0`` guard (TM:3952) does to it.
        With ``root_finder = 'newton'`` (an extension, off by default): safeguarded Newton steps inside the same
        bracket and with the same stopping rule - the same roots to |S - z| <= 1e-9 in a fifth of the evaluations,
        not the reference's last midpoints and without its sample-0 quirk."""
        torch = _torch()
        ncomp = k1 - k0
        iters = self._zeros(ncomp, dtype=torch.int32)
        if self.root_finder == 'newton':
            _capi.check(self._lib.ttm_inverse_newton(self._pp, self._ptr(coef), self._ptr(coef._ttm_fold), k0, k1, self._ptr(Zs), Zs.shape[1],
                                                     self._ptr(Xs), Xs.shape[1], N, ctypes.c_void_p(iters.data_ptr()), self._stream()))
            if self.verbose and int(iters.max().item()) >= 100:
                print('WARNING: root search stopped at maximum iterations.')
            return
        dist = self._dist()
        owns_first = dist is None or dist.get_rank() == 0
        first = 1 if owns_first else 0
        if N - first > 0:
            _capi.check(self._lib.ttm_inverse_bisect(self._pp, self._ptr(coef), self._ptr(coef._ttm_fold), k0, k1, self._ptr(Zs, first), Zs.shape[1],
                                                     self._ptr(Xs, first), Xs.shape[1], N - first,
                                                     ctypes.c_void_p(iters.data_ptr()), None, self._stream()))
        self._allreduce(iters, op='max')
        if owns_first:
            dummy = self._zeros(ncomp, dtype=torch.int32)
            _capi.check(self._lib.ttm_inverse_bisect(self._pp, self._ptr(coef), self._ptr(coef._ttm_fold), k0, k1, self._ptr(Zs), Zs.shape[1], self._ptr(Xs), Xs.shape[1], 1,
                                                     ctypes.c_void_p(dummy.data_ptr()), ctypes.c_void_p(iters.data_ptr()),
                                                     self._stream()))
        if self.verbose and int(iters.max().item()) >= 100:
            print('WARNING: root search stopped at maximum iterations.')

    # ------------------------------------------------------------------------
    # objective functions
    # ------------------------------------------------------------------------

    def _device_sums(self, k, coef_k):
        """One fused objective+gradient reduction over the (local) training samples."""
        nout = 1 + int(self._cm.n_mon[k]) + (0 if self.monotonicity.lower() == 'separable monotonicity'
                                              else int(self._cm.n_nm[k]))
        work = self._workspace(self._lib.ttm_reduce_work_size(nout))
        coef_k = np.ascontiguousarray(coef_k, dtype=float)
        if self._dist() is None and len(coef_k) <= HOSTCOEF_MAX:
            # host-driven optimiser: coefficients as kernel arguments, result written to pinned host memory,
            # one stream synchronisation per evaluation (no H2D / D2H copies, two launches)
            torch = _torch()
            self._objective_launch(k, coef_k, work)
            self._sync_stream()
            return self._obj_out[:nout].numpy().copy()
        out = self._empty(nout)
        ck = self._to_dev(coef_k)
        _capi.check(self._lib.ttm_objective(self._pp, int(k), self._ptr(ck), self._ptr(self._Xs), self._Xs.shape[1],
                                            self._N, self._ptr(work), self._ptr(out), self._stream()))
        self._allreduce(out)
        return out.cpu().numpy()

    def _objective_launch(self, k, coef_k, work=None):
        """The launches of one objective + gradient evaluation of component k (ttm_objective_host: sums into pinned host
        memory), no synchronisation - what `_device_sums` waits for, and what bench.py times."""
        torch = _torch()
        if work is None:
            nout = 1 + int(self._cm.n_mon[k]) + (0 if self.monotonicity.lower() == 'separable monotonicity' else int(self._cm.n_nm[k]))
            work = self._workspace(self._lib.ttm_reduce_work_size(nout))
        if getattr(self, '_obj_out', None) is None:
            pin = self._dev.type == 'cuda'
            self._obj_out = torch.zeros(256, dtype=torch.float64, pin_memory=pin)
            self._obj_cnt = self._zeros(16, dtype=torch.int32)
        _capi.check(self._lib.ttm_objective_host(self._pp, int(k), ctypes.c_void_p(coef_k.ctypes.data), self._ptr(self._Xs),
                                                 self._Xs.shape[1], self._N, self._ptr(work),
                                                 ctypes.c_void_p(self._obj_cnt.data_ptr()),
                                                 ctypes.c_void_p(self._obj_out.data_ptr()), self._stream()))

    def _regularization_terms(self, k, div, cn, cm):
        """TM:3382-3431 and TM:3575-3633."""
        if self.regularization is None:
            return 0.0, 0.0
        if type(self.regularization) != str:
            raise ValueError("The variable 'regularization' must be either None, 'l1', or 'l2'.")
        r = self.regularization.lower()
        if r not in ('l1', 'l2'):
            raise ValueError("regularization_type must be either 'l1' or 'l2'.")
        lam = self.regularization_lambda
        if np.isscalar(lam):
            ln, lm = lam, lam
        elif type(lam) == list:
            ln, lm = np.asarray(lam[k][:div]), np.asarray(lam[k][div:])
        else:
            raise ValueError("Data type of regularization_lambda not understood. Must be either scalar or list.")
        if r == 'l1':
            return (np.sum(lm * np.abs(cm)) + np.sum(ln * np.abs(cn)),
                    np.concatenate((ln * np.sign(cn), lm * np.sign(cm))))
        return (np.sum(lm * cm ** 2) + np.sum(ln * cn ** 2), np.concatenate((ln * 2 * cn, lm * 2 * cm)))

    def _objective_and_gradient(self, coeffs, k, div):
        if self.monotonicity.lower() != 'integrated rectifier':
            raise Exception("objective_function is defined for monotonicity = 'integrated rectifier'; separable maps "
                            "are optimised through the reduced objective of optimize()")
        if coeffs is None:
            coeffs = np.concatenate((self.coeffs_nonmon[k], self.coeffs_mon[k]))
            div = len(self.coeffs_nonmon[k])
        coeffs = np.asarray(coeffs, dtype=float)
        key = (int(k), coeffs.tobytes())
        if self._obj_cache is None or self._obj_cache[0] != key:
            sums = self._device_sums(k, coeffs)
            cn, cm = coeffs[:div], coeffs[div:]
            rJ, rG = self._regularization_terms(k, div, cn, cm)
            self._obj_cache = (key, sums[0] / self._Nglobal + rJ, sums[1:] / self._Nglobal + rG)
        return self._obj_cache[1], self._obj_cache[2]

    def objective_function(self, coeffs, k, div=0):
        """TM:3300-3433."""
        return self._objective_and_gradient(coeffs, k, div)[0]

    def objective_function_jacobian(self, coeffs, k, div=0):
        """TM:3435-3635."""
        return np.array(self._objective_and_gradient(coeffs, k, div)[1], copy=True)

    # ------------------------------------------------------------------------
    # optimisation
    # ------------------------------------------------------------------------

    def _gram(self, k):
        m = int(self._cm.n_nm[k] + self._cm.n_mon[k])
        work = self._workspace(self._lib.ttm_reduce_work_size(m * m))
        out = self._empty(m * m)
        _capi.check(self._lib.ttm_gram(self._pp, int(k), self._ptr(self._Xs), self._Xs.shape[1], self._N, self._ptr(work),
                                       self._ptr(out), self._stream()))
        self._allreduce(out)
        return out.cpu().numpy().reshape(m, m)

    def _gram_many(self, K, before_read=None, launch_only=False):
        """Gram matrices of several components: the launches back to back, ONE all-reduce and ONE device-to-host
        copy for all of them."""
        torch = _torch()
        # (sizes, the device vector and its page-locked mirror are kept per set of components: the filter asks for the same
        # matrices in every update; the copy into page-locked memory needs no staging copy inside the runtime)
        memo = getattr(self, '_gram_memo', None)
        sizes = [int(self._cm.n_nm[k] + self._cm.n_mon[k]) for k in K]
        key = (tuple(int(k) for k in K), tuple(sizes))
        if memo is None or memo[0] != key:
            offs = [int(o) for o in np.concatenate(([0], np.cumsum([m * m for m in sizes])))]
            out = self._empty(offs[-1])
            pinned = torch.empty(offs[-1], dtype=torch.float64, pin_memory=self._dev.type == 'cuda')
            memo = self._gram_memo = (key, sizes, offs, out, pinned)
        _, sizes, offs, out, pinned = memo
        # (launched ahead by the reset that placed the special terms - _gram_ahead - and still good: same samples, same constants)
        token = (key, id(self._Xs), int(self._N), getattr(self, '_dpar_version', 0))
        ahead = getattr(self, '_gram_inflight', None)
        self._gram_inflight = None
        if launch_only:
            self._gram_inflight = token
        elif ahead == token:
            if before_read is not None:
                before_read()
            self._sync_stream()
            host = pinned.numpy().copy()
            return {k: host[o:o + m * m].reshape(m, m) for k, o, m in zip(key[0], offs, sizes)}
        work = self._workspace(self._lib.ttm_reduce_work_size(max(m * m for m in sizes)))
        st = self._stream()
        ks = np.asarray(key[0], dtype=np.int32)
        for c0 in range(0, len(ks), 8):                 # (ttm_gram_many: up to 8 components per launch)
            kc = np.ascontiguousarray(ks[c0:c0 + 8])
            rc = self._lib.ttm_gram_many(self._pp, ctypes.c_void_p(kc.ctypes.data), len(kc), self._ptr(self._Xs), self._Xs.shape[1],
                                         self._N, self._ptr(work), self._ptr(out, offs[c0]), st) if len(kc) > 1 else _capi.TTM_E_UNSUPPORTED
            if rc == _capi.TTM_E_UNSUPPORTED:           # (one component, or shapes the batched launch does not take)
                for k, o in zip(kc.tolist(), offs[c0:c0 + 8]):
                    _capi.check(self._lib.ttm_gram(self._pp, k, self._ptr(self._Xs), self._Xs.shape[1], self._N, self._ptr(work),
                                                   self._ptr(out, o), st))
            else:
                _capi.check(rc)
        self._allreduce(out)
        pinned.copy_(out, non_blocking=True)
        if launch_only:
            return None
        if before_read is not None:
            before_read()                               # (launches that do not need the matrices: they run while the host waits)
        self._sync_stream()
        host = pinned.numpy().copy()
        return {k: host[o:o + m * m].reshape(m, m) for k, o, m in zip(key[0], offs, sizes)}

    def _gram_ahead(self):
        """The Gram matrices the next optimize() will ask for, launched as soon as the special-term constants of a reset are on
        the device - in front of the host's work on the spline geometry, which the matrices do not depend on (the filter: the
        kernels run while the host prepares the U-form tables).  Only when the last optimize() took all its components as one
        batch of the native loops and nothing about the map's structure has changed since; a reset that is not followed by
        optimize() has launched them in vain."""
        hint = getattr(self, '_gram_hint', None)
        if hint is None or self._dist() is not None or hint[1] != id(self._cm):
            return
        try:
            self._gram_many(list(hint[0]), launch_only=True)
            self._bases_inflight = None
            if hint[2] is not None and hint[3] == int(self._N) and hint[4] == int(self._Xs.shape[1]):
                # (the cached derivative bases of the same batch: the closure of the optimize() that left the hint launches them
                # into the buffers it keeps - laid out for the same number of samples)
                hint[2](ahead=True)
        except Exception:                               # noqa: BLE001  (a hint, never an error: optimize() launches them itself)
            self._gram_inflight = None
            self._bases_inflight = None

    def _ahead_token(self):
        """What the kernels launched ahead by a reset depend on: the samples, their number, the special-term constants, the map."""
        return (id(self._cm), id(self._Xs), int(self._N), getattr(self, '_dpar_version', 0))

    def separable_setup(self, k, G=None):
        """The reduced separable problem of TM:2959-3050 from the Gram matrix of
        [Psi_nonmon | Psi_mon] (one device pass instead of QR / inv on N x m
        matrices).  Returns (A, solve_nonmon) with solve_nonmon(c_mon) -> c_nonmon
        (TM:3148-3169)."""
        n_nm = int(self._cm.n_nm[k])
        if n_nm == 0:
            raise ValueError('separable monotonicity needs at least one nonmonotone term (TM:2966)')
        if G is None:
            G = self._gram(k)
        Gnn, Gnm, Gmm = G[:n_nm, :n_nm], G[:n_nm, n_nm:], G[n_nm:, n_nm:]
        N = self._Nglobal
        if self.regularization is None:
            sol = self._normal_solve(Gnn, Gnm, 0.0)
            if sol is None:
                return self._separable_setup_qr(k)                   # nearly collinear nonmonotone basis: the reference's QR
            A = (Gmm - Gnm.T @ sol) / N
            A = (A + A.T) / 2
            return A, (lambda c: -(sol @ c))
        if self.regularization.lower() == 'l2':
            lam = self.regularization_lambda
            if self.native_optimizer and lam > 0:           # the same arithmetic in the library (no NumPy: 15 against 75 us)
                Gc = np.ascontiguousarray(G, dtype=float)
                m = Gc.shape[0] - n_nm
                A, sol2 = np.empty((m, m)), np.empty((n_nm, m))
                rc = self._lib.ttm_separable_reduce_l2(ctypes.c_void_p(Gc.ctypes.data), n_nm, m, float(lam), ctypes.c_void_p(A.ctypes.data),
                                                       ctypes.c_void_p(sol2.ctypes.data))
                if rc == 0:
                    return A, (lambda c: -(sol2 @ c))
            Gm = self._normal_solve(Gnn, Gnm, lam)
            dd = Gmm - Gnm.T @ Gm - Gm.T @ Gnm + Gm.T @ Gnn @ Gm
            A = dd / 2 + lam * (Gm.T @ Gm + np.identity(Gm.shape[-1]))      # no 1/N: as TM:3040-3050
            A = (A + A.T) / 2
            sol2 = self._normal_solve(Gnn, Gnm, 2 * lam)
            return A, (lambda c: -(sol2 @ c))
        raise ValueError("separable monotonicity supports regularization None or 'l2' (TM:2959, 3021)")

    # condition number (of the diagonally equilibrated Gram matrix) above which the normal equations are not trusted:
    # they square the condition number of Psi_nonmon, the reference's Householder QR (TM:2966-2975) does not
    GRAM_COND_MAX = 1e11

    @classmethod
    def _normal_solve(cls, Gnn, Gnm, ridge):
        """(Gnn + ridge I)^-1 Gnm by Cholesky on the diagonally equilibrated matrix with one step of iterative
        refinement; None when the equilibrated matrix is too ill-conditioned for the normal equations (ridge == 0
        only: a ridge bounds the condition number by itself)."""
        # (LAPACK's dpotrf / dpotrs called directly - what scipy.linalg.cho_factor / cho_solve call, without their argument
        # checks: 15 instead of 42 us per solve, six solves per update of the filter)
        potrf, potrs = cls._lapack_chol()
        M = Gnn + ridge * np.identity(Gnn.shape[0])
        d = np.sqrt(np.diag(M))
        if not np.all(d > 0) or not np.all(np.isfinite(d)):
            return None if ridge == 0.0 else np.linalg.solve(M, Gnm)
        Ms = M / d[:, None] / d[None, :]
        if ridge == 0.0:
            w = np.linalg.eigvalsh(Ms)
            if not (w[0] > 0) or w[-1] / w[0] > cls.GRAM_COND_MAX:
                return None
        if not np.all(np.isfinite(Ms)) or not np.all(np.isfinite(Gnm)):
            raise ValueError('array must not contain infs or NaNs')
        cf, info = potrf(Ms, lower=False, overwrite_a=False, clean=False)
        if info != 0:
            return None if ridge == 0.0 else np.linalg.solve(M, Gnm)
        rhs = Gnm / d[:, None]
        y, _ = potrs(cf, rhs, lower=False, overwrite_b=False)
        dy, _ = potrs(cf, rhs - Ms @ y, lower=False, overwrite_b=False)
        y += dy                                                       # one refinement step
        return y / d[:, None]

    _LAPACK_CHOL = None

    @classmethod
    def _lapack_chol(cls):
        if cls._LAPACK_CHOL is None:
            import scipy.linalg.lapack as lapack
            transport_map._LAPACK_CHOL = (lapack.dpotrf, lapack.dpotrs)
        return cls._LAPACK_CHOL

    def _separable_setup_qr(self, k):
        """The reduced problem exactly as the reference forms it (TM:2966-2975, 3152-3157): Householder QR of Psi_nonmon
        on the host.  Fallback for nonmonotone bases whose Gram matrix is too ill-conditioned (the basis matrices come
        off the device once: N x m doubles)."""
        if self._dist() is not None:
            raise np.linalg.LinAlgError('component %d: the nonmonotone basis is numerically rank deficient (equilibrated Gram '
                                        'matrix beyond cond %.0e) and samples are sharded over ranks' % (k, self.GRAM_COND_MAX))
        Pn, Pm = self.basis(k, 'nonmon'), self.basis(k, 'mon')
        Q, R = np.linalg.qr(Pn, mode='reduced')
        A = (Pm.T @ Pm - (Q.T @ Pm).T @ (Q.T @ Pm)) / self._Nglobal
        A = (A + A.T) / 2
        QtPm = Q.T @ Pm
        return A, (lambda c: -np.linalg.solve(R, QtPm @ c))

    def _sep_cache_begin(self, k):
        """Cache dPsi_mon of component k on the device for the duration of its optimisation (what the reference's
        precalculate() keeps as der_Psi_mon): every L-BFGS-B evaluation is then one streaming launch."""
        self._sep_cache = None
        m = int(self._cm.n_mon[k])
        if m < 1 or m > 16:
            return
        torch = _torch()
        dpsi = self._cols(m, self._N)
        _capi.check(self._lib.ttm_basis(self._pp, int(k), 2, self._ptr(self._Xs), self._Xs.shape[1], self._N, self._ptr(dpsi),
                                        dpsi.shape[1], self._stream()))
        if getattr(self, '_obj_out', None) is None:
            pin = self._dev.type == 'cuda'
            self._obj_out = torch.zeros(256, dtype=torch.float64, pin_memory=pin)
            self._obj_cnt = self._zeros(16, dtype=torch.int32)
        self._sep_cache = (int(k), dpsi)

    def _sep_cache_end(self):
        self._sep_cache = None

    def separable_objective(self, coeffs_mon, A, k):
        """TM:2978-3018: (objective, gradient) of the reduced problem."""
        c = np.ascontiguousarray(coeffs_mon, dtype=float)
        cache = getattr(self, '_sep_cache', None)
        if cache is not None and cache[0] == int(k) and self._dist() is None:
            m = len(c)
            self._sep_objective_launch(c)
            self._sync_stream()
            sums = self._obj_out[:1 + m].numpy().copy()
        else:
            sums = self._device_sums(k, np.concatenate((np.zeros(int(self._cm.n_nm[k])), c)))
        N = self._Nglobal
        b = self.delta * np.sum(A, axis=-1)
        Ax = A @ c
        objective = c @ Ax / 2 - sums[0] / N + np.inner(c, b)
        grad = Ax - sums[1:] / N + b
        return objective, grad

    def _sep_objective_launch(self, c):
        """The launch(es) of one reduced-objective evaluation on the cached derivative basis (ttm_objective_sep_cached),
        no synchronisation."""
        cache, m = self._sep_cache, len(c)
        work = self._workspace(self._lib.ttm_reduce_work_size(1 + m))
        _capi.check(self._lib.ttm_objective_sep_cached(self._ptr(cache[1]), cache[1].shape[1], self._N, m,
                                                       ctypes.c_void_p(c.ctypes.data), float(self.delta), self._ptr(work),
                                                       ctypes.c_void_p(self._obj_cnt.data_ptr()),
                                                       ctypes.c_void_p(self._obj_out.data_ptr()), self._stream()))

    def _sep_objective_launch_sent(self, c):
        """The launch of one evaluation as the native L-BFGS-B loops make it for ensembles of up to 131 072 samples
        (ttm_objective_sep_cached_sent: self-validating partial sums and results), no synchronisation - what bench.py times; the
        sums land in self._obj_out.  False: the library declines (a larger grid, option sep_sentinel = 0): _sep_objective_launch."""
        cache, m = self._sep_cache, len(c)
        key = (m, int(self._N))
        st = getattr(self, '_sep_sent_work', None)
        if st is None or st[0] != key:
            work = self._empty(int(self._lib.ttm_reduce_work_size(1 + m)))
            if self._lib.ttm_sentinel_fill(self._ptr(work), m, self._N, self._stream()) != 0:
                return False
            st = self._sep_sent_work = (key, work)          # (armed once: every evaluation leaves the rows armed)
        _capi.check(self._lib.ttm_objective_sep_cached_sent(self._ptr(cache[1]), cache[1].shape[1], self._N, m,
                                                            ctypes.c_void_p(c.ctypes.data), float(self.delta), self._ptr(st[1]),
                                                            ctypes.c_void_p(self._obj_out.data_ptr()), self._stream()))
        return True

    def _respecify(self, monotone, nonmonotone):
        """New term lists on the resident ensemble (TM:432-438: function_constructor_alternative + precalculate)."""
        self.monotone = copy.deepcopy(monotone)
        self.nonmonotone = copy.deepcopy(nonmonotone)
        self._build_program(self._cm.d_cols)
        self._obj_cache = None
        self.determine_special_term_locations()

    def adapt_map(self, coeffs={}, maxorder_mon=10, maxorder_nonmon=10, threshold_sw=0.1, threshold_prec=0.1,
                  sequential_updates=False, map_finished=None):
        """Greedy growth of the term lists (reference behaviour: TM:373-636 for adaptation_map_type = 'separable',
        TM:4575-4950 for 'cross-terms').

        'separable' runs two phases on the resident ensemble, each round = re-specify (coefficients back to
        coeffs_init, special terms re-placed) -> optimize -> map, i.e. the hot path; only the statistics on the N x D
        pushforward are host NumPy / SciPy:
          1. marginals: every component starts as the linear term [[k]]; while its pushforward marginal fails a
             Shapiro-Wilk test (p < threshold_sw) it gets one more 'iRBF k' per round, up to maxorder_mon;
          2. dependence: a pair (k, j < k) whose dependence statistic exceeds threshold_prec gets one more nonmonotone
             term in x_j per round ([j], then [j, j, 'HF'], [j, j, j, 'HF'], ...); a pair that falls below the threshold
             once is closed for good.  The statistic is the standardised precision of the pushforward in the first
             round and its correlation afterwards."""
        if self.adaptation_map_type == 'cross-terms':
            return self.adaptation_cross_terms(*coeffs)                                                  # (TM:642)
        if self.adaptation_map_type != 'separable':
            raise Exception("Currently, only adaptation_map_type = 'cross-terms' is implemented.")      # (TM:648, sic)
        D = self.D
        spec = {'monotone': [[[k]] for k in range(D)], 'nonmonotone': [[[]] for _ in range(D)]}
        orders = np.zeros((D, D), dtype=int)
        orders[np.arange(D), np.arange(D)] = 1
        Z = self._adapt_marginals(spec, orders, maxorder_mon, threshold_sw)
        # the reference keeps both dependence matrices of the marginal fit (absolute correlation, absolute standardised
        # precision) as attributes; the second phase recomputes what it uses
        self.covmat = self._unit_diagonal(np.abs(np.cov(Z.T)))
        self.precmat = self._unit_diagonal(np.abs(np.linalg.inv(np.cov(Z.T))))
        closed = np.zeros((D, D), dtype=bool) if map_finished is None else map_finished
        self._adapt_dependence(spec, orders, closed, maxorder_nonmon, threshold_prec)
        self._respecify(spec['monotone'], spec['nonmonotone'])
        self.optimize()
        self.maporders = orders

    @staticmethod
    def _unit_diagonal(M):
        """M scaled to unit diagonal: M_ij / sqrt(M_ii M_jj) (columns first, then rows, as the reference divides)."""
        M = np.array(M, dtype=float, copy=True)
        root = np.sqrt(np.diag(M))
        M /= root[np.newaxis, :]
        M /= root[:, np.newaxis]
        return M

    def _fit_and_push(self, spec):
        """One adaptation round on the device: new term lists, optimisation of every component, pushforward of the
        training ensemble."""
        self._respecify(spec['monotone'], spec['nonmonotone'])
        self.optimize()
        return self.map()

    def _adapt_marginals(self, spec, orders, max_terms, p_accept):
        """Phase 1 of the separable adaptation (marginal Gaussianisation).  `orders[k, k + skip]` counts the monotone
        terms of component k; self.pvals_mat is not kept by the reference either (a local there)."""
        import scipy.stats
        D = self.D
        accepted = np.zeros(D, dtype=bool)
        rounds = 0
        while True:
            rounds += 1
            Z = self._fit_and_push(spec)
            p = np.array([scipy.stats.shapiro(Z[:, k]).pvalue for k in range(D)])
            accepted |= p >= p_accept                      # a component that passed once stays accepted
            for k in np.flatnonzero(~accepted):
                col = k + self.skip_dimensions
                if orders[k, col] < max_terms:
                    orders[k, col] += 1
                    spec['monotone'][k] += ['iRBF ' + str(k)]
            if accepted.all() or rounds >= max_terms - 1:
                return Z

    def _adapt_dependence(self, spec, orders, closed, max_rounds, threshold):
        """Phase 2 of the separable adaptation.  Any exception while the lists are being extended ends the phase, as
        in the reference (TM:614) - notably the list sort, which cannot order a term with the 'HF' marker against a
        longer all-integer prefix; what was appended before the exception stays."""
        D = self.D
        rounds = 0
        while True:
            rounds += 1
            Z = self._fit_and_push(spec)
            stop = False
            try:
                if rounds == 1:
                    stat = self._unit_diagonal(np.abs(np.linalg.inv(np.cov(Z.T))))
                else:
                    stat = np.corrcoef(Z.T)
                for k in range(D):
                    for j in range(k):
                        if stat[k, j] > threshold and not closed[k, j]:
                            orders[k, j] += 1
                            term = [j] * orders[k, j]
                            spec['nonmonotone'][k].append(term if orders[k, j] == 1 else term + ['HF'])
                        else:
                            closed[k, j] = True
                    spec['nonmonotone'][k].sort()
            except Exception:                      # noqa: BLE001
                stop = True
            if np.sum(closed) >= D * (D - 1) / 2:
                stop = True
            if rounds >= max_rounds:
                print("WARNING: Map adaptation stopped at maximum number of iterations.")
                stop = True
            if stop:
                return

    def _respecify_component(self, k, monotone_k, nonmonotone_k):
        """New term lists for component k only (the reference's function_constructor_alternative(k = k), TM:1304-1310):
        the other components keep their coefficients."""
        keep = ([np.array(c, copy=True) for c in self.coeffs_mon], [np.array(c, copy=True) for c in self.coeffs_nonmon])
        mon, non = copy.deepcopy(self.monotone), copy.deepcopy(self.nonmonotone)
        mon[k], non[k] = copy.deepcopy(monotone_k), copy.deepcopy(nonmonotone_k)
        self._respecify(mon, non)
        for j in range(self.D):
            if j != k:
                self.coeffs_mon[j], self.coeffs_nonmon[j] = keep[0][j], keep[1][j]

    def adaptation_cross_terms(self, increment=1E-6, chronicle=False):
        """TM:4575-4950 (integrated rectifier): per component a multi-index set grows one cell per round.  Active cells
        propose their axis neighbours; a proposal is admissible when all its lower neighbours are active (its proposal
        count, boundary coordinates counted as given, reaches the number of variables); every admissible cell is scored
        by a one-sided finite difference of the objective in its new coefficient, the best one is added and the
        component is re-optimised (L-BFGS-B on the objective alone, gradients by SciPy's finite differences, as the
        reference calls it).  Cell (i_0, ..., i_k) is the term [0]*i_0 + ... + [k]*i_k (+ 'HF'); cells with i_k > 0 are
        monotone terms.  The bookkeeping that carries coefficients over - positions in the order of np.where over the
        index set, not in [nonmonotone | monotone] order - is the reference's and is kept."""
        from scipy.optimize import minimize
        hf = self.polynomial_type.lower() == 'hermite function'

        def cell_term(cell):
            term = []
            for var, order in enumerate(cell):
                term += [int(var)] * int(order)
            return term + ['HF'] if (hf and len(term) > 0) else term

        def lists_of(M):
            mono, nonmono, proposed, original = [], [], [], []
            for pos, cell in enumerate(np.asarray(np.where(M != 0)).T):
                (proposed if M[tuple(cell)] < 0 else original).append(pos)
                (mono if cell[-1] > 0 else nonmono).append(cell_term(cell))
            return mono, nonmono, proposed, original

        history = {}
        for k in range(self.D):
            nvar = k + 1 + self.skip_dimensions
            M = np.zeros(tuple([self.adaptation_max_order + 1] * nvar), dtype=int)
            M[tuple([0] * nvar)] = 1
            M[tuple([0] * (nvar - 1) + [1])] = 1
            self.multi_index_matrix = M
            coeffs = np.concatenate((np.asarray(self.coeffs_nonmon[k], dtype=float), np.asarray(self.coeffs_mon[k], dtype=float)))
            div = len(self.coeffs_nonmon[k])
            opt = minimize(method='BFGS', fun=self.objective_function, jac=self.objective_function_jacobian, x0=coeffs,
                           args=(k, div))
            coeffs = np.array(opt.x, copy=True)
            self.coeffs_nonmon[k], self.coeffs_mon[k] = coeffs[:div].copy(), coeffs[div:].copy()
            rounds = 0
            history[k] = {}
            while True:
                rounds += 1
                for cell in np.asarray(np.where(M > 0)).T:
                    for ax in range(nvar):
                        for step in (-1, +1):
                            if 0 <= cell[ax] + step < self.adaptation_max_order + 1:
                                nb = list(cell)
                                nb[ax] += step
                                if M[tuple(nb)] <= 0:
                                    M[tuple(nb)] -= 1
                if len(np.asarray(np.where(M < 0)).T) == 0:
                    break
                for cell in np.asarray(np.where(M < 0)).T:
                    for val in cell:
                        if val == 0:
                            M[tuple(cell)] -= 1
                candidates = np.asarray(np.where(M <= -nvar)).T
                if self.verbose:
                    print(M)
                coeffs = np.concatenate((np.asarray(self.coeffs_nonmon[k], dtype=float), np.asarray(self.coeffs_mon[k], dtype=float)))
                obj_ref = self.objective_function(coeffs=coeffs, k=k, div=div)
                grads = np.zeros(len(candidates))
                for ci, cell in enumerate(candidates):
                    M[M < 0] = 0
                    M[tuple(cell)] = -1
                    mono, nonmono, _, orig = lists_of(M)
                    self._respecify_component(k, mono, nonmono)
                    trial = np.ones(len(nonmono) + len(mono)) * self.coeffs_init + increment
                    trial[orig] = copy.copy(coeffs)
                    div = len(nonmono)
                    grads[ci] = (self.objective_function(coeffs=trial, k=k, div=div) - obj_ref) / increment
                best = np.where(np.abs(grads) == np.max(np.abs(grads)))[0][0]
                M[M < 0] = 0
                added = candidates[best]
                M[tuple(added)] = -1
                mono, nonmono, _, orig = lists_of(M)
                M[tuple(added)] = 1
                start = np.ones(len(nonmono) + len(mono)) * self.coeffs_init
                start[orig] = copy.copy(coeffs)
                div = len(nonmono)
                self._respecify_component(k, mono, nonmono)
                opt = minimize(method='L-BFGS-B', fun=self.objective_function, x0=start, args=(k, div))
                coeffs = np.array(opt.x, copy=True)
                self.coeffs_nonmon[k], self.coeffs_mon[k] = coeffs[:div].copy(), coeffs[div:].copy()
                history[k][rounds] = dict(monotone=copy.deepcopy(self.monotone[k]), nonmonotone=copy.deepcopy(self.nonmonotone[k]),
                                          coeffs_nonmon=copy.copy(self.coeffs_nonmon[k]), coeffs_mon=copy.copy(self.coeffs_mon[k]),
                                          multi_index_matrix=copy.copy(M))
                if rounds >= self.adaptation_max_iterations:
                    break
        if chronicle:
            import pickle
            pickle.dump(history, open('dictionary_adaptation_chronicle.p', 'wb'))

    def _sep_objective_fast(self, A, k):
        """separable_objective(., A, k) for the optimiser's inner loop: everything that does not depend on the
        coefficient vector (pointers, stream handle, workspace, b = delta * rowsum(A)) is prepared once per component;
        one evaluation = one ctypes call, one stream synchronisation and O(m^2) host arithmetic.  Same operations on
        the same values as separable_objective (which stays the public method)."""
        cache = getattr(self, '_sep_cache', None)
        if cache is None or cache[0] != int(k) or self._dev.type != 'cuda' or self._dist() is not None:
            return None
        torch = _torch()
        dpsi = cache[1]
        m = int(self._cm.n_mon[k])
        work = self._workspace(self._lib.ttm_reduce_work_size(1 + m))
        stream = torch.cuda.current_stream()
        fn = self._lib.ttm_objective_sep_cached
        a_dpsi, a_ld, a_N, a_m = self._ptr(dpsi), dpsi.shape[1], self._N, m
        a_delta, a_work = float(self.delta), self._ptr(work)
        a_cnt, a_out = ctypes.c_void_p(self._obj_cnt.data_ptr()), ctypes.c_void_p(self._obj_out.data_ptr())
        a_stream = ctypes.c_void_p(stream.cuda_stream)
        out_np = self._obj_out.numpy()                       # (view of the pinned result buffer)
        N = self._Nglobal
        b = self.delta * np.sum(A, axis=-1)
        check, sync = _capi.check, stream.synchronize

        def fun(coeffs_mon, *_):
            c = np.ascontiguousarray(coeffs_mon, dtype=float)
            check(fn(a_dpsi, a_ld, a_N, a_m, ctypes.c_void_p(c.ctypes.data), a_delta, a_work, a_cnt, a_out, a_stream))
            sync()
            sums = out_np[:1 + m].copy()
            Ax = A @ c
            return c @ Ax / 2 - sums[0] / N + np.inner(c, b), Ax - sums[1:] / N + b
        return fun

    # separable components are minimised by the library's own L-BFGS-B loop (ttm_optimize_separable: no Python per
    # evaluation); False = scipy.optimize's loop driven from Python with the same device reductions
    native_optimizer = True

    # device entry points: read the per-vector checks (spline fit errors, table sortedness) at validate() instead of
    # behind every new coefficient vector (see validate())
    deferred_checks = False

    # map() / inverse_map() on host arrays of PIPE_MIN_ROWS rows or more: chunks of rows stream over PCIe, through the
    # kernels and back on three streams (see _host_pipeline); False = one copy in, one copy out
    host_pipeline = True

    def _optimize_separable_native(self, A, k, x0, bounds):
        """TM:3108-3114 for one component without leaving the library; None when the native loop does not apply
        (no cached derivative basis, or ranks that share samples without an RCCL / test-double communicator)."""
        cache = getattr(self, '_sep_cache', None)
        if not self.native_optimizer or cache is None or cache[0] != int(k):
            return None
        handle = None
        if self._dist() is not None:
            handle = comm.get(self._lib, force=self._dev.type != 'cuda')
            if handle is None:
                return None
        dpsi = cache[1]
        m = int(self._cm.n_mon[k])
        A = np.ascontiguousarray(A, dtype=float)
        b = np.ascontiguousarray(self.delta * np.sum(A, axis=-1))
        x = np.array(x0, dtype=float, copy=True)
        lb = np.array([-np.inf if lo is None else lo for lo, _ in bounds], dtype=float)
        ub = np.array([np.inf if hi is None else hi for _, hi in bounds], dtype=float)
        work = self._workspace(self._lib.ttm_reduce_work_size(1 + m))
        sums_dev = self._empty(1 + m) if handle is not None else None
        res = np.zeros(5)
        p = lambda a: ctypes.c_void_p(a.ctypes.data)                   # noqa: E731
        _capi.check(self._lib.ttm_optimize_separable(
            self._ptr(dpsi), dpsi.shape[1], self._N, m, p(A), p(b), float(self._Nglobal), float(self.delta), p(lb), p(ub), p(x),
            self._ptr(work), ctypes.c_void_p(self._obj_cnt.data_ptr()), self._ptr(sums_dev),
            ctypes.c_void_p(self._obj_out.data_ptr()), handle, self._stream(), 0, p(res)))

        class _Result:
            pass
        out = _Result()
        out.x, out.fun, out.nit, out.nfev, out.status = x, float(res[0]), int(res[2]), int(res[3]), int(res[4])
        return out

    # host threads of the batched component optimisation (each with a HIP stream of its own)
    optimizer_threads = 8
    # device memory the cached derivative bases of one batch may take (bytes)
    optimizer_batch_bytes = 8 << 30
    # True: components whose monotone terms are plain special terms of x_k recompute the derivative basis from the x_k
    # column in every evaluation instead of caching an N x m matrix (ttm_objective_sep_direct_marked).  Same bits and
    # no N x m matrices (1.3 GB at C5), but the erf-table gathers cost more than the bytes they save: optimize() at
    # C5 0.027 s against 0.021 s - the memory-lean variant, off by default
    direct_objective = False

    def _optimize_separable_batch(self, K):
        """TM:2746-2845 for separable maps: the components of K are independent problems (the reference hands them to
        a process pool).  Per batch: every Gram matrix in one pass, the reduced problems on the host, every derivative
        basis cached on the device, then ttm_optimize_separable_batch - the L-BFGS-B loops of the components side by
        side on host threads with a HIP stream each.  Returns {k: result} or None when the batched native path does
        not apply (the caller then walks the components one by one)."""
        if not self.native_optimizer or len(K) < 2 or self.optimizer_threads < 2 or self._dist() is not None:
            return None
        if any(int(self._cm.n_mon[k]) < 1 or int(self._cm.n_mon[k]) > 16 or int(self._cm.n_nm[k]) == 0 for k in K):
            return None
        torch = _torch()
        results = {}
        direct = {k: self._cm.sep_direct[k] if self.direct_objective else None for k in K}
        per_k = [0 if direct[k] is not None else int(self._cm.n_mon[k]) * self._Xs.shape[1] * 8 for k in K]
        start = 0
        while start < len(K):
            stop, used = start, 0
            while stop < len(K) and (stop == start or used + per_k[stop] <= self.optimizer_batch_bytes):
                used += per_k[stop]
                stop += 1
            batch = K[start:stop]
            start = stop
            n = len(batch)
            # (scratch of the native loops, kept between calls: reduction workspaces, ticket counters - the kernels leave them
            # zero -, page-locked result vectors and the cached derivative bases; the filter optimises the same components
            # over the same number of samples in every update)
            skey = (tuple(batch), int(self._N), int(self._Xs.shape[1]), tuple(int(self._cm.n_mon[k]) for k in batch))
            scr = getattr(self, '_sep_batch_scratch', None)
            if scr is None or scr[0] != skey:
                wsz = int(self._lib.ttm_reduce_work_size(17))
                scr = self._sep_batch_scratch = (skey, wsz, self._empty(n * wsz), self._zeros(n * 16, dtype=torch.int32),
                                                 torch.zeros(n * 32, dtype=torch.float64, pin_memory=self._dev.type == 'cuda'), {}, {}, {})
            _, wsz, work, counters, sums, dpsi_keep, armed, host_keep = scr

            def launch_bases(ahead=False):
                # the cached derivative bases: queued behind the Gram kernels, in front of the host's wait for the matrices
                # (or launched ahead with them by the reset that placed the special terms - _gram_ahead -, into THESE buffers)
                token = (self._ahead_token(), tuple(batch), id(dpsi_keep))
                if not ahead and getattr(self, '_bases_inflight', None) == token:
                    self._bases_inflight = None
                    return
                self._bases_inflight = token if ahead else None
                for k in batch:
                    if direct[k] is None:
                        m = int(self._cm.n_mon[k])
                        dpsi = dpsi_keep.get(k)
                        if dpsi is None or dpsi.shape[0] != m:
                            dpsi = dpsi_keep[k] = self._cols(m, self._N)
                        _capi.check(self._lib.ttm_basis(self._pp, int(k), 2, self._ptr(self._Xs), self._Xs.shape[1], self._N,
                                                        self._ptr(dpsi), dpsi.shape[1], self._stream()))
            grams = self._gram_many(batch, before_read=launch_bases)
            # (a reset of the same map launches these matrices - and the bases, once their buffers exist - ahead: _gram_ahead)
            self._gram_hint = (tuple(batch), id(self._cm), launch_bases if not any(direct[k] is not None for k in batch) else None,
                               int(self._N), int(self._Xs.shape[1])) if len(batch) == len(K) else None
            # the task structures and the host vectors they point to are kept with the scratch: values are written in place
            tasks = host_keep.get('tasks')
            fresh = tasks is None
            if fresh:
                tasks = host_keep['tasks'] = (_capi.ttm_sep_task * n)()
                host_keep['vec'] = {}
            keep = []
            # special-term kinds and constants of the components that recompute their basis: one upload for the batch
            kinds_all, pars_all, where = [], [], {}
            for k in batch:
                if direct[k] is not None:
                    base = int(self._cm.dpar_off[k])
                    where[k] = (len(kinds_all), len(pars_all))
                    for kind, p0 in direct[k][1]:
                        kinds_all.append(kind)
                        pars_all.extend(self._cm.dpar[base + p0:base + p0 + 5])
            kinds_d = self._to_dev(np.asarray(kinds_all, dtype=np.int32), dtype=torch.int32) if kinds_all else None
            pars_d = self._to_dev(np.asarray(pars_all, dtype=np.float64)) if pars_all else None
            for i, k in enumerate(batch):
                A, solve_nonmon = self.separable_setup(k, G=grams[k])
                m = int(self._cm.n_mon[k])
                dpsi = dpsi_keep[k] if direct[k] is None else None
                vec = host_keep['vec'].get(k)
                if vec is None or vec[0].shape != (m, m):
                    vec = host_keep['vec'][k] = (np.empty((m, m)), np.empty(m), np.empty(m),
                                                 np.array([-np.inf if v is None else v for v in self.optimization_constraints_lb[k]], dtype=float),
                                                 np.array([np.inf if v is None else v for v in self.optimization_constraints_ub[k]], dtype=float))
                    fresh = True
                Ab, b, x, lb, ub = vec
                Ab[...] = A
                A = Ab
                b[...] = self.delta * np.sum(A, axis=-1)
                x[...] = self.coeffs_mon[k]
                keep.append((A, b, x, lb, ub, dpsi, solve_nonmon))
                t = tasks[i]
                if not fresh and direct[k] is None:
                    t.dPsi, t.ldp = dpsi.data_ptr(), dpsi.shape[1]
                    t.armed = armed.get(k, 0)
                    continue                             # (every other pointer of the task is what it was)
                t.m = m
                if dpsi is not None:
                    t.dPsi, t.ldp = dpsi.data_ptr(), dpsi.shape[1]
                else:
                    t.dPsi, t.ldp = None, 0
                    t.xk = self._Xs.data_ptr() + 8 * int(direct[k][0]) * self._Xs.shape[1]
                    t.kinds = kinds_d.data_ptr() + 4 * where[k][0]
                    t.pars = pars_d.data_ptr() + 8 * where[k][1]
                t.A, t.b, t.lb, t.ub, t.x = A.ctypes.data, b.ctypes.data, lb.ctypes.data, ub.ctypes.data, x.ctypes.data
                t.armed = armed.get(k, 0)                # (the rows of partial sums this slice of `work` was left with)
                t.work = work.data_ptr() + 8 * i * wsz
                t.counter = counters.data_ptr() + 4 * 16 * i
                t.sums_host = sums.data_ptr() + 8 * 32 * i
            rc = self._lib.ttm_optimize_separable_batch(tasks, n, self._N, float(self._Nglobal), float(self.delta),
                                                        int(min(self.optimizer_threads, n)), self._stream(), 0)
            if rc != 0:
                self._sep_batch_scratch = None          # (a loop that was cut short may have left a ticket counter behind)
            _capi.check(rc)
            for i, k in enumerate(batch):
                armed[k] = int(tasks[i].armed)
            for i, k in enumerate(batch):
                out = _Result()
                r = tasks[i].result
                out.x, out.fun, out.nit, out.nfev, out.status = keep[i][2], float(r[0]), int(r[2]), int(r[3]), int(r[4])
                out.solve_nonmon = keep[i][6]
                results[k] = out
        return results

    def _penalty_vector(self, k, m):
        """(kind, lambda per coefficient) of the penalty of TM:3382-3431 for the native loops: kind 0 none, 1 l1, 2 l2;
        None when the setting is one the Python objective has to answer (it raises the reference's errors)."""
        if self.regularization is None:
            return 0, None
        if type(self.regularization) != str or self.regularization.lower() not in ('l1', 'l2'):
            return None
        kind = 1 if self.regularization.lower() == 'l1' else 2
        if np.isscalar(self.regularization_lambda):
            return kind, np.full(m, float(self.regularization_lambda))
        if type(self.regularization_lambda) == list:
            lam = np.ascontiguousarray(self.regularization_lambda[k], dtype=float)
            return (kind, lam) if lam.shape == (m,) else None
        return None

    def _optimize_integrated_batch(self, K):
        """TM:2746-2845 for integrated-rectifier maps: the BFGS loops of the components of K side by side
        (ttm_optimize_integrated_batch); {k: result} or None when the batched native path does not apply."""
        if not self.native_optimizer or len(K) < 2 or self.optimizer_threads < 2 or self._dist() is not None:
            return None
        torch = _torch()
        n = len(K)
        sizes = [len(self.coeffs_nonmon[k]) + len(self.coeffs_mon[k]) for k in K]
        pens = [self._penalty_vector(k, m) for k, m in zip(K, sizes)]
        if any(m < 1 or m > HOSTCOEF_MAX for m in sizes) or any(p is None for p in pens):
            return None
        wsz = int(self._lib.ttm_reduce_work_size(1 + max(sizes)))
        work = self._empty(n * wsz)
        counters = self._zeros(n * 16, dtype=torch.int32)
        sums = torch.zeros(n * 256, dtype=torch.float64, pin_memory=self._dev.type == 'cuda')
        tasks = (_capi.ttm_int_task * n)()
        xs = []
        for i, (k, m, (kind, lam)) in enumerate(zip(K, sizes, pens)):
            x = np.concatenate((np.asarray(self.coeffs_nonmon[k], dtype=float), np.asarray(self.coeffs_mon[k], dtype=float)))
            xs.append((x, lam))
            t = tasks[i]
            t.k, t.m, t.regularization = int(k), m, kind
            t.lam = lam.ctypes.data if lam is not None else None
            t.x = x.ctypes.data
            t.work = work.data_ptr() + 8 * i * wsz
            t.counter = counters.data_ptr() + 4 * 16 * i
            t.sums_host = sums.data_ptr() + 8 * 256 * i
        _capi.check(self._lib.ttm_optimize_integrated_batch(self._pp, tasks, n, self._ptr(self._Xs), self._Xs.shape[1], self._N,
                                                            float(self._Nglobal), int(min(self.optimizer_threads, n)),
                                                            self._stream(), 0))
        self._obj_cache = None
        results = {}
        for i, k in enumerate(K):
            class _Result:
                pass
            out = _Result()
            r = tasks[i].result
            out.x, out.fun, out.nit, out.nfev, out.status = xs[i][0], float(r[0]), int(r[2]), int(r[3]), int(r[4])
            results[k] = out
        return results

    def _optimize_integrated_native(self, k, x0, div):
        """TM:3252-3257 for one component without leaving the library (ttm_optimize_integrated: SciPy's BFGS restated
        over ttm_objective_host); None when the native loop does not apply (more than 128 coefficients, a penalty it
        does not know, ranks that share samples without an RCCL / test-double communicator)."""
        m = len(x0)
        if not self.native_optimizer or m > HOSTCOEF_MAX:
            return None
        pen = self._penalty_vector(k, m)
        if pen is None:
            return None
        reg, lam = pen
        handle = None
        if self._dist() is not None:
            handle = comm.get(self._lib, force=self._dev.type != 'cuda')
            if handle is None:
                return None
        torch = _torch()
        if getattr(self, '_obj_out', None) is None:
            self._obj_out = torch.zeros(256, dtype=torch.float64, pin_memory=self._dev.type == 'cuda')
            self._obj_cnt = self._zeros(16, dtype=torch.int32)
        x = np.array(x0, dtype=float, copy=True)
        work = self._workspace(self._lib.ttm_reduce_work_size(1 + m))
        sums_dev = self._empty(1 + m) if handle is not None else None
        res = np.zeros(5)
        p = lambda a: ctypes.c_void_p(a.ctypes.data) if a is not None else None                   # noqa: E731
        _capi.check(self._lib.ttm_optimize_integrated(
            self._pp, int(k), m, self._ptr(self._Xs), self._Xs.shape[1], self._N, float(self._Nglobal), reg, p(lam), p(x),
            self._ptr(work), ctypes.c_void_p(self._obj_cnt.data_ptr()), self._ptr(sums_dev),
            ctypes.c_void_p(self._obj_out.data_ptr()), handle, self._stream(), 0, p(res)))
        self._obj_cache = None

        class _Result:
            pass
        out = _Result()
        out.x, out.fun, out.nit, out.nfev, out.status = x, float(res[0]), int(res[2]), int(res[3]), int(res[4])
        out.success = out.status == 0
        return out

    def optimize(self, K=None):
        """TM:2714-2901: per-component minimisation (BFGS / L-BFGS-B as TM:3252-3257 / TM:3108-3114) on the device
        reductions, by the library's own optimiser loops (L-BFGS-B for separable, BFGS for integrated components)."""
        from scipy.optimize import minimize
        if K is None:
            K = np.arange(self.D)
        K = [int(k) for k in K]
        import torch.distributed as tdist
        part = self.shard_components and tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1
        if part and self.shard_samples:
            raise ValueError('shard_components needs the full ensemble on every rank (shard_samples=False)')
        # components are independent problems (TM:2746-2786): with shard_components every rank optimises a
        # strided subset (most expensive, i.e. last, components first as TM:2814-2822) on its replica of X
        owner = self._partition_components(K, tdist.get_world_size()) if part else None
        K_local = [k for k in reversed(K) if owner[k] == tdist.get_rank()] if part else K
        J_local = 0.0
        n_eval = 0
        if self.monotonicity == "separable monotonicity":
            batched = self._optimize_separable_batch(K_local)
        elif self.monotonicity == "integrated rectifier":
            batched = self._optimize_integrated_batch(K_local)
        else:
            batched = None
        for k in K_local:
            if batched is not None and self.monotonicity == "separable monotonicity":
                opt = batched[k]
                self.coeffs_mon[k] = np.array(opt.x, copy=True)
                self.coeffs_nonmon[k] = opt.solve_nonmon(opt.x)
            elif batched is not None:
                opt = batched[k]
                div = len(self.coeffs_nonmon[k])
                self.coeffs_nonmon[k] = copy.deepcopy(opt.x[:div])
                self.coeffs_mon[k] = copy.deepcopy(opt.x[div:])
            elif self.monotonicity == "integrated rectifier":
                div = len(self.coeffs_nonmon[k])
                x0 = np.concatenate((np.asarray(self.coeffs_nonmon[k], dtype=float),
                                     np.asarray(self.coeffs_mon[k], dtype=float)))
                opt = self._optimize_integrated_native(k, x0, div)
                if opt is None:
                    opt = minimize(method='BFGS', fun=self.objective_function, jac=self.objective_function_jacobian,
                                   x0=x0, args=(k, div))
                self.coeffs_nonmon[k] = copy.deepcopy(opt.x[:div])
                self.coeffs_mon[k] = copy.deepcopy(opt.x[div:])
            elif self.monotonicity == "separable monotonicity":
                A, solve_nonmon = self.separable_setup(k)
                bounds = [[self.optimization_constraints_lb[k][i], self.optimization_constraints_ub[k][i]]
                          for i in range(len(self.optimization_constraints_lb[k]))]
                self._sep_cache_begin(k)
                try:
                    # (scipy.optimize.minimize(method='L-BFGS-B') as TM:3108-3114, minus its per-evaluation wrappers)
                    opt = self._optimize_separable_native(A, k, np.asarray(self.coeffs_mon[k], dtype=float), bounds)
                    if opt is None:
                        fast = self._sep_objective_fast(A, k)
                        opt = lbfgsb.minimize_lbfgsb(fast if fast is not None else self.separable_objective,
                                                     np.asarray(self.coeffs_mon[k], dtype=float), bounds, (A, k))
                finally:
                    self._sep_cache_end()
                self.coeffs_mon[k] = copy.deepcopy(opt.x)
                self.coeffs_nonmon[k] = solve_nonmon(opt.x)
            J_local += float(opt.fun)
            n_eval += int(getattr(opt, 'nfev', 0) or 0)
            if self.verbose:
                string = '\r' + 'Progress: |' + (k + 1) * '█' + (len(K) - k - 1) * ' ' + '|'
                print(string, end='\r')
        if part:
            # exchange: coefficients of every component from its owner (a few KB) and ONE scalar all-reduce
            # of the summed objective - the only collective of the partitioned optimisation
            torch = _torch()
            n_tot = int(self._cm.coef_off[-1])
            buf = torch.zeros(n_tot, dtype=torch.float64, device=self._dev)
            for k in K_local:
                o = int(self._cm.coef_off[k])
                ck = np.concatenate((self.coeffs_nonmon[k], self.coeffs_mon[k]))
                buf[o:o + len(ck)] = torch.from_numpy(ck).to(self._dev)
            self._allreduce_world(buf)                 # disjoint supports: sum == gather
            allc = buf.cpu().numpy()
            for k in K:
                o, nn, nm = int(self._cm.coef_off[k]), int(self._cm.n_nm[k]), int(self._cm.n_mon[k])
                self.coeffs_nonmon[k] = allc[o:o + nn].copy()
                self.coeffs_mon[k] = allc[o + nn:o + nn + nm].copy()
            jt = torch.tensor([J_local], dtype=torch.float64, device=self._dev)
            self._allreduce_world(jt)                  # the one scalar all-reduce of the partitioned optimisation
            self.objective_total = float(jt.item())
        else:
            self.objective_total = J_local
        self.last_optimize_evaluations = n_eval          # objective evaluations of this rank's components (benchmarks)
        return

    def _partition_components(self, K, world):
        """Owner rank of every component of K when the components are partitioned (TM:2789-2845: the reference's pool takes
        them in reverse order, most expensive first).  Cost of a component = coefficients x columns it reads (the work of
        one evaluation of its reduced problem); longest-processing-time assignment: most expensive first, each to the
        rank with the least work so far - the same on every rank."""
        cm = self._cm

        def cost(k):
            cols = {k + self.skip_dimensions}
            for entry in list(self.monotone[k]) + list(self.nonmonotone[k]):
                if isinstance(entry, str):
                    cols.add(int(entry.split(' ')[1]))
                else:
                    cols.update(int(e) for e in entry if not isinstance(e, str))
            return (int(cm.n_nm[k]) + int(cm.n_mon[k])) * len(cols)
        load = [0] * world
        owner = {}
        for k in sorted(K, key=lambda kk: (-cost(kk), -kk)):
            r = min(range(world), key=lambda i: (load[i], i))
            owner[k] = r
            load[r] += cost(k)
        return owner
