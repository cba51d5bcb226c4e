"""Build libttm.so (HIP kernels + C ABI) in-tree with hipcc for gfx950.

Staleness is decided from what the compiler itself reports: every build writes the dependency list of the
translation unit (`hipcc -MD`, every header it really included) and a stamp of the command line next to the
library; the library is rebuilt when any of those files is newer, when the flags differ, or when the stamp is
missing.  A tuning build with extra flags (TTM_BUILD_FLAGS, TTM_BUILD_LIB) therefore never passes for the product
library: a different flag set is a different stamp, and a different output path if TTM_BUILD_LIB says so."""
import hashlib
import os
import shlex
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, 'csrc')
LIB = os.environ.get('TTM_BUILD_LIB') or os.path.join(PKG, 'libttm.so')
SOURCES = ['ttm_kernels.hip', 'ttm_band.hip', 'ttm_int.hip', 'ttm_optim.cpp', 'ttm_comm.cpp']
# what the translation units include (the depfile of the last build supersedes this list)
HEADERS = ['ttm_eval.h', 'ttm_math.h', 'ttm_vec.h', 'ttm_erf_table.h', 'ttm_uform.h', 'ttm_cheb_table.h', 'ttm_lbfgsb.h', 'ttm_bfgs.h', 'ttm_rng.h', 'ttm_band.h', 'ttm_band_etab.h', 'ttm_dev.h', 'ttm_dense.h', 'ttm_dense_table.h', 'ttm_int.h', 'ttm_xprog.h', 'ttm_handover.h',
           os.path.join('..', '..', 'include', 'ttm.h')]
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-ffp-contract=off', '-fPIC', '-shared',
         '-DNDEBUG'] + shlex.split(os.environ.get('TTM_BUILD_FLAGS', ''))
LINK = ['-ldl', '-pthread']                   # (RCCL is bound at run time, csrc/ttm_comm.cpp)
# tuning builds: extra flags for ONE translation unit (only that object is recompiled), e.g. TTM_BAND_FLAGS=-DBAND_X=1
EXTRA = {'ttm_band.hip': shlex.split(os.environ.get('TTM_BAND_FLAGS', '')), 'ttm_int.hip': shlex.split(os.environ.get('TTM_INT_FLAGS', ''))}


def hipcc_path():
    for cand in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found: the MI355X engine needs the ROCm toolchain to build libttm.so')


def _sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _stamp():
    extra = [k + ':' + ' '.join(v) for k, v in sorted(EXTRA.items()) if v]
    return hashlib.sha256(' '.join(FLAGS + LINK + [os.path.basename(s) for s in _sources()] + extra).encode()).hexdigest()


def _deps():
    """Files the library was built from: the compiler's depfiles of the last build when present, else the static list."""
    deps = set(_sources()) | {os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS}
    dpath = LIB + '.d'
    if os.path.exists(dpath):
        txt = open(dpath).read().replace('\\\n', ' ')
        for tok in txt.split():
            if tok.endswith(':'):
                continue
            if not tok.startswith(('/opt/', '/usr/')):
                deps.add(os.path.normpath(tok))
    return [d for d in deps if os.path.exists(d)]


def is_stale():
    if not os.path.exists(LIB):
        return True
    spath = LIB + '.stamp'
    if not os.path.exists(spath) or open(spath).read().strip() != _stamp():
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in _deps())


def can_build():
    try:
        hipcc_path()
        return True
    except RuntimeError:
        return False


def _obj_deps(dpath):
    deps = []
    if os.path.exists(dpath):
        txt = open(dpath).read().replace('\\\n', ' ')
        deps = [tok for tok in txt.split() if not tok.endswith(':') and not tok.startswith(('/opt/', '/usr/'))]
    return deps


def build_lib(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -shared ... -> triangular_transport_toolbox_amd/libttm.so
    Objects are kept per source file under _obj/ (keyed by the flag stamp) and recompiled only when the source or a
    header the compiler reported for it is newer: a change to one translation unit does not rebuild the others."""
    if not force and not is_stale():
        return LIB
    tmp = '%s.tmp.%d' % (LIB, os.getpid())        # atomic replace: several ranks may build at the same time
    objdir = os.path.join(PKG, '_obj')
    os.makedirs(objdir, exist_ok=True)
    base_stamp = hashlib.sha256(' '.join(FLAGS).encode()).hexdigest()[:12]
    objs, deptxt = [], []
    try:
        for src in _sources():
            extra = EXTRA.get(os.path.basename(src), [])
            stamp = base_stamp if not extra else hashlib.sha256(' '.join(FLAGS + extra).encode()).hexdigest()[:12]
            obj = os.path.join(objdir, '%s.%s.o' % (os.path.basename(src), stamp))
            dpath = obj + '.d'
            fresh = (not force) and os.path.exists(obj) and os.path.exists(dpath) and \
                all(os.path.exists(d) and os.path.getmtime(d) <= os.path.getmtime(obj) for d in [src] + _obj_deps(dpath))
            if not fresh:
                otmp = '%s.tmp.%d' % (obj, os.getpid())
                cflags = [f for f in FLAGS if f != '-shared'] + extra
                cmd = [hipcc_path()] + cflags + ['-x', 'hip', '-c', src, '-MD', '-MF', otmp + '.d', '-o', otmp]
                if verbose:
                    print(' '.join(cmd))
                res = subprocess.run(cmd, capture_output=True, text=True)
                if res.returncode != 0:
                    for path in (otmp, otmp + '.d'):
                        if os.path.exists(path):
                            os.remove(path)
                    raise RuntimeError('hipcc failed:\n' + res.stdout + res.stderr)
                os.replace(otmp, obj)
                os.replace(otmp + '.d', dpath)
            objs.append(obj)
            if os.path.exists(dpath):
                deptxt.append(open(dpath).read())
        cmd = [hipcc_path(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', tmp] + objs + LINK
        if verbose:
            print(' '.join(cmd))
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError('hipcc (link) failed:\n' + res.stdout + res.stderr)
        os.replace(tmp, LIB)
        with open(LIB + '.d', 'w') as f:
            f.write('\n'.join(deptxt))
        with open(LIB + '.stamp', 'w') as f:
            f.write(_stamp() + '\n')
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return LIB


if __name__ == '__main__':
    print(build_lib(force=True, verbose=True))
