"""Build libttm.so (HIP kernels + C ABI) in-tree with hipcc for gfx950."""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, 'csrc')
LIB = os.path.join(PKG, 'libttm.so')
SOURCES = ['ttm_kernels.hip']
HEADERS = ['ttm_eval.h', 'ttm_math.h', 'ttm_vec.h', 'ttm_erf_table.h', os.path.join('..', '..', 'include', 'ttm.h')]
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-ffp-contract=off', '-fPIC', '-shared',
         '-DNDEBUG']


def hipcc_path():
    for cand in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found: the MI355X engine needs the ROCm toolchain to build libttm.so')


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -shared ... -> triangular_transport_toolbox_amd/libttm.so"""
    if not force and not is_stale():
        return LIB
    tmp = '%s.tmp.%d' % (LIB, os.getpid())        # atomic replace: several ranks may build at the same time
    cmd = [hipcc_path()] + FLAGS + ['-o', tmp] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(' '.join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError('hipcc failed:\n' + res.stdout + res.stderr)
    os.replace(tmp, LIB)
    return LIB


if __name__ == '__main__':
    print(build_lib(force=True, verbose=True))
