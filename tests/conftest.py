import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def repo_root():
    return ROOT


@pytest.fixture
def ttm_opt():
    """Pin launch-planning options (include/ttm.h: ttm_set_option) for one test, on the device library and on the host
    test double alike; everything goes back to the defaults afterwards."""
    import ctypes
    libs = []
    try:
        from triangular_transport_toolbox_amd import build as _b
        if os.path.exists(_b.LIB) and not _b.is_stale():
            import torch  # noqa: F401  (its HIP runtime first, as _capi.load() does)
            dev = ctypes.CDLL(_b.LIB)               # (same dlopen handle, hence the same option table, as _capi's)
            dev.ttm_set_option.argtypes = [ctypes.c_char_p, ctypes.c_int32]
            libs.append(dev)
    except Exception:                                   # noqa: BLE001  (no device library here: the double alone)
        pass
    from tests.hostemu import emu

    def targets():
        # the host test double only when a test has put it in place (emu.install()): a GPU run never loads it
        return libs + ([emu._lib] if emu._lib is not None else [])

    def opt(name, value):
        for lib in targets():
            rc = lib.ttm_set_option(name.encode(), int(value))
            assert rc == 0, name
    yield opt
    for lib in targets():
        lib.ttm_reset_options()
