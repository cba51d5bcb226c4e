"""The C-ABI shared library loads and exports every symbol include/ttm.h declares (no compute calls)."""
import ctypes
import os
import re

from triangular_transport_toolbox_amd import _capi, build


def declared_symbols():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, 'include', 'ttm.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(ttm_[a-z_0-9]+)\s*\(', text)))


def test_header_and_binding_agree():
    assert declared_symbols() == _capi.EXPORTED_SYMBOLS


def test_library_builds_and_exports_all_symbols():
    path = build.build_lib()
    lib = ctypes.CDLL(path)
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert _capi.load().ttm_version() == 100


def test_struct_layout_matches_header():
    # 17 + 6 pointers, 8 + 2 + 2 int32, 1 double, 3 + 1 int64 on LP64 - and what the library itself was compiled with
    assert ctypes.sizeof(_capi.ttm_program) == 23 * 8 + 12 * 4 + 8 + 4 * 8
    assert _capi.load().ttm_program_sizeof() == ctypes.sizeof(_capi.ttm_program)
    from tests.hostemu import emu
    assert emu.lib().ttm_program_sizeof() == ctypes.sizeof(_capi.ttm_program)


def test_no_device_fails_loudly():
    """Without a HIP device the product refuses to construct a map (no CPU fallback)."""
    import numpy as np
    import pytest
    if _capi.device_count() > 0:
        pytest.skip('a GPU is visible')
    from triangular_transport_toolbox_amd.transport_map import transport_map
    with pytest.raises(RuntimeError):
        transport_map(X=np.zeros((4, 1)), monotone=[[[0]]], nonmonotone=[[[]]], verbose=False)


def test_host_double_exports_same_abi():
    from tests.hostemu import emu
    lib = emu.lib()
    for name in declared_symbols():
        assert hasattr(lib, name), name
