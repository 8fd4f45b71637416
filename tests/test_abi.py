"""The C-ABI library loads and exports every symbol include/wrt.h declares
(no compute calls: runs without a GPU)."""
import ctypes
import os
import re

from tests.helpers import REPO
from weekend_raytracer import _native


def declared_functions():
    text = open(os.path.join(REPO, "include", "wrt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(wrt_[a-z_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    names = declared_functions()
    assert len(names) >= 20
    lib = ctypes.CDLL(_native.LIB_PATH)
    for name in names:
        assert hasattr(lib, name), f"{name} is declared in wrt.h but not exported by libwrt.so"
    assert sorted(_native.SIGNATURES) == names
    _native.load_library()


def test_struct_layouts_match_header():
    # wrt_material: 2 int32 + 6 float + 6 double + 2 double; wrt_stats: 64 + 7 + 8 + 5 uint64
    assert ctypes.sizeof(_native.WrtMaterial) == 8 + 24 + 48 + 16
    assert ctypes.sizeof(_native.WrtStats) == 8 * (64 + 7 + 8 + 5)


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        return
    lib = _native.load_library()
    handle = ctypes.c_void_p()
    rc = lib.wrt_create(0, ctypes.byref(handle))
    assert rc != 0 and handle.value is None
    assert lib.wrt_last_error(None)
    try:
        _native.context()
    except _native.NativeError as exc:
        assert "no CPU fallback" in str(exc)
    else:
        raise AssertionError("context() must not succeed without a GPU")
