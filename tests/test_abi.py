"""CPU test: libplanar_hip.so loads and exports every symbol include/planar_abi.h declares (no compute)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "planar_abi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(planar_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from planarslam_amd import _lib
    syms = declared_symbols()
    assert len(syms) >= 20
    L = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing


def test_python_binding_covers_the_header():
    from planarslam_amd import _lib
    assert sorted(_lib.exported_symbols()) == declared_symbols()


def test_version_and_error_string_without_gpu():
    from planarslam_amd import _lib
    L = _lib.lib()
    assert L.planar_abi_version() >= 100
    assert isinstance(L.planar_last_error(), bytes)
