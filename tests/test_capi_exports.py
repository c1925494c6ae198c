"""CPU: the C-ABI shared library loads and exports every symbol that include/morec_hip.h declares
(no compute calls -- there is no GPU in the build container)."""
import os
import re

import pytest

from idvs.morec_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "morec_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(morec_[a-z0-9_]+)\s*\(", text)))


def test_library_present_and_loads():
    assert os.path.exists(_lib.LIB_PATH), "build with `python -c 'import __graft_entry__ as g; g.build()'`"
    h = _lib.lib()
    assert h.morec_version() >= 100
    assert b"MOREC_E_ALIGN" in h.morec_strerror(-2)


def test_every_declared_symbol_is_exported_and_bound():
    h = _lib.lib()
    names = _declared()
    assert len(names) >= 24
    for n in names:
        assert hasattr(h, n), f"{n} declared in morec_hip.h but not exported by libmorec_hip.so"
        assert n in _lib.EXPORTS, f"{n} has no ctypes signature in idvs/morec_amd/_lib.py"
    assert set(_lib.EXPORTS) <= set(names)


def test_argument_validation_without_gpu():
    """Entry points reject bad arguments before touching the device."""
    import ctypes as C
    h = _lib.lib()
    d = _lib.GemmDesc(0, 8, 8, 8, 8, 8, 0, 0, 0, 0, 0, 1, 1.0)
    assert h.morec_gemm_nt(C.byref(d), None, None, None, None, None, None, None) == -1
    assert h.morec_transpose(None, None, 4, 4, 4, 4, 0, 0, None) == -1
    assert h.morec_inbatch_ce_workspace_bytes(C.byref(_lib.CeDesc(128, 20, 512, 2688, 0, 1, 0))) > 0


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libmorec_hip.so")
    with pytest.raises(_lib.MorecError):
        _lib.lib()
