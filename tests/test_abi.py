"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/quip_amd.h declares, and
rejects bad arguments before touching a device (no compute calls here -- those are the -m gpu tests)."""
import ctypes
import os
import re

import pytest

from quip_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "quip_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(quipamd_[a-z0-9_]+)\s*\(", hdr)))


def test_library_is_built_and_loads():
    lib = _lib.load()
    assert lib.quipamd_version() == 100


def test_every_declared_symbol_is_exported_and_bound():
    names = _declared()
    assert len(names) >= 11
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/quip_amd.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in quip_amd/_lib.py"
    assert sorted(_lib.SIGNATURES) == names


def test_bad_arguments_fail_loudly_without_a_device():
    null = ctypes.c_void_p(0)
    with pytest.raises(_lib.QuipAmdError, match="null pointer"):
        _lib.call("quipamd_pack", null, 2, 0, null, 16, 256, null)
    one = ctypes.c_void_p(16)   # non-null, never dereferenced: argument checks come first
    with pytest.raises(_lib.QuipAmdError, match="bits must be 2 or 4"):
        _lib.call("quipamd_pack", one, 5, 0, one, 16, 256, null)
    with pytest.raises(_lib.QuipAmdError, match="3-bit layout needs"):       # the reference's 32-codes-in-3-words rule
        _lib.call("quipamd_pack", one, 3, 0, one, 16, 40, null)
    with pytest.raises(_lib.QuipAmdError, match="workspace too small"):
        _lib.call("quipamd_vecquant4matmul", one, one, one, one, one, 16, 128, one, 8, null)
    assert _lib.load().quipamd_vecquant_workspace_bytes(4, 16, 128) >= 16 * 128 // 2 + 2 * 128 * 2 + 16 * 4
    with pytest.raises(_lib.QuipAmdError, match="stream layout needs"):
        _lib.call("quipamd_pack", one, 2, 1, one, 10, 256, null)
    with pytest.raises(_lib.QuipAmdError, match="STREAM layout"):
        _lib.call("quipamd_dequant_gemm", one, 2, one, 2, 0, 1, one, null, null, one, 2, 0, 1, 16, 256, null)
    with pytest.raises(_lib.QuipAmdError, match="d % 16"):
        _lib.call("quipamd_ldlq_round", one, one, null, 2, one, one, 4, 24, null)


def test_ops_refuse_cpu_tensors():
    import torch
    from quip_amd import ops
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.pack(torch.zeros(16, 256, dtype=torch.uint8), 2)
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.ldlq_round(torch.zeros(16, 16), torch.zeros(16, 16), 2)


def test_missing_library_is_an_error_not_a_fallback(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libquip_amd.so")
    with pytest.raises(_lib.QuipAmdError, match="no CPU fallback"):
        _lib.load()
