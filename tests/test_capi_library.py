"""CPU: the product's C-ABI library loads and exports every symbol include/batotp_hip.h declares."""
import os
import re

import helpers
from batotp_amd import capi


def _declared_symbols():
    text = open(os.path.join(helpers.ROOT, "include", "batotp_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(batotp_hip_\w+)\s*\(", text)))


def test_header_symbols_are_exported(hip_lib):
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(hip_lib.lib, name), f"{name} declared in include/batotp_hip.h but not exported"
    assert set(declared) == set(hip_lib.symbols)


def test_oracle_shim_implements_the_same_abi(oracle_lib):
    for name in _declared_symbols():
        assert hasattr(oracle_lib.lib, name)


def test_struct_layouts():
    import ctypes as C
    assert C.sizeof(capi.Problem) == 4 * 4 + 4 * 64 + 6 * 8 + 72
    assert C.sizeof(capi.PathResult) == 64


def test_no_gpu_means_loud_failure(hip_lib):
    """without a GPU the product refuses to create a context (no CPU fallback)"""
    import pytest
    if hip_lib.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(capi.BatotpError):
        capi.Context(hip_lib, 0)
