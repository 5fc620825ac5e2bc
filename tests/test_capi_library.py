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


def test_flag_and_status_constants_match_the_header():
    """the Python binding's BATOTP_F_* / BATOTP_ST_* values are the header's"""
    text = open(os.path.join(helpers.ROOT, "include", "batotp_hip.h")).read()
    defs = {m.group(1): 1 << int(m.group(2)) for m in re.finditer(r"#define\s+BATOTP_(F_\w+|ST_\w+)\s+\(1u<<(\d+)\)", text)}
    assert len(defs) >= 15
    checked = 0
    for name, value in defs.items():
        if hasattr(capi, name):
            assert getattr(capi, name) == value, name
            checked += 1
    for name in ("F_COMPACT_SPLINES", "F_CURVES_IN_PLACE", "F_MVC_IN_CURVES", "F_NO_SAMPLES", "ST_CAPACITY", "ST_MAX_INTEG_TIME"):
        assert name in defs and hasattr(capi, name), name
    assert checked >= 12
