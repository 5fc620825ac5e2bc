"""pytest configuration: the `gpu` marker and the libraries under test.

CPU tests (-m "not gpu") exercise the oracle against the golden vectors produced by the reference
itself, the host-side BA library, and that the HIP C-ABI library loads and exports its symbols.
GPU tests (-m gpu) are the parity tests proper: HIP kernels vs the oracle, through the C-ABI.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun)")
    if os.environ.get("BATOTP_TEST_POISON") == "1":
        # debug run: every context of the suite poisons its workspaces and batch arrays before use (batotp_hip_set_poison): a kernel
        # that reads memory nobody wrote then fails a parity test instead of depending on what the memory held
        from batotp_amd import capi
        capi.DEFAULT_POISON = True


def _ensure_oracle_built():
    build = os.path.join(ROOT, "oracle", "_build")
    need = ["libbatotp_oracle_abi.so", "libbatotp_oracle.so", "batest_oracle", "dump_knots"]
    if all(os.path.exists(os.path.join(build, n)) for n in need):
        return
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)


@pytest.fixture(scope="session")
def oracle_lib():
    """TEST INFRASTRUCTURE: the CPU oracle behind the C-ABI."""
    _ensure_oracle_built()
    import helpers
    return helpers.load_oracle()


@pytest.fixture(scope="session")
def oracle_ctx(oracle_lib):
    from batotp_amd import capi
    return capi.Context(oracle_lib, 0)


@pytest.fixture(scope="session")
def hip_lib():
    from batotp_amd import capi
    return capi.load_hip()  # raises if the product library was not built


@pytest.fixture(scope="session")
def hip_ctx(hip_lib):
    from batotp_amd import capi
    return capi.Context(hip_lib, 0)  # raises without a usable GPU: no fallback
