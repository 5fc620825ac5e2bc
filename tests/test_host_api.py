"""CPU: the drop-in boundary.  The reference's own driver (test/main.cpp) must compile UNCHANGED against
this repository's headers and, linked with the host library (+ the oracle shim for the device calls),
reproduce the reference binary's files.  Needs /root/reference, i.e. runs in the build container only."""
import filecmp
import os
import shutil
import subprocess

import pytest

import helpers

REF_MAIN = "/root/reference/test/main.cpp"
HOST = os.path.join(helpers.ROOT, "batotp_amd", "host")


@pytest.mark.skipif(not os.path.exists(REF_MAIN), reason="reference sources are not on this machine")
def test_reference_driver_compiles_unchanged_and_runs(tmp_path, oracle_lib):
    srcs = [os.path.join(HOST, f) for f in ("ba.cpp", "ba_input.cpp", "ba_output.cpp", "ba_io.cpp", "ba_device.cpp",
                                            "spline.cpp", "robot.cpp", "util.cpp")]
    exe = tmp_path / "batest_refmain"
    cmd = ["g++", "-std=c++11", "-O2", "-ffp-contract=off", "-DNDEBUG", f"-I{HOST}", f"-I{helpers.ROOT}/include",
           REF_MAIN, *srcs, f"-L{helpers.BUILD}", "-lbatotp_oracle_abi", f"-Wl,-rpath,{helpers.BUILD}", "-fopenmp", "-pthread", "-lm", "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    for name in ("RR", "GEN7DOF"):
        work = tmp_path / name
        work.mkdir()
        src = os.path.join(helpers.GOLD, name)
        for f in os.listdir(src):
            if not f.startswith("ref_") and f not in ("knots.npz", "expected.json"):
                shutil.copy(os.path.join(src, f), work / f)
        run = subprocess.run([str(exe), "config.dat"], cwd=work, capture_output=True, text=True)
        assert run.returncode == 0, run.stdout[-2000:]
        assert filecmp.cmp(work / "s-sdot.dat", os.path.join(src, "ref_s-sdot.dat"), shallow=False)
        assert filecmp.cmp(work / "traj_out.dat", os.path.join(src, "ref_traj_out.dat"), shallow=False)


def test_product_batest_fails_loudly_without_gpu(tmp_path, hip_lib):
    """the product driver is linked against the HIP library: without a GPU it must refuse, not fall back"""
    exe = os.path.join(HOST, "_build", "batest")
    if not os.path.exists(exe):
        pytest.skip("batest not built")
    if hip_lib.device_count() > 0:
        pytest.skip("a GPU is present")
    src = os.path.join(helpers.GOLD, "GEN7DOF")
    for f in ("config.dat", "GEN7DOFpathBasic.csv"):
        shutil.copy(os.path.join(src, f), tmp_path / f)
    r = subprocess.run([exe, "config.dat"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode != 0
    assert "no usable HIP device" in r.stdout and "no CPU fallback" in r.stdout
    assert not os.path.exists(tmp_path / "traj_out.dat")
