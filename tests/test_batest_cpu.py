"""CPU: the host-side BA library end to end (file IO, resampling, output stage) against the files
the reference binary wrote for the same inputs.  The device calls are served by the oracle shim
(oracle/_build/batest_oracle), so this also covers the BA <-> C-ABI marshalling."""
import filecmp
import os
import shutil
import subprocess

import pytest

import helpers


@pytest.mark.parametrize("name", helpers.FULL_CASES)
def test_batest_files_equal_reference(tmp_path, oracle_lib, name):
    src = os.path.join(helpers.GOLD, name)
    for f in os.listdir(src):
        if not f.startswith("ref_") and f not in ("knots.npz", "expected.json"):
            shutil.copy(os.path.join(src, f), tmp_path / f)
    r = subprocess.run([os.path.join(helpers.BUILD, "batest_oracle"), "config.dat"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    assert filecmp.cmp(tmp_path / "s-sdot.dat", os.path.join(src, "ref_s-sdot.dat"), shallow=False)
    assert filecmp.cmp(tmp_path / "traj_out.dat", os.path.join(src, "ref_traj_out.dat"), shallow=False)
    assert os.path.getsize(tmp_path / "compTimes.dat") == 12


def _stage(src, dst):
    for f in os.listdir(src):
        if not f.startswith("ref_") and not f.endswith(".npz") and not f.endswith(".json"):
            shutil.copy(os.path.join(src, f), dst / f)


@pytest.mark.parametrize("mode", ["device-resample", "host-resample", "host-output"])
@pytest.mark.parametrize("name", ["synth_cspr_s3", "synth_gen7dof_s1_vel", "synth_ur_s2", "GEN7DOF", "UR5", "RR"])
def test_batch_driver_files_equal_reference(tmp_path, oracle_lib, name, mode):
    """BA::optimizeBatch (the many-path extension) writes, for every copy of the path, the files the
    reference binary wrote for the single path -- with the resampling done behind the C-ABI
    (batotp_hip_resample, configurations it covers) and by the host resampler, the output stage behind the C-ABI
    (batotp_hip_output, configurations it covers) and by the host code"""
    src = os.path.join(helpers.GOLD, name)
    _stage(src, tmp_path)
    cmd = [os.path.join(helpers.BUILD, "batest_batch_oracle"), "config.dat", "3"] + (["--" + mode] if mode.startswith("host-") else [])
    r = subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    for d in ("out_first", "out_last"):
        assert filecmp.cmp(tmp_path / d / "s-sdot.dat", os.path.join(src, "ref_s-sdot.dat"), shallow=False)
        assert filecmp.cmp(tmp_path / d / "traj_out.dat", os.path.join(src, "ref_traj_out.dat"), shallow=False)
