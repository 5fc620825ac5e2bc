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


@pytest.mark.parametrize("name", ["synth_cspr_s3", "synth_gen7dof_s1_vel", "synth_ur_s2", "GEN7DOF", "UR5", "UR5_pos3", "RR", "RR_acc", "KUKA-LWR-IV", "KUKA_cartacc",
                                  "CSPR3DOF", "CSPR3DOF_svd", "synth_gen7dof_s10_decim", "synth_cspr_s9_dup"])
def test_batch_driver_files_equal_reference(tmp_path, oracle_lib, name):
    """BA::optimizeBatch (the many-path extension) writes, for every copy of the path, the files the reference binary wrote for
    the single path -- resampling (batotp_hip_resample) and output stage (batotp_hip_output) behind the C-ABI"""
    src = os.path.join(helpers.GOLD, name)
    _stage(src, tmp_path)
    cmd = [os.path.join(helpers.BUILD, "batest_batch_oracle"), "config.dat", "3"]
    r = subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    for d in ("out_first", "out_last"):
        assert filecmp.cmp(tmp_path / d / "s-sdot.dat", os.path.join(src, "ref_s-sdot.dat"), shallow=False)
        assert filecmp.cmp(tmp_path / d / "traj_out.dat", os.path.join(src, "ref_traj_out.dat"), shallow=False)


def _edit_config(path, subs):
    import re
    out = []
    for line in open(path):
        for key, val in subs.items():
            if re.search(r"//\s*" + re.escape(key) + r"\b", line):
                line = f"{val} // {key} (edited)\n"
        out.append(line)
    open(path, "w").write("".join(out))


def test_batch_driver_grows_the_curve_capacity_like_the_single_path_api(tmp_path, oracle_lib):
    """a path that needs more integration steps than the batch's first guess (8 per knot + 4096): BA::optimizeBatch runs the
    batch again with more room instead of failing the path -- same files as BA::optimize on the same input"""
    src = os.path.join(helpers.GOLD, "GEN7DOF")
    one, many = tmp_path / "one", tmp_path / "many"
    for d in (one, many):
        d.mkdir()
        _stage(src, d)
        _edit_config(d / "config.dat", {"integRes": "0.0004", "outRes": "0.004"})
    r1 = subprocess.run([os.path.join(helpers.BUILD, "batest_oracle"), "config.dat"], cwd=one, capture_output=True, text=True)
    assert r1.returncode == 0, r1.stdout[-2000:]
    import re
    steps = int(re.search(r"fwd\. integ\.:\s*(\d+) steps", r1.stdout).group(1))
    assert steps > 8 * 231 + 4096, steps          # beyond the first guess
    r = subprocess.run([os.path.join(helpers.BUILD, "batest_batch_oracle"), "config.dat", "2"], cwd=many, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "running the batch again" in r.stdout
    for d in ("out_first", "out_last"):
        assert filecmp.cmp(many / d / "s-sdot.dat", one / "s-sdot.dat", shallow=False)
        assert filecmp.cmp(many / d / "traj_out.dat", one / "traj_out.dat", shallow=False)


def test_batch_driver_grows_a_shared_curve_buffer_when_only_the_forward_curve_runs_out(tmp_path, oracle_lib):
    """is_sdotOut = 0: one curve buffer per path (BATOTP_F_CURVES_IN_PLACE).  The reverse curve fits the first guess, the
    forward curve does not: its sweep gives up 64 points before it would overwrite unread reverse points, i.e. with
    steps_fwd well below the capacity -- BA::optimizeBatch must still read that as 'out of room' and run the batch again
    (round-2 advisor finding; the checker library applies the same margin through bo_sweep_ex)"""
    import re
    src = os.path.join(helpers.GOLD, "GEN7DOF")
    one, many = tmp_path / "one", tmp_path / "many"
    for d, sdot_out in ((one, "1"), (many, "0")):
        d.mkdir()
        _stage(src, d)
        _edit_config(d / "config.dat", {"integRes": "0.00076", "outRes": "0.004", "is_sdotOut": sdot_out})
    r1 = subprocess.run([os.path.join(helpers.BUILD, "batest_oracle"), "config.dat"], cwd=one, capture_output=True, text=True)
    assert r1.returncode == 0, r1.stdout[-2000:]
    fwd = int(re.search(r"fwd\. integ\.:\s*(\d+) steps", r1.stdout).group(1))
    rev = int(re.search(r"rev\. integ\.:\s*(\d+) steps", r1.stdout).group(1))
    cap = 8 * 231 + 4096
    assert rev + 1 < cap <= fwd + 66, (rev, fwd, cap)          # the window the finding is about
    r = subprocess.run([os.path.join(helpers.BUILD, "batest_batch_oracle"), "config.dat", "2"], cwd=many, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "running the batch again" in r.stdout and "2 paths, 0 failed" in r.stdout
    for d in ("out_first", "out_last"):
        assert filecmp.cmp(many / d / "traj_out.dat", one / "traj_out.dat", shallow=False)
        assert not os.path.exists(many / d / "s-sdot.dat")


def test_svd_solver_reproduces_the_reference_binary(tmp_path, oracle_lib):
    """isSVD = 1 on the cable robot: solveLinSys goes through the two-sided Jacobi SVD of Eigen (reference util.cpp:421-438),
    restated in the host library (util.cpp), the checker (batotp_oracle_svd.c) and the kernels.  The reference binary's own
    isSVD = 1 outputs differ from its LU outputs in dozens of float32 values -- and are what the drop-in writes (the golden
    cases CSPR3DOF_svd / CSPR3DOF_par_svd are part of FULL_CASES above); here: the two solvers really are different code paths"""
    for name, other in (("CSPR3DOF_svd", "CSPR3DOF"), ("CSPR3DOF_par_svd", "CSPR3DOF_par")):
        a = open(os.path.join(helpers.GOLD, name, "ref_s-sdot.dat"), "rb").read()
        b = open(os.path.join(helpers.GOLD, other, "ref_s-sdot.dat"), "rb").read()
        assert a != b
    src = os.path.join(helpers.GOLD, "CSPR3DOF_par_svd")
    _stage(src, tmp_path)
    r = subprocess.run([os.path.join(helpers.BUILD, "batest_batch_oracle"), "config.dat", "2"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    for d in ("out_first", "out_last"):
        assert filecmp.cmp(tmp_path / d / "s-sdot.dat", os.path.join(src, "ref_s-sdot.dat"), shallow=False)
        assert filecmp.cmp(tmp_path / d / "traj_out.dat", os.path.join(src, "ref_traj_out.dat"), shallow=False)


@pytest.mark.parametrize("name,mode", [("synth_gen7dof_s1_vel", None), ("synth_cspr_s3", None), ("RR", "host-output")])
def test_batch_driver_over_several_devices(tmp_path, oracle_lib, name, mode):
    """BA::optimizeBatch with setDevices({0, 1, 2}): contiguous blocks of paths, one host thread and one device context per
    block, nothing exchanged -- every path comes back in place with the single-path files (device calls served by the
    checker library, whose contexts accept any device index)"""
    src = os.path.join(helpers.GOLD, name)
    _stage(src, tmp_path)
    cmd = [os.path.join(helpers.BUILD, "batest_batch_oracle"), "config.dat", "7", "--devices", "3"] + (["--" + mode] if mode else [])
    r = subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "sharded over 3 devices" in r.stdout and "7 paths, 0 failed" in r.stdout
    for d in ("out_first", "out_last"):
        assert filecmp.cmp(tmp_path / d / "s-sdot.dat", os.path.join(src, "ref_s-sdot.dat"), shallow=False)
        assert filecmp.cmp(tmp_path / d / "traj_out.dat", os.path.join(src, "ref_traj_out.dat"), shallow=False)


AUTORES = sorted(d for d in os.listdir(helpers.GOLD) if os.path.exists(os.path.join(helpers.GOLD, d, "autores.npz")))


@pytest.mark.parametrize("name", AUTORES)
def test_default_ba_with_automatic_integration_resolution(tmp_path, oracle_lib, name):
    """A BA object with _isAutoIntegRes left at the class default (reference ba.h:309; ba.cpp:493-556): the step-by-step API
    (interpInputData -> sweep x2 -> interpOutputData, one path) and optimizeBatch (three copies as one batch, every path with the
    integration step the rule derives from it) write the same files, and the step is the one of tests/golden/<case>/autores.npz"""
    import re
    import numpy as np
    src = os.path.join(helpers.GOLD, name)
    want = float(np.load(os.path.join(src, "autores.npz"))["integ_res"])
    one, many = tmp_path / "one", tmp_path / "many"
    one.mkdir(); many.mkdir()
    _stage(src, one); _stage(src, many)
    r1 = subprocess.run([os.path.join(helpers.BUILD, "batest_oracle"), "config.dat", "--auto-integ-res"], cwd=one, capture_output=True, text=True)
    m = re.findall(r"Final integ\. res is ([-0-9.naif]+) s", r1.stdout)
    assert m, r1.stdout[-2000:]
    if want != want:
        # a robot without Cartesian limits: the rule divides 0 by 0 and the reference would integrate with a NaN step
        assert "nan" in m[-1] and r1.returncode != 0
        rb = subprocess.run([os.path.join(helpers.BUILD, "batest_batch_oracle"), "config.dat", "3", "--auto-integ-res"], cwd=many, capture_output=True, text=True)
        assert rb.returncode != 0 and "3 failed" in rb.stdout
        return
    assert r1.returncode == 0, r1.stdout[-3000:]
    assert abs(float(m[-1]) - want) < 5.1e-7
    rb = subprocess.run([os.path.join(helpers.BUILD, "batest_batch_oracle"), "config.dat", "3", "--auto-integ-res"], cwd=many, capture_output=True, text=True)
    assert rb.returncode == 0, rb.stdout[-3000:]
    for d in ("out_first", "out_last"):
        assert filecmp.cmp(many / d / "s-sdot.dat", one / "s-sdot.dat", shallow=False)
        assert filecmp.cmp(many / d / "traj_out.dat", one / "traj_out.dat", shallow=False)


@pytest.mark.parametrize("what", ["cable robot without a cable limit", "JOINT path of a robot without forward kinematics"])
def test_configurations_the_device_resampler_refuses(tmp_path, oracle_lib, what):
    """BA::exportResampleParams covers the path kinds of the shipped examples and the BASELINE configs; the rest is refused with
    a message and -1 instead of taking another code path (INTEGRATION.md 2 has the table and what the reference does with each:
    a -1 of its own or undefined behaviour; BOTH paths without orientations, refused until round 5, are covered since round 6: golden
    case UR5_pos3)"""
    if what == "cable robot without a cable limit":
        _stage(os.path.join(helpers.GOLD, "synth_cspr_s3"), tmp_path)
        _edit_config(tmp_path / "config.dat", {"isJntVelConOn": 0, "isJntAccConOn": 0, "isTrqConOn": 0})
    else:
        _stage(os.path.join(helpers.GOLD, "synth_ur_s2"), tmp_path)
        _edit_config(tmp_path / "config.dat", {"robotTypeStr": "UR"})
    for tool, args in (("batest_oracle", ["config.dat"]), ("batest_batch_oracle", ["config.dat", "2"])):
        r = subprocess.run([os.path.join(helpers.BUILD, tool)] + args, cwd=tmp_path, capture_output=True, text=True)
        assert r.returncode != 0, (what, tool, r.stdout[-1500:])
        assert "not covered by the device resampler" in r.stdout, (what, tool, r.stdout[-1500:])
        assert not os.path.exists(tmp_path / "traj_out.dat")


@pytest.mark.parametrize("driver, copies", [("batest_oracle", None), ("batest_batch_oracle", "3")])
def test_resampler_results_are_used_only_when_two_evaluations_agree(tmp_path, oracle_lib, driver, copies):
    """BA::interpInputData and BA::optimizeBatch evaluate the resampler twice and compare (knots bit for bit resp. their checksums).
    With a fault injected into the checker's resampler (BATOTP_SHIM_RESAMPLE_FAULT: the named calls return knots with one value moved
    by an ulp and status 0 -- what round 4's unexplained event looked like): a single bad evaluation is reported, the path is evaluated
    again and the files are the reference's; when no two consecutive evaluations agree the result is refused"""
    src = os.path.join(helpers.GOLD, "synth_gen7dof_s0")
    cmd = [os.path.join(helpers.BUILD, driver), "config.dat"] + ([copies] if copies else [])
    outs = ("out_first", "out_last") if copies else (".",)
    # (evaluations are compared in consecutive pairs, four at most: "1" -> 2 and 3 agree; "2" -> 3 and 4 agree; "1,3" and "2,4" -> good and
    #  bad evaluations alternate, no pair agrees)
    for fault in ("", "1", "2", "1,3", "2,4"):
        work = tmp_path / f"run_{fault.replace(',', '_') or 'clean'}"
        work.mkdir()
        _stage(src, work)
        env = dict(os.environ)
        env["BATOTP_SHIM_RESAMPLE_FAULT"] = fault
        r = subprocess.run(cmd, cwd=work, capture_output=True, text=True, env=env)
        good = fault in ("", "1", "2")
        assert (r.returncode == 0) == good, (fault, r.stdout[-2000:])
        if fault:
            assert "disagree" in r.stdout, (fault, r.stdout[-2000:])
        else:
            assert "disagree" not in r.stdout
        if good:
            for d in outs:
                assert filecmp.cmp(work / d / "s-sdot.dat", os.path.join(src, "ref_s-sdot.dat"), shallow=False), fault
                assert filecmp.cmp(work / d / "traj_out.dat", os.path.join(src, "ref_traj_out.dat"), shallow=False), fault
        else:
            assert "no two consecutive evaluations" in r.stdout, (fault, r.stdout[-2000:])
        if not copies and fault == "2":
            # the one-path route traces its stages: the report names the first stage whose checksum differs (the injected fault looks
            # like the event of round 6: taught points and sites identical, one value of stage 2 moved) and keeps both evaluations'
            # arrays of that stage -- evaluation 2 against 1, then 3 against 2
            import numpy as np
            assert "stage checksums of the two evaluations" in r.stdout and "first difference in stage 2 (their second derivatives)" in r.stdout, r.stdout[-3000:]
            ev = [np.fromfile(work / f"resampler_disagreement_stage2_eval{k}.bin", dtype=np.uint64) for k in (1, 2, 3)]
            assert ev[0].size == ev[1].size == ev[2].size > 0
            assert np.array_equal(ev[0], ev[2]) and int((ev[0] != ev[1]).sum()) == 1
            assert f"{ev[0].size} / {ev[0].size} values, 1 differ, first at {ev[0].size // 2}, last at {ev[0].size // 2}" in r.stdout, r.stdout[-3000:]
        if not copies and not fault:
            assert not [f for f in os.listdir(work) if f.startswith("resampler_disagreement")]
