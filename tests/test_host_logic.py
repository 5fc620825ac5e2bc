"""CPU: host-side pieces -- synthetic path generator determinism, config writer / parser round trip,
oracle building blocks against independent computations."""
import ctypes as C
import os
import subprocess

import numpy as np

import helpers
from batotp_amd import capi, pathgen


def test_splitmix_is_deterministic_and_uniform():
    a = pathgen.splitmix64_uniform(42, 1000)
    b = pathgen.splitmix64_uniform(42, 1000)
    assert np.array_equal(a, b) and a.min() >= 0 and a.max() < 1 and abs(a.mean() - 0.5) < 0.05
    assert not np.array_equal(a, pathgen.splitmix64_uniform(43, 1000))


def test_generated_inputs_are_float32_paths():
    th = pathgen.gen7dof_fine(3, 12)
    assert th.dtype == np.float32 and th.shape == (7, 240)
    ca = pathgen.cspr_fine(3, 6)
    assert ca.shape[0] == 3 and ca.shape[1] == 1001


def test_config_writer_round_trips_through_the_host_parser(tmp_path):
    th = pathgen.gen7dof_fine(9, 10)
    pathgen.write_traj_bin(str(tmp_path / "path.dat"), 0.01, th, None)
    pathgen.write_config(str(tmp_path / "config.dat"), robot="GENJNT", is_parallel=0, n_joints=7, n_cart=3, traj_file="path.dat",
                         is_bin=1, path_type="JOINT", degrees=0, jnt_vel=[5, 4, 3, 2, 1, 6, 7], jnt_acc_on=1, jnt_acc=[10] * 7,
                         integ_res=0.0125, max_integ_time=123.0, theta_res=0.1, theta_res2=0.1)
    r = subprocess.run([os.path.join(helpers.BUILD, "dump_knots"), "config.dat"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    prob = capi.Problem.from_buffer_copy(open(tmp_path / "problem.bin", "rb").read())
    assert prob.n_joints == 7 and prob.robot_type == capi.ROBOT_GENJNT
    assert list(prob.jnt_vel_max)[:7] == [5, 4, 3, 2, 1, 6, 7]
    assert prob.integ_res == 0.0125 and prob.max_integ_time == 123.0
    assert prob.flags & capi.F_JNT_ACC_ON and not (prob.flags & capi.F_TRQ_ON)
    kb = open(tmp_path / "knots.bin", "rb").read()
    N, nJ, nC = (int(v) for v in np.frombuffer(kb, "<i8", 3, 0))
    y = np.frombuffer(kb, "<f8", (nJ + nC) * N, 32).reshape(nJ + nC, N)
    # resampled to (almost) constant joint-space spacing thetaNormRes2
    d = np.linalg.norm(np.diff(y[:7], axis=1), axis=0)
    assert N > 50 and np.all(d[:-1] < 0.1 * 1.02) and np.median(d) > 0.09


def _oracle():
    lib = C.CDLL(os.path.join(helpers.BUILD, "libbatotp_oracle.so"))
    return lib


def test_oracle_spline_interpolates_and_matches_its_own_derivative_relations(oracle_lib):
    lib = _oracle()
    n = 200
    x = np.linspace(0, 6, n)
    y = np.ascontiguousarray(np.sin(x))
    c = np.zeros((4, n))
    lib.bo_spline_coeffs.argtypes = [C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_double), C.c_int]
    lib.bo_spline_coeffs(y.ctypes.data_as(C.POINTER(C.c_double)), n, c.ctypes.data_as(C.POINTER(C.c_double)), 0)
    c0, c1, c2, c3 = c
    assert np.array_equal(c0[:-1], y[:-1])                              # interpolation at the left knots
    assert np.allclose(c0[:-1] + c1[:-1] + c2[:-1] + c3[:-1], y[1:], atol=1e-13)   # and at the right knots
    # C1 / C2 continuity at the interior knots
    assert np.allclose(c1[:-2] + 2 * c2[:-2] + 3 * c3[:-2], c1[1:-1], atol=1e-12)
    assert np.allclose(2 * c2[:-2] + 6 * c3[:-2], 2 * c2[1:-1], atol=1e-12)
    assert c2[0] == 0.0                                                 # natural left end
    assert np.all(c[:, -1] == 0.0)                                      # the last row is never written


def test_oracle_lu_and_quadratic(oracle_lib):
    lib = _oracle()
    lib.bo_solve_lin_sys.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    rng = np.random.default_rng(0)
    for _ in range(50):
        A = np.ascontiguousarray(rng.standard_normal((3, 3)))
        b = np.ascontiguousarray(rng.standard_normal(3))
        x = np.zeros(3)
        lib.bo_solve_lin_sys(3, A.ctypes.data_as(C.POINTER(C.c_double)), b.ctypes.data_as(C.POINTER(C.c_double)), x.ctypes.data_as(C.POINTER(C.c_double)))
        assert np.allclose(A @ x, b, atol=1e-9)
    lib.bo_solve_quadratic.argtypes = [C.c_double, C.c_double, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    s1, s2 = C.c_double(), C.c_double()
    assert lib.bo_solve_quadratic(1.0, -3.0, 2.0, C.byref(s1), C.byref(s2)) == 0 and {s1.value, s2.value} == {1.0, 2.0}
    assert lib.bo_solve_quadratic(1.0, 0.0, 1.0, C.byref(s1), C.byref(s2)) == -1
    assert lib.bo_solve_quadratic(0.0, 0.0, 1.0, C.byref(s1), C.byref(s2)) == -2


def test_edge_inputs_minimum_knots_and_short_sweeps(oracle_ctx):
    """smallest legal path (4 knots) and a path traversed in fewer than 4 steps (re-interpolation branch)"""
    prob = capi.make_problem(2, 0, flags=capi.F_JNT_ACC_ON, jnt_vel_max=[1e3, 1e3], jnt_acc_max=[1e6, 1e6], integ_res=0.05)
    y = np.array([[0.0, 0.1, 0.2, 0.3], [0.0, 0.05, 0.1, 0.15]])
    b = capi.Batch(oracle_ctx, prob, [4], 64)
    b.upload_knots(0, [y], [0.1])
    b.precompute(0); b.pointwise_mvc(); b.sweep(-1); b.sweep(1)
    r = b.results()[0]
    assert r["n_rev"] >= 4 and r["n_fwd"] >= 4
    if r["steps_fwd"] < 3:
        assert r["status_fwd"] & capi.ST_SHORT and r["n_fwd"] == 4
    s, sd = b.curve(0, 1)
    assert s[0] == 0.0 and abs(s[-1] - 0.3) < 1e-15 and np.all(np.diff(s) >= 0)
    b.close()


def test_which_configurations_the_device_stages_cover():
    """BA::exportResampleParams / exportOutputParams decide per configuration whether the stage before / after the hot
    path runs on the device; the fixtures (written by oracle/make_resample_fixtures.py from those decisions) pin the matrix"""
    import helpers
    assert set(helpers.RESAMPLE_CASES) == {"CSPR3DOF", "CSPR3DOF_par", "GEN7DOF", "synth_cspr_s3", "synth_cspr_s5", "synth_cspr_s9_dup",
                                           "synth_cspr_s11_decim", "synth_gen7dof_s0", "synth_gen7dof_s1_vel", "synth_gen7dof_s6_dup",
                                           "synth_gen7dof_s10_decim", "synth_ur_s2",
                                           # round 3: robots with forward kinematics, serial-robot torque recomputation,
                                           # pose paths (path type BOTH: axis-angle <-> quaternion)
                                           "KUKA-LWR-IV", "KUKA_cartacc", "RR", "RR_acc", "UR5", "UR5_nocartacc",
                                           # the cable robot with solveLinSys through the Jacobi SVD (isSVD = 1)
                                           "CSPR3DOF_svd", "CSPR3DOF_par_svd",
                                           # round 6: joints and tool POSITIONS taught together (path type BOTH with nCart = 3, no orientations)
                                           "UR5_pos3"}
    # paths whose s is the teach time (sWeights 1 0 0) need no resampling stage: adjust_s returns at once (ba.cpp:416) and the
    # knots are the taught points after the host filters (BA::keepTaughtSpacing); the output stage runs on the device for them too
    teach_time = {"synth_gen7dof_s14_teachtime", "synth_gen7dof_s15_teachtime_decim2"}
    assert set(helpers.RESAMPLE_CASES) | teach_time == set(helpers.FULL_CASES)   # every shipped example and edited variant
    assert set(helpers.OUTPUT_CASES) == set(helpers.FULL_CASES)
