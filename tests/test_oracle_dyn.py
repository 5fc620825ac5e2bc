"""CPU: the serial-chain dynamics of BASELINE config 3 (7-DOF arm with torque limits) in the oracle.

The reference has no dynamics model for this robot (Robot::dynSerial, reference robot.cpp:349-360, knows the two-link
arm only): PARITY UNPINNED for the arm model.  What can be checked, and is checked here:
  * the recursion against the reference's one serial model, Robot::dynRR (robot.cpp:377-431), with that model's
    two point masses as the link table;
  * the link table's geometry against the reference's forward kinematics of the same arm (fwdKinKuka,
    robot.cpp:105-165);
  * the recursion against an independent formulation (Lagrange's equations by numerical differentiation of the
    kinetic and potential energy) on the 7-link table;
  * regression fixtures produced by this repository's own oracle (tests/golden_self, oracle/make_selfgolden.py).
"""
import ctypes as C
import math

import numpy as np
import pytest

import helpers
from helpers import Case, run_pipeline, assert_matches_reference
from batotp_amd import capi

D = C.POINTER(C.c_double)


def _rnea(lib, model, q, qd, qdd, a0):
    n = model.n_links
    cq = np.array([math.cos(v) for v in q]); sq = np.array([math.sin(v) for v in q])
    qd, qdd, a0 = (np.ascontiguousarray(v, dtype=np.float64) for v in (qd, qdd, a0))
    tau = np.zeros(n)
    lib.lib.bo_rnea.restype = None
    lib.lib.bo_rnea(C.byref(model), cq.ctypes.data_as(D), sq.ctypes.data_as(D), qd.ctypes.data_as(D), qdd.ctypes.data_as(D),
                    a0.ctypes.data_as(D), tau.ctypes.data_as(D))
    return tau


def test_recursion_reproduces_the_reference_two_link_arm(oracle_lib, oracle_ctx):
    """RR golden case twice through the oracle: with the reference's closed form (dynRR) and with the same two point
    masses as a link table.  a1, a3, a4 and row 0 of a2 agree to rounding.  Row 1 of a2 differs by exactly
    ccFact * dth1^2: the reference carries the centrifugal term of joint 2 with the sign of joint 1's
    (robot.cpp:420: "-.5*ccFact*dth1*dth1"; the textbook two-link arm has +h q1dot^2 there) -- the closed form is kept
    as it is for RR (bit parity with the reference), the recursion is the textbook one."""
    case = Case("RR")
    ref = run_pipeline(oracle_ctx, [case], mvc=False)[0]
    model = oracle_lib.builtin_serial_model(capi.ROBOT_RR)
    chain = run_pipeline(oracle_ctx, [case], mvc=False, serial_model=model)[0]
    a_ref, a_chain = ref["dyn"], chain["dyn"]          # [4][2][N]
    scale = np.abs(a_ref).max(axis=2, keepdims=True) + 1e-300
    for k, row in ((0, 0), (0, 1), (1, 0), (2, 0), (2, 1), (3, 0), (3, 1)):
        assert np.max(np.abs(a_chain[k][row] - a_ref[k][row]) / scale[k][row]) < 1e-12, (k, row)
    d2r = 3.14159265358979323846 / 180.0
    th2 = d2r * ref["samp"][1][0]
    dth1 = d2r * ref["samp"][0][1]
    cc = 8 * .4 * .6 * np.sin(th2)
    assert np.max(np.abs((a_chain[1][1] - a_ref[1][1]) - cc * dth1 * dth1)) < 1e-12 * scale[1][1].max()


def _rod(a, q):
    a = np.array(a)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.cos(q) * np.eye(3) + np.sin(q) * K + (1 - np.cos(q)) * np.outer(a, a)


def _frames(model, q):
    """joint origins and link orientations of the chain at q (base coordinates)"""
    R, p, out = np.eye(3), np.zeros(3), []
    for i in range(model.n_links):
        L = model.link[i]
        p = p + R @ np.array(L.off)
        R = R @ _rod(L.axis, q[i])
        out.append((p.copy(), R.copy()))
    return out


def test_kuka_link_table_is_the_chain_of_the_reference_forward_kinematics(oracle_lib):
    """tool point through the link table == fwdKinKuka's closed form (reference robot.cpp:105-174)"""
    model = oracle_lib.builtin_serial_model(capi.ROBOT_KUKA)
    assert model.n_links == 7 and model.degrees == 1
    rng = np.random.default_rng(3)
    for _ in range(20):
        t = rng.uniform(-2.5, 2.5, 7)
        c, s = np.cos(t), np.sin(t)
        c1, c2, c3, c4, c5, c6, c7 = c; s1, s2, s3, s4, s5, s6, s7 = s
        Q12 = np.array([[c1 * c2, -s1, -c1 * s2], [c2 * s1, c1, -s1 * s2], [s2, 0, c2]])
        Q34 = np.array([[c3 * c4, -s3, c3 * s4], [c4 * s3, c3, s3 * s4], [-s4, 0, c4]])
        Q567 = np.array([[c5 * c6 * c7 - s5 * s7, -c7 * s5 - c5 * c6 * s7, -c5 * s6],
                         [c5 * s7 + c6 * c7 * s5, c5 * c7 - c6 * s5 * s7, -s5 * s6], [c7 * s6, -s6 * s7, c6]])
        Q1234 = Q12 @ Q34
        Q = Q1234 @ Q567
        elbow = .4 * Q12[:, 2] + np.array([0, 0, .3105])
        tool = elbow + .39 * Q1234[:, 2] + Q @ np.array([0, -.08, .545])
        p7, R7 = _frames(model, t)[-1]
        assert np.max(np.abs(p7 + R7 @ np.array([0, -.08, .545]) - tool)) < 1e-14


def _energies(model, q, qd):
    """kinetic and potential energy of the chain from its frames (independent of the recursion): velocities of the
    centres of mass and angular velocities by central differences of the kinematics along qd"""
    def com_and_R(qq):
        return [(p + R @ np.array(model.link[i].com), R) for i, (p, R) in enumerate(_frames(model, qq))]
    h = 1e-6
    plus, minus, mid = com_and_R(q + h * qd), com_and_R(q - h * qd), com_and_R(q)
    T = U = 0.0
    g = np.array(model.gravity)
    for i in range(model.n_links):
        L = model.link[i]
        v = (plus[i][0] - minus[i][0]) / (2 * h)
        Rd = (plus[i][1] - minus[i][1]) / (2 * h)
        W = mid[i][1].T @ Rd                      # skew(omega) in link coordinates
        w = np.array([W[2, 1], W[0, 2], W[1, 0]])
        I = np.array([[L.inertia[0], L.inertia[3], L.inertia[4]], [L.inertia[3], L.inertia[1], L.inertia[5]],
                      [L.inertia[4], L.inertia[5], L.inertia[2]]])
        T += 0.5 * L.mass * v @ v + 0.5 * w @ I @ w
        U += -L.mass * g @ mid[i][0]
    return T, U


def test_recursion_satisfies_lagranges_equations_on_the_seven_link_arm(oracle_lib):
    """tau = d/dt (dT/dqd) - dT/dq + dU/dq along a smooth motion q(t), all derivatives numerical"""
    model = oracle_lib.builtin_serial_model(capi.ROBOT_KUKA)
    n = 7
    rng = np.random.default_rng(11)
    for _ in range(3):
        A, w0, ph = rng.uniform(0.3, 1.0, n), rng.uniform(0.5, 2.0, n), rng.uniform(0, 6, n)
        q_of = lambda t: A * np.sin(w0 * t + ph)
        qd_of = lambda t: A * w0 * np.cos(w0 * t + ph)
        qdd_of = lambda t: -A * w0 * w0 * np.sin(w0 * t + ph)
        t0, e = 0.37, 1e-4

        def dT_dqd(t):
            q, qd = q_of(t), qd_of(t)
            return np.array([(_energies(model, q, qd + e * np.eye(n)[j])[0] - _energies(model, q, qd - e * np.eye(n)[j])[0]) / (2 * e)
                             for j in range(n)])
        ddt = (dT_dqd(t0 + e) - dT_dqd(t0 - e)) / (2 * e)
        q, qd, qdd = q_of(t0), qd_of(t0), qdd_of(t0)
        dL_dq = np.array([((lambda p, m: (p[0] - p[1]) - (m[0] - m[1]))(_energies(model, q + e * np.eye(n)[j], qd),
                                                                        _energies(model, q - e * np.eye(n)[j], qd))) / (2 * e)
                          for j in range(n)])
        tau_lagrange = ddt - dL_dq
        g0 = -np.array(model.gravity)
        tau = _rnea(oracle_lib, model, q, qd, qdd, g0)
        assert np.max(np.abs(tau - tau_lagrange)) < 2e-4 * max(1.0, np.abs(tau).max()), (tau, tau_lagrange)
        # the three passes the coefficients are made of add up to the full inverse dynamics at sdot = 1, sddot = 0
        z = np.zeros(n)
        parts = _rnea(oracle_lib, model, q, qd, qdd, [0, 0, 0]) + _rnea(oracle_lib, model, q, z, z, g0)
        assert np.max(np.abs(parts - tau)) < 1e-12 * max(1.0, np.abs(tau).max())


def test_mass_matrix_from_the_recursion_is_symmetric_positive_definite(oracle_lib):
    model = oracle_lib.builtin_serial_model(capi.ROBOT_KUKA)
    rng = np.random.default_rng(5)
    for _ in range(5):
        q = rng.uniform(-2, 2, 7)
        z = np.zeros(7)
        M = np.stack([_rnea(oracle_lib, model, q, z, np.eye(7)[j], [0, 0, 0]) for j in range(7)], axis=1)
        assert np.max(np.abs(M - M.T)) < 1e-13
        assert np.linalg.eigvalsh(0.5 * (M + M.T)).min() > 0


@pytest.mark.parametrize("name", helpers.SELF_CASES)
def test_oracle_reproduces_its_own_fixtures(oracle_ctx, name):
    """self-generated fixtures (no reference model exists for this robot: parity unpinned) -- a change of the chain
    model, of the oracle or of the host trig policy shows up here"""
    case = Case(name)
    assert "NONE" in case.expected["reference"]
    assert case.n == case.expected["n_knots"]
    out = run_pipeline(oracle_ctx, [case])[0]
    assert_matches_reference(case, out)
    assert np.all(np.isfinite(out["dyn"]))
