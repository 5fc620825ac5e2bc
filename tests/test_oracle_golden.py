"""CPU: the oracle (behind the C-ABI shim) against golden vectors produced by the reference itself.

Golden vectors = outputs of the reference's prebuilt bin/batest run in the build container by
oracle/make_golden.py on the reference's five shipped examples, on edited configurations that
reach the other constraint branches, and on synthetic inputs in the shapes of the BASELINE configs.
"""
import numpy as np
import pytest

import helpers
from helpers import Case, run_pipeline, assert_matches_reference


@pytest.mark.parametrize("name", helpers.FULL_CASES)
def test_oracle_reproduces_reference_curves(oracle_ctx, name):
    case = Case(name)
    assert case.n == case.expected["n_knots"]
    out = run_pipeline(oracle_ctx, [case])[0]
    assert_matches_reference(case, out)
    # sanity of the published curves
    for s, sd in (out["rev"], out["fwd"]):
        assert np.all(np.diff(s) >= 0) and s[0] == 0.0
        assert np.all(np.isfinite(sd))


@pytest.mark.parametrize("name", helpers.DIGEST_CASES)
def test_oracle_reproduces_reference_digest(oracle_ctx, name):
    """BASELINE-size single paths: step counts, T and the sha256 of the float32 curves"""
    case = Case(name)
    assert case.n == case.expected["n_knots"]
    out = run_pipeline(oracle_ctx, [case], mvc=False, details=False)[0]
    assert_matches_reference(case, out)
    z = case.sampled
    assert np.array_equal(z["fwd_s"], out["fwd"][0].astype(np.float32)[::64])
    assert np.array_equal(z["fwd_sd"], out["fwd"][1].astype(np.float32)[::64])


def _oracle_batch(oracle_ctx, case):
    from batotp_amd import capi
    b = capi.Batch(oracle_ctx, case.problem, [case.n], case.max_steps())
    b.upload_knots(0, [case.y], [case.sres])
    helpers.precompute_with_trig(oracle_ctx, b, case.problem, 1, None)
    return b


@pytest.mark.parametrize("name", helpers.FULL_CASES)
def test_oracle_replays_the_reference_per_point_known_answers(oracle_ctx, name):
    """SURVEY.md 8c iii: ~250 sampled calls of BA::applyAccelConstraintsBisectionPt (ba.cpp:1248-1332) and ~300 of BA::sdotLim
    (ba.cpp:1204-1236) per case, cursor state on entry and results on return read out of the running reference binary
    (oracle/make_golden_f64.py): the oracle's routines, started from the same state, must leave the same fp64 bits"""
    import ctypes as C
    import os
    case = Case(name)
    z = np.load(os.path.join(case.dir, "ref_point_kats.npz"))
    ref = np.load(os.path.join(case.dir, "ref_curves_f64.npz"))
    lib = oracle_ctx.library.lib
    b = _oracle_batch(oracle_ctx, case)
    D = C.POINTER(C.c_double)
    lib.batotp_oracle_kat_accel.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_int64, C.c_double, D,
                                            C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.batotp_oracle_kat_sdot_lim.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, D, C.c_double, D, D,
                                               C.c_int64, C.c_int64, D]
    assert z["accel"].size >= 200 and z["sdot_lim"].size >= 200
    n_fail = 0
    for k in z["accel"]:
        out5 = (C.c_double * 5)()
        n_iter, rc = C.c_int32(0), C.c_int32(0)
        assert lib.batotp_oracle_kat_accel(b.handle, 0, int(k["dir"]), float(k["s_cur"]), float(k["sdot_in"]), int(k["seg_in"]),
                                           float(k["sddot_in"]), out5, C.byref(n_iter), C.byref(rc)) == 0
        got = np.array(out5[:4])
        want = np.array([k["sdot_out"], k["sddot_out"], k["sddot_l"], k["sddot_h"]])
        helpers.assert_bit_equal(got, want, f"{name}: applyAccelConstraintsBisectionPt at s = {float(k['s_cur'])!r}")
        assert (n_iter.value, rc.value, int(out5[4])) == (int(k["n_iter"]), int(k["rc"]), int(k["seg_out"])), (name, k)
        n_fail += rc.value != 0
    rev_s, rev_sd = np.ascontiguousarray(ref["rev_s"]), np.ascontiguousarray(ref["rev_sd"])
    for k in z["sdot_lim"]:
        out2 = (C.c_double * 2)()
        th = np.ascontiguousarray(k["theta_d_pt"], dtype=np.float64)
        fwd = int(k["dir"]) == 1
        assert lib.batotp_oracle_kat_sdot_lim(b.handle, 0, int(k["dir"]), float(k["s_cur"]), float(k["sdot_in"]), float(k["sdot_min"]),
                                              th.ctypes.data_as(D), float(k["cart0"]),
                                              rev_s.ctypes.data_as(D) if fwd else None, rev_sd.ctypes.data_as(D) if fwd else None,
                                              rev_s.size if fwd else 0, int(k["seg_mvc_in"]), out2) == 0
        helpers.assert_bit_equal(np.array([out2[0]]), np.array([k["sdot_out"]]), f"{name}: sdotLim at s = {float(k['s_cur'])!r}")
        if fwd:
            assert int(out2[1]) == int(k["seg_mvc_out"]), (name, k)
    b.close()


def test_bisection_failure_branch_is_pinned():
    """one golden case makes the reference print a bisection failure (stale sddot, ignored -1)"""
    assert Case.__init__  # keep import
    import json, os
    e = json.load(open(os.path.join(helpers.GOLD, "synth_cspr_s8_40k", "expected.json")))
    assert e["ref_bisect_fail_msgs"] >= 1


def test_checker_mirrors_the_state_rules_of_in_place_curves(oracle_ctx):
    """BATOTP_F_CURVES_IN_PLACE through the checker library: same results (it keeps both curves), and the product's state rules
    -- after the forward sweep the reverse curve cannot be fetched and the forward sweep cannot be repeated"""
    from batotp_amd import capi
    case = Case("GEN7DOF")
    plain = run_pipeline(oracle_ctx, [case], mvc=False, details=False)[0]
    flagged = run_pipeline(oracle_ctx, [case], mvc=False, details=False, extra_flags=capi.F_CURVES_IN_PLACE)[0]
    for f in plain["result"].dtype.names:
        assert plain["result"][f] == flagged["result"][f], f
    for which in ("rev", "fwd"):
        assert np.array_equal(plain[which][0], flagged[which][0]) and np.array_equal(plain[which][1], flagged[which][1])
    prob = capi.Problem.from_buffer_copy(bytes(case.problem))
    prob.flags |= capi.F_CURVES_IN_PLACE
    b = capi.Batch(oracle_ctx, prob, [case.n], case.max_steps() + 16)
    b.upload_knots(0, [case.y], [case.sres])
    b.optimize()
    with pytest.raises(capi.BatotpError):
        b.curve(0, -1)
    with pytest.raises(capi.BatotpError):
        b.sweep(+1)
    b.sweep(-1)
    assert len(b.curve(0, -1)[0]) == int(plain["result"]["n_rev"])
    b.close()


def test_checker_mirrors_the_state_rules_of_pointwise_values_in_the_curve_slots(oracle_ctx):
    """BATOTP_F_MVC_IN_CURVES through the checker library: a pointwise evaluation after a sweep invalidates both curves (they
    shared the slots with its values) until their sweeps have run again; in-place forward sweeps give up within 64 points of
    the unread reverse points exactly as the kernels do (bo_sweep_ex)"""
    from batotp_amd import capi
    case = Case("GEN7DOF")
    prob = capi.Problem.from_buffer_copy(bytes(case.problem))
    prob.flags |= capi.F_MVC_IN_CURVES
    b = capi.Batch(oracle_ctx, prob, [case.n], case.max_steps())
    b.upload_knots(0, [case.y], [case.sres])
    b.optimize()
    n_rev, n_fwd = len(b.curve(0, -1)[0]), len(b.curve(0, +1)[0])
    b.pointwise_mvc()
    for which in (-1, +1):
        with pytest.raises(capi.BatotpError):
            b.curve(0, which)
    b.sweep(-1)
    assert len(b.curve(0, -1)[0]) == n_rev
    with pytest.raises(capi.BatotpError):
        b.curve(0, +1)
    b.sweep(+1)
    assert len(b.curve(0, +1)[0]) == n_fwd
    b.close()
    # the capacity margin of the shared buffer: room for the forward curve + 72 points is enough, + 8 is not
    prob = capi.Problem.from_buffer_copy(bytes(case.problem))
    prob.flags |= capi.F_CURVES_IN_PLACE
    for extra, ok in ((72, True), (8, False)):
        b = capi.Batch(oracle_ctx, prob, [case.n], n_fwd + extra)
        b.upload_knots(0, [case.y], [case.sres])
        b.optimize()
        r = b.results()[0]
        assert int(r["n_rev"]) == n_rev
        if ok:
            assert int(r["n_fwd"]) == n_fwd
        else:
            assert int(r["n_fwd"]) == 0 and (int(r["status_fwd"]) & capi.ST_CAPACITY) and int(r["steps_fwd"]) + 1 < n_fwd + extra
        b.close()


def test_second_derivatives_of_the_spline_kat_are_those_of_the_pinned_rows(oracle_ctx):
    """the checker side of batotp_hip_spline_lanes_kat (oracle: bo_spline_sol) against the coefficient rows the golden cases pin:
    c2 of a row is sol / 2 exactly, c3 the difference of two neighbours over 6"""
    from batotp_amd import capi
    case = Case("synth_gen7dof_s0")
    b = capi.Batch(oracle_ctx, case.problem, [case.n], 64)
    b.upload_knots(0, [case.y], [case.sres])
    b.precompute(1)
    for ch in range(case.problem.n_joints):
        rows = b.coeffs(0, ch)
        sol, seq, redone = capi.spline_lanes_kat(oracle_ctx, np.ascontiguousarray(case.y[ch]))
        assert redone == 0 and sol.tobytes() == seq.tobytes()
        assert sol[0] == 0.0
        assert (sol[:-1] / 2.0).tobytes() == rows[2][:-1].tobytes()
        assert ((sol[1:] - sol[:-1]) / 6.0).tobytes() == rows[3][:-1].tobytes()
    b.close()
