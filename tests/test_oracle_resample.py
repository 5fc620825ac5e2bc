"""Pins the oracle's restatement of the path resampling to the golden knots: the knots
it produces from the taught points must be byte-identical to knots.npz -- the knots behind the
s-sdot / trajectory outputs that match the reference binary byte for byte."""
import os

import numpy as np
import pytest

from batotp_amd import capi
import helpers
from helpers import RESAMPLE_CASES, ResampleCase


@pytest.fixture(scope="module")
def octx():
    return capi.Context(helpers.load_oracle(), 0)


@pytest.mark.parametrize("name", RESAMPLE_CASES)
def test_oracle_resample_matches_golden_knots(octx, name):
    c = ResampleCase(name)
    r = capi.Resampled(octx, c.params, [c.x], [c.sres_in])
    assert int(r.status[0]) == 0
    assert int(r.n_knots[0]) == c.y.shape[1]
    assert r.sres[0] == c.sres
    assert r.knots(0).tobytes() == c.y.tobytes()
    r.close()


def test_oracle_resample_batch_and_ragged(octx):
    cs = [ResampleCase(n) for n in RESAMPLE_CASES if n.startswith("synth_cspr")]
    ref = bytes(ResampleCase("synth_cspr_s3").params)
    cs = [c for c in cs if bytes(c.params) == ref]   # one parameter set per call
    assert len(cs) >= 3
    r = capi.Resampled(octx, cs[0].params, [c.x for c in cs], [c.sres_in for c in cs])
    for k, c in enumerate(cs):
        assert r.knots(k).tobytes() == c.y.tobytes() and r.sres[k] == c.sres
    r.close()


def test_oracle_resample_status_paths(octx):
    c = ResampleCase("synth_gen7dof_s0")
    # all points identical -> the reference's "no optimization will be performed" exit
    x = np.repeat(c.x[:, :1], 16, axis=1)
    r = capi.Resampled(octx, c.params, [x, c.x], [c.sres_in, c.sres_in])
    assert int(r.status[0]) != 0 and int(r.n_knots[0]) == 4
    assert int(r.status[1]) == 0 and r.knots(1).tobytes() == c.y.tobytes()
    r.close()
    # close points are dropped before resampling: duplicating taught points must not change the result
    xd = np.repeat(c.x, 2, axis=1)
    r = capi.Resampled(octx, c.params, [xd], [c.sres_in])
    assert int(r.status[0]) == 0
    assert r.knots(0).tobytes() == c.y.tobytes()
    r.close()


def test_unsupported_kind_is_refused(octx):
    c = ResampleCase("synth_gen7dof_s0")
    p = capi.ResampleParams.from_buffer_copy(bytes(c.params))
    p.robot_type = capi.ROBOT_UR
    with pytest.raises(capi.BatotpError):
        capi.Resampled(octx, p, [c.x], [c.sres_in])


AUTORES_CASES = sorted(d for d in os.listdir(helpers.GOLD) if os.path.exists(os.path.join(helpers.GOLD, d, "autores.npz")))


def autores_params(c, problem):
    """the resampling parameters of a golden case with the automatic integration resolution switched on (the reference's
    class default, ba.h:309) and the inputs its rule reads (ba.cpp:493-556)"""
    p = capi.ResampleParams.from_buffer_copy(bytes(c.params))
    p.flags |= capi.RS_AUTO_INTEG_RES
    for j in range(p.n_joints):
        p.jnt_vel_max[j], p.jnt_acc_max[j] = problem.jnt_vel_max[j], problem.jnt_acc_max[j]
    p.cart_vel_max, p.cart_acc_max, p.quad_rad_thresh = problem.cart_vel_max, problem.cart_acc_max, problem.quad_rad_thresh
    cfg = open(os.path.join(helpers.GOLD, c.name, "config.dat")).read().splitlines()
    p.degrees = int([l for l in cfg if "areJntAnglesDegrees" in l or "areJointAnglesDegrees" in l][0].split()[0])
    return p


@pytest.mark.parametrize("name", AUTORES_CASES)
def test_automatic_integration_resolution(octx, name):
    """reference ba.cpp:462-470, 493-556 (class default ba.h:309; batest switches it off, so the reference binary cannot pin it):
    knots, integration step, s weights and scale type against tests/golden/<case>/autores.npz, written once by the round-3
    statement-level host restatement of adjust_s (oracle/make_autores_fixtures.py)"""
    c = ResampleCase(name)
    z = np.load(os.path.join(helpers.GOLD, name, "autores.npz"))
    p = autores_params(c, helpers.Case(name).problem)
    r = capi.Resampled(octx, p, [c.x], [c.sres_in])
    assert int(r.status[0]) == 0
    integ, sw, st = r.auto()
    assert np.array_equal(np.array([integ[0]]).view(np.uint64), np.array([float(z["integ_res"])]).view(np.uint64)), (integ[0], float(z["integ_res"]))
    assert sw[0].tobytes() == np.asarray(z["s_weights"], dtype=np.float64).tobytes(), (sw[0], z["s_weights"])
    assert int(st[0]) == int(z["scale_type"])
    assert float(r.sres[0]) == float(z["sres"]) and int(r.n_knots[0]) == z["y"].shape[1]
    assert r.knots(0).tobytes() == np.ascontiguousarray(z["y"]).tobytes()
    r.close()


def test_knot_checksums_are_the_sums_the_header_describes(oracle_ctx):
    """batotp_hip_resampled_checksums (include/batotp_hip.h): per path the wrap-around sum of splitmix64-finalised (bits XOR (index + 1) *
    golden ratio) over the knot values as they lie in memory; 0 for a path with a non-zero status"""
    c = ResampleCase("synth_gen7dof_s0")
    same = c.x.copy(); same[:, :] = same[:, :1]          # all points identical: the resampler reports a status
    r = capi.Resampled(oracle_ctx, c.params, [c.x, c.x[:, : c.x.shape[1] // 2].copy(), same], [c.sres_in] * 3)
    sums = r.checksums()
    assert int(r.status[2]) != 0 and int(sums[2]) == 0

    def mix(x):
        x = x ^ (x >> np.uint64(30)); x = x * np.uint64(0xBF58476D1CE4E5B9)
        x = x ^ (x >> np.uint64(27)); x = x * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))
    with np.errstate(over="ignore"):
        for k in range(2):
            bits = np.ascontiguousarray(r.knots(k)).reshape(-1).view(np.uint64)
            idx = (np.arange(bits.size, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
            assert int(mix(bits ^ idx).sum(dtype=np.uint64)) == int(sums[k]), k
    assert int(sums[0]) != int(sums[1])
    r.close()


def test_checker_keeps_the_stage_trace_of_one_path_calls(oracle_lib):
    """batotp_hip_set_resample_trace through the checker (its resampler's stages observed: bo_resample_set_stage_observer): eight stage
    checksums of a one-path call, the same from call to call, the last one the checksum of the knots; stages 2 and 3 are kept as arrays;
    the knots are those of an untraced call; a call with more than one path keeps none (the product: tests/test_gpu_resample.py)"""
    ctx = capi.Context(oracle_lib, 0)
    plain = {}
    for name in ("synth_gen7dof_s0", "synth_cspr_s3", "UR5_pos3"):
        c = ResampleCase(name)
        r = capi.Resampled(ctx, c.params, [c.x], [c.sres_in])
        plain[name] = r.knots(0).tobytes()
        with pytest.raises(capi.BatotpError):
            r.trace()
        r.close()
    ctx.set_resample_trace(True)
    seen = set()
    for name in plain:
        c = ResampleCase(name)
        per_case = []
        for _ in range(2):
            r = capi.Resampled(ctx, c.params, [c.x], [c.sres_in])
            t = r.trace()
            assert int(t[7]) == int(r.checksums()[0]) and all(int(v) != 0 for v in t), (name, t)
            assert r.trace_data(2).size > 0 and r.trace_data(3).size % c.y.shape[0] == 0 and r.trace_data(0).size == 0
            assert r.knots(0).tobytes() == plain[name]
            per_case.append(t.tobytes())
            r.close()
        assert len(set(per_case)) == 1, name
        seen.add(per_case[0])
    assert len(seen) == 3
    c = ResampleCase("synth_gen7dof_s0")
    r = capi.Resampled(ctx, c.params, [c.x, c.x], [c.sres_in] * 2)
    with pytest.raises(capi.BatotpError):
        r.trace()
    r.close()
    ctx.set_resample_trace(False)
    ctx.close()
