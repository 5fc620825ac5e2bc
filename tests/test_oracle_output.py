"""Pins the oracle's restatement of the output stage (BA::interpOutputData) to the reference binary:
its trajectories, rounded to float32 the way BA::trajWriteBIN writes them, are the bytes of the reference's
traj_out.dat for every golden case the device output stage covers."""
import numpy as np
import pytest

from batotp_amd import capi
import helpers
from helpers import OUTPUT_CASES, Case, run_to_output, assert_output_equals_reference_file, output_params


@pytest.fixture(scope="module")
def octx():
    return capi.Context(helpers.load_oracle(), 0)


@pytest.mark.parametrize("name", OUTPUT_CASES)
def test_oracle_output_matches_reference_traj_out(octx, name):
    case = Case(name)
    out, b = run_to_output(octx, [case])
    assert_output_equals_reference_file(case, out)
    out.close()
    b.close()


def test_cases_cover_both_branches():
    """the fixtures reach the smoothing + down-sampling branch and the re-interpolation branch (out_res < integ_res)"""
    prm = [output_params(n) for n in OUTPUT_CASES]
    assert any(p.out_smooth_fact > 1.5 for p in prm)
    assert any(p.out_res < p.integ_res for p in prm)
    assert any(not (p.out_res < p.integ_res) for p in prm)


def test_unsupported_configuration_is_refused(octx):
    case = Case("CSPR3DOF")  # a cable robot run as a JOINT path: torques could not be recomputed the reference's way
    b = capi.Batch(octx, case.problem, [case.n], case.max_steps())
    b.upload_knots(0, [case.y], [case.sres])
    b.optimize()
    prm = capi.OutputParams(case.problem.n_joints, capi.PATH_JOINT, 0.01, 0.008, 5.0)
    with pytest.raises(capi.BatotpError):
        capi.Output(b, prm, 0, 1)
    b.close()
