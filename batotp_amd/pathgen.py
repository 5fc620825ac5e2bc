"""Synthetic input paths for tests, golden fixtures and the benchmark (host side, numpy).

The generators restate the recipes of the reference's Octave scripts with an own counter-based
RNG (the reference uses rand()):
  * gen7dof_*  : reference input/GEN7DOF/generateGEN7DOFpath.m:1-19 (waypoints 5*U[0,1)^7,
                 not-a-knot cubic up-sampling x20, float32 binary file, tres 0.01)
  * cspr_*     : reference input/CSPR3DOF/generatePathPointsCSPR.m:5-33
  * ur_like_*  : a 6-joint path in degrees around a UR5 home pose (cfg 2)
  * kuka_like_*: a 7-joint path in degrees for the KUKA LWR IV+ with torque limits (cfg 3)
File writers follow the formats of SURVEY.md Appendix A (reference ba.cpp:2257-2312 reader,
ba.cpp:1942-2087 config parser).
"""
from __future__ import annotations

import struct
from typing import Optional, Sequence

import numpy as np

_MASK = (1 << 64) - 1


def splitmix64_uniform(seed: int, n: int) -> np.ndarray:
    """n doubles in [0,1) from the splitmix64 stream started at `seed` (counter-based: element i depends on seed and i only,
    so the whole stream is one vectorised expression in 64-bit wrap-around arithmetic)."""
    golden = np.uint64(0x9E3779B97F4A7C15)
    with np.errstate(over="ignore"):
        z = np.uint64(seed & _MASK) + golden * np.arange(1, n + 1, dtype=np.uint64)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _upsample(way: np.ndarray, factor: int) -> np.ndarray:
    """way [n][ch] -> [ch][n*factor] through a not-a-knot cubic spline (MATLAB interp1 'spline')."""
    from scipy.interpolate import CubicSpline

    n = way.shape[0]
    cs = CubicSpline(np.arange(n, dtype=np.float64), way, axis=0)
    s_out = np.linspace(0.0, n - 1.0, n * factor)
    return np.ascontiguousarray(cs(s_out).T)


def gen7dof_fine(seed: int, n_coarse: int, factor: int = 20, n_joints: int = 7, scale: float = 5.0) -> np.ndarray:
    u = splitmix64_uniform(seed, n_coarse * n_joints).reshape(n_coarse, n_joints)
    return _upsample(scale * u, factor).astype(np.float32)


def ur_like_fine(seed: int, n_coarse: int, factor: int = 20) -> np.ndarray:
    centre = np.array([-47.0, -116.0, -80.0, -72.0, 87.0, -93.0])
    u = splitmix64_uniform(seed, n_coarse * 6).reshape(n_coarse, 6)
    return _upsample(centre + 60.0 * (u - 0.5), factor).astype(np.float32)


def kuka_like_fine(seed: int, n_coarse: int, factor: int = 20) -> np.ndarray:
    """7 joints in degrees around a bent-elbow pose of a 7-DOF arm (BASELINE config 3: the GEN7DOF recipe in degrees,
    SURVEY.md 8d), inside the LWR IV+ joint ranges"""
    centre = np.array([0.0, 30.0, 0.0, -60.0, 0.0, 45.0, 0.0])
    u = splitmix64_uniform(seed, n_coarse * 7).reshape(n_coarse, 7)
    return _upsample(centre + 60.0 * (u - 0.5), factor).astype(np.float32)


def cspr_fine(seed: int, n_coarse: int, sres: float = 0.005, amp: float = 3.0) -> np.ndarray:
    from scipy.interpolate import CubicSpline

    u = splitmix64_uniform(seed, n_coarse * 3).reshape(n_coarse, 3)
    way = np.stack([amp * (u[:, 0] - 0.5), amp * (u[:, 1] - 0.35), amp * (u[:, 2] + 0.75)], axis=1)
    cs = CubicSpline(np.arange(n_coarse, dtype=np.float64), way, axis=0)
    ss = np.arange(0.0, n_coarse - 1.0 + 1e-12, sres)
    return np.ascontiguousarray(cs(ss).T).astype(np.float32)


def write_traj_bin(path: str, tres: float, theta: Optional[np.ndarray], cart: Optional[np.ndarray]) -> None:
    """float32 tres; int32 nPts; int32 hasTheta; [theta f32 channel-major]; int32 hasCart; [cart]."""
    n = (theta if theta is not None else cart).shape[1]
    with open(path, "wb") as f:
        f.write(struct.pack("<f", tres))
        f.write(struct.pack("<i", n))
        f.write(struct.pack("<i", 1 if theta is not None else 0))
        if theta is not None:
            f.write(np.ascontiguousarray(theta, dtype="<f4").tobytes())
        f.write(struct.pack("<i", 1 if cart is not None else 0))
        if cart is not None:
            f.write(np.ascontiguousarray(cart, dtype="<f4").tobytes())


def read_traj_bin(path: str, n_joints: int, n_cart: int):
    """inverse of write_traj_bin: (tres, theta or None, cart or None), values widened to float64"""
    raw = open(path, "rb").read()
    tres = float(np.frombuffer(raw, "<f4", 1, 0)[0])
    n = int(np.frombuffer(raw, "<i4", 1, 4)[0])
    pos = 8
    out = []
    for rows in (n_joints, n_cart):
        has = int(np.frombuffer(raw, "<i4", 1, pos)[0])
        pos += 4
        if has == 1:
            out.append(np.frombuffer(raw, "<f4", rows * n, pos).reshape(rows, n).astype(np.float64))
            pos += 4 * rows * n
        else:
            out.append(None)
    return tres, out[0], out[1]


def _vec(v: Sequence[float]) -> str:
    return " ".join(repr(float(x)) if not (isinstance(x, float) and np.isnan(x)) else "NAN" for x in v)


def write_config(path: str, *, robot: str, is_parallel: int, n_joints: int, n_cart: int, traj_file: str,
                 is_bin: int, path_type: str, degrees: int, jnt_vel: Sequence[float], jnt_acc_on: int,
                 jnt_acc: Sequence[float], trq_on: int = 0, trq_max: Optional[Sequence[float]] = None,
                 trq_min: Optional[Sequence[float]] = None, cart_vel_on: int = 0, cart_vel: float = 0.0,
                 cart_acc_on: int = 0, cart_acc: float = 0.0, integ_res: float = 0.01,
                 max_integ_time: float = 20000.0, decim: int = 1, smooth: int = 1, sdot_out: int = 1,
                 jnt_thresh: float = 1e-6, cart_thresh: float = 1e-6, s_weights=(0, 1, 0), scale_type: int = 1,
                 theta_res: float = 0.1, theta_res2: float = 0.1, cart_res: float = 0.02, cart_res2: float = 0.02,
                 out_res: float = 0.008, out_smooth: int = 1, is_svd: int = 0, par2ser: int = 0) -> None:
    """config.dat in the positional format parsed by reference ba.cpp:1942-2087."""
    z = [0.0] * n_joints
    trq_max = z if trq_max is None else trq_max
    trq_min = z if trq_min is None else trq_min
    lines = [
        "// synthetic configuration written by batotp_amd.pathgen.write_config",
        "// same positional layout as the reference's input/*/config.dat",
        "",
        f"{robot} // robotTypeStr", f"{is_parallel} // isParallel", f"{n_joints} // _nJoints", f"{n_cart} // nCart",
        f"{traj_file} // trajFileName", f"{is_bin} // isBINfile", f"{path_type} // pathType",
        "", "// CONSTRAINTS",
        f"{degrees} // areJntAnglesDegrees", "1 // isJntVelConOn", f"{_vec(jnt_vel)} // JntVelLims",
        f"{jnt_acc_on} // isJntAccConOn", f"{_vec(jnt_acc)} // JntAccLims", f"{trq_on} // isTrqConOn",
        f"{_vec(trq_max)} // JntTrqMax", f"{_vec(trq_min)} // JntTrqMin", f"{cart_vel_on} // isCartVelConOn",
        f"{cart_vel!r} // CartVelMax", f"{cart_acc_on} // isCartAccConOn", f"{cart_acc!r} // CartAccMax",
        "", "// INTEGRATION PARAMETERS",
        f"{integ_res!r} // integRes", f"{max_integ_time!r} // maxIntegTime",
        "", "// OTHER CONTROLS",
        f"{decim} // inputDecimFact", f"{smooth} // smoothWindow", f"{sdot_out} // is_sdotOut",
        f"{jnt_thresh!r} // jntThresh", f"{cart_thresh!r} // cartThresh", f"{_vec(s_weights)} // sWeights",
        f"{scale_type} // scaleType", f"{theta_res!r} // thetaNormRes", f"{theta_res2!r} // thetaNormRes2",
        f"{cart_res!r} // cartNormRes", f"{cart_res2!r} // cartNormRes2", f"{out_res!r} // outRes",
        f"{out_smooth} // outSmoothFact", f"{is_svd} // isSVD", f"{par2ser} // isPar2Ser", "", "",
    ]
    with open(path, "w") as f:
        f.write("\n".join(lines))


def read_s_sdot(path: str):
    """s-sdot.dat (reference ba.cpp:2726-2759): twice {f64 sres; i32 n; f32 s[n]; f32 sdot[n]}."""
    b = open(path, "rb").read()
    o, out = 0, []
    for _ in range(2):
        sres = struct.unpack_from("<d", b, o)[0]; o += 8
        n = struct.unpack_from("<i", b, o)[0]; o += 4
        s = np.frombuffer(b, "<f4", n, o).copy(); o += 4 * n
        sd = np.frombuffer(b, "<f4", n, o).copy(); o += 4 * n
        out.append((sres, s, sd))
    return out
