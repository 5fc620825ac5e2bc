"""ctypes binding of the C-ABI in include/batotp_hip.h.

The product library is ``batotp_amd/csrc/libbatotp_hip.so`` (HIP kernels for gfx950); ``load_hip`` raises
if it is missing and ``Context`` raises if no GPU is usable -- there is no fallback.  ``Library`` binds any
shared object that exports the same entry points; the test-suite uses that to put its CPU checker behind
the same classes (``tests/helpers.load_oracle``) -- nothing in this package knows where that checker lives.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
HIP_LIB_PATH = os.path.join(_HERE, "csrc", "libbatotp_hip.so")

MAX_JOINTS = 8
MAX_CART = 8

# robot ids (reference batotp/robot.h:33-37)
ROBOT_KUKA, ROBOT_UR, ROBOT_RR, ROBOT_CSPR3DOF, ROBOT_GENJNT = 1, 2, 3, 4, 5
# problem flags
F_JNT_ACC_ON, F_TRQ_ON, F_CART_VEL_ON, F_CART_ACC_ON = 1 << 0, 1 << 1, 1 << 2, 1 << 3
F_PARALLEL, F_PAR2SER, F_HOST_TRIG, F_NO_SAMPLES, F_COMPACT_SPLINES = 1 << 4, 1 << 5, 1 << 6, 1 << 7, 1 << 8
F_CURVES_IN_PLACE = 1 << 9  # the forward curve overwrites the reverse curve (one curve buffer per path)
F_MVC_IN_CURVES = 1 << 10   # the pointwise evaluation writes into the curve buffers (valid until a sweep starts)
F_SVD = 1 << 11             # solveLinSys by Jacobi SVD instead of LU (_isSVD)
# per-path status bits
ST_MAX_INTEG_TIME, ST_CAPACITY, ST_BISECT_FAIL = 1 << 0, 1 << 1, 1 << 2
ST_NONFINITE, ST_SHORT, ST_SEG_ERROR = 1 << 3, 1 << 4, 1 << 5

# path kinds / status bits of the resampler
PATH_JOINT, PATH_CART, PATH_BOTH = 1, 2, 3
RS_TOO_SHORT, RS_IDENTICAL, RS_CAPACITY, RS_SMALL_STEP, RS_SEG_ERROR = 1, 2, 4, 8, 16

_d8 = C.c_double * 8


class Problem(C.Structure):
    """struct batotp_problem"""

    _fields_ = [
        ("n_joints", C.c_int32),
        ("n_cart", C.c_int32),
        ("robot_type", C.c_int32),
        ("flags", C.c_uint32),
        ("jnt_vel_max", _d8),
        ("jnt_acc_max", _d8),
        ("jnt_trq_max", _d8),
        ("jnt_trq_min", _d8),
        ("cart_vel_max", C.c_double),
        ("cart_acc_max", C.c_double),
        ("jnt_thresh", C.c_double),
        ("quad_rad_thresh", C.c_double),
        ("integ_res", C.c_double),
        ("max_integ_time", C.c_double),
        ("pmat", C.c_double * 9),
    ]

    @property
    def dyn_dim(self) -> int:
        if not (self.flags & F_TRQ_ON):
            return 0
        return self.n_cart if (self.flags & F_PARALLEL) else self.n_joints

    @property
    def n_channels(self) -> int:
        return self.n_joints + self.n_cart + 4 * self.dyn_dim


class SerialLink(C.Structure):
    """struct batotp_serial_link"""
    _fields_ = [("axis", C.c_double * 3), ("off", C.c_double * 3), ("com", C.c_double * 3), ("mass", C.c_double),
                ("inertia", C.c_double * 6), ("fv", C.c_double)]


class SerialModel(C.Structure):
    """struct batotp_serial_model"""
    _fields_ = [("n_links", C.c_int32), ("degrees", C.c_int32), ("gravity", C.c_double * 3), ("link", SerialLink * 8)]


class PathResult(C.Structure):
    """struct batotp_path_result"""

    _fields_ = [
        ("t_rev", C.c_double),
        ("t_total", C.c_double),
        ("n_rev", C.c_int64),
        ("n_fwd", C.c_int64),
        ("steps_rev", C.c_int64),
        ("steps_fwd", C.c_int64),
        ("status_rev", C.c_uint32),
        ("status_fwd", C.c_uint32),
        ("n_bisect_fail_rev", C.c_int32),
        ("n_bisect_fail_fwd", C.c_int32),
    ]


RESULT_DTYPE = np.dtype(
    [
        ("t_rev", "<f8"), ("t_total", "<f8"), ("n_rev", "<i8"), ("n_fwd", "<i8"),
        ("steps_rev", "<i8"), ("steps_fwd", "<i8"), ("status_rev", "<u4"), ("status_fwd", "<u4"),
        ("n_bisect_fail_rev", "<i4"), ("n_bisect_fail_fwd", "<i4"),
    ]
)
assert RESULT_DTYPE.itemsize == C.sizeof(PathResult)


def make_problem(n_joints: int, n_cart: int = 0, robot_type: int = ROBOT_GENJNT, flags: int = 0,
                 jnt_vel_max: Sequence[float] = (), jnt_acc_max: Sequence[float] = (),
                 jnt_trq_max: Sequence[float] = (), jnt_trq_min: Sequence[float] = (),
                 cart_vel_max: float = 0.0, cart_acc_max: float = 0.0, jnt_thresh: float = 1e-6,
                 cart_thresh: float = 1e-6, integ_res: float = 0.01, max_integ_time: float = 6000.0,
                 pmat: Optional[Sequence[float]] = None) -> Problem:
    p = Problem()
    p.n_joints, p.n_cart, p.robot_type, p.flags = n_joints, n_cart, robot_type, flags
    for name, vals in (("jnt_vel_max", jnt_vel_max), ("jnt_acc_max", jnt_acc_max),
                       ("jnt_trq_max", jnt_trq_max), ("jnt_trq_min", jnt_trq_min)):
        arr = getattr(p, name)
        for k, v in enumerate(vals):
            arr[k] = float(v)
    p.cart_vel_max, p.cart_acc_max = cart_vel_max, cart_acc_max
    p.jnt_thresh = jnt_thresh
    p.quad_rad_thresh = cart_thresh * cart_thresh  # reference ba.cpp:2048
    p.integ_res, p.max_integ_time = integ_res, max_integ_time
    if pmat is not None:
        for k, v in enumerate(pmat):
            p.pmat[k] = float(v)
    return p


class ResampleParams(C.Structure):
    """struct batotp_resample_params"""
    _fields_ = [
        ("n_joints", C.c_int32), ("n_cart", C.c_int32), ("robot_type", C.c_int32), ("path_type", C.c_int32),
        ("scale_type", C.c_int32), ("flags", C.c_uint32),
        ("s_weights", C.c_double * 3),
        ("theta_norm_res", C.c_double), ("theta_norm_res2", C.c_double),
        ("cart_norm_res", C.c_double), ("cart_norm_res2", C.c_double),
        ("jnt_thresh", C.c_double), ("cart_thresh", C.c_double),
        ("pmat", C.c_double * 9),
        ("input_decim_fact", C.c_int32), ("smooth_window", C.c_int32),
        # automatic integration resolution (flags & RS_AUTO_INTEG_RES): inputs of the rule
        ("jnt_vel_max", C.c_double * MAX_JOINTS), ("jnt_acc_max", C.c_double * MAX_JOINTS),
        ("cart_vel_max", C.c_double), ("cart_acc_max", C.c_double), ("quad_rad_thresh", C.c_double),
        ("degrees", C.c_int32), ("reserved", C.c_int32),
    ]

    @classmethod
    def from_bytes(cls, raw: bytes):
        """a struct stored by an earlier round (shorter: without the inputs of the automatic integration resolution)"""
        raw = bytes(raw)
        return cls.from_buffer_copy(raw + b"\0" * max(0, C.sizeof(cls) - len(raw)))


RS_AUTO_INTEG_RES = 1 << 16


class OutputParams(C.Structure):
    """struct batotp_output_params"""
    _fields_ = [("n_joints", C.c_int32), ("path_type", C.c_int32), ("integ_res", C.c_double), ("out_res", C.c_double),
                ("out_smooth_fact", C.c_double)]


class BatotpError(RuntimeError):
    pass


def _dptr(a: Optional[np.ndarray]):
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Library:
    """One loaded implementation of the C-ABI."""

    def __init__(self, path: str):
        if not os.path.exists(path):
            raise BatotpError(f"{path} is missing -- build it first (python -c 'import __graft_entry__ as g; g.build()')")
        self.path = path
        self.lib = C.CDLL(path)
        L = self.lib
        P, I32, I64, D = C.c_void_p, C.c_int32, C.c_int64, C.POINTER(C.c_double)
        sig = {
            "batotp_hip_device_count": [C.POINTER(C.c_int)],
            "batotp_hip_ctx_create": [C.c_int, C.POINTER(P)],
            "batotp_hip_ctx_destroy": [P],
            "batotp_hip_ctx_trim": [P],
            "batotp_hip_fp64_kat": [P, I64, D, D, D, D, D],
            "batotp_hip_div6_kat": [P, I64, D, D],
            "batotp_hip_spline_lanes_kat": [P, I64, D, D, D, C.POINTER(I32)],
            "batotp_hip_sdiv_kat": [P, I64, D, D, D, C.POINTER(I32)],
            "batotp_hip_batch_create": [P, C.POINTER(Problem), I32, C.POINTER(C.c_int64), I64, C.POINTER(P)],
            "batotp_hip_batch_destroy": [P],
            "batotp_hip_upload_knots": [P, I32, I32, D, D],
            "batotp_hip_upload_knots_device": [P, I32, I32, P, D],
            "batotp_hip_upload_knots_device_rows": [P, I32, I32, P, I32, D],
            "batotp_hip_upload_rr_trig": [P, I32, D],
            "batotp_hip_set_serial_model": [P, C.POINTER(SerialModel)],
            "batotp_hip_upload_joint_trig": [P, I32, D],
            "batotp_hip_builtin_serial_model": [I32, C.POINTER(SerialModel)],
            "batotp_hip_upload_path_sites": [P, I32, D, C.c_double, C.c_double, I32],
            "batotp_hip_upload_coeffs": [P, I32, I32, D],
            "batotp_hip_upload_curve": [P, I32, D, D, I64],
            "batotp_hip_precompute": [P, I32],
            "batotp_hip_pointwise_mvc": [P],
            "batotp_hip_sweep": [P, I32],
            "batotp_hip_optimize": [P],
            "batotp_hip_synchronize": [P],
            "batotp_hip_get_results": [P, C.c_void_p],
            "batotp_hip_download_curve": [P, I32, I32, D, D, I64, C.POINTER(C.c_int64)],
            "batotp_hip_download_coeffs": [P, I32, I32, D],
            "batotp_hip_download_samples": [P, I32, I32, D],
            "batotp_hip_download_dyn": [P, I32, I32, I32, D],
            "batotp_hip_download_mvc": [P, I32, D, D, D],
            "batotp_hip_results_device_ptr": [P, C.POINTER(P), C.POINTER(C.c_int64)],
            "batotp_hip_pack_curves": [P, I32, I32, I32, P, I64, C.POINTER(C.c_int64)],
            "batotp_hip_last_kernel_ms": [P, I32, C.POINTER(C.c_float)],
            "batotp_hip_batch_bytes": [P, C.POINTER(C.c_int64)],
            "batotp_hip_set_sweep_group": [P, I32],
            "batotp_hip_set_paths_per_wave": [P, I32],
            "batotp_hip_set_overlap": [P, I32],
            "batotp_hip_set_sweep_hold": [P, I32, I32],
            "batotp_hip_set_flat_form": [P, I32],
            "batotp_hip_set_sweep_prefetch": [P, I32, I32],
            "batotp_hip_set_spline_tiles": [P, I32],
            "batotp_hip_set_fast_forward": [P, I32],
            "batotp_hip_set_cert_hold": [P, I32],
            "batotp_hip_set_poison": [P, I32],
            "batotp_hip_set_resample_trace": [P, I32],
            "batotp_hip_resampled_trace": [P, C.POINTER(C.c_uint64)],
            "batotp_hip_resampled_trace_data": [P, I32, D, I64, C.POINTER(I64)],
            "batotp_hip_set_k3_form": [P, I32],
            "batotp_hip_set_path_order": [P, I32],
            "batotp_hip_set_workspace_budget": [P, C.c_int64, C.c_int64],
            "batotp_hip_spline_tile_fallbacks": [P, C.POINTER(I32)],
            "batotp_hip_flat_loop_status": [P, C.POINTER(I32)],
            "batotp_hip_toolchain": [C.c_char_p, C.c_char_p, I32],
            "batotp_hip_last_sweep_launch": [P, I32, C.POINTER(I32), C.POINTER(I32), C.POINTER(I32)],
            "batotp_hip_resample": [P, C.POINTER(ResampleParams), I32, C.POINTER(C.c_int64), D, D, C.POINTER(P)],
            "batotp_hip_resampled_destroy": [P],
            "batotp_hip_resampled_auto": [P, D, D, C.POINTER(I32)],
            "batotp_hip_upload_forward_curve": [P, I32, D, D, I64, C.c_double],
            "batotp_hip_set_path_integ_res": [P, I32, I32, D],
            "batotp_hip_resampled_info": [P, C.POINTER(C.c_int64), D, C.POINTER(C.c_uint32)],
            "batotp_hip_resampled_knots_device": [P, C.POINTER(P), C.POINTER(C.c_int64)],
            "batotp_hip_resampled_download": [P, I32, D],
            "batotp_hip_resampled_ms": [P, C.POINTER(C.c_float)],
            "batotp_hip_resampled_checksums": [P, C.POINTER(C.c_uint64)],
            "batotp_hip_output": [P, C.POINTER(OutputParams), I32, I32, C.POINTER(P)],
            "batotp_hip_output_destroy": [P],
            "batotp_hip_out_segmax_kat": [P, I32, C.POINTER(I32), C.POINTER(I32)],
            "batotp_hip_output_info": [P, C.POINTER(C.c_int64), D],
            "batotp_hip_output_channels": [P, C.POINTER(I32), C.POINTER(I32), C.POINTER(I32)],
            "batotp_hip_output_download": [P, I32, D],
            "batotp_hip_output_download_all": [P, D],
            "batotp_hip_output_device": [P, C.POINTER(P), C.POINTER(C.c_int64)],
            "batotp_hip_output_ms": [P, C.POINTER(C.c_float)],
        }
        for name, argtypes in sig.items():
            fn = getattr(L, name)  # raises AttributeError if the symbol is not exported
            fn.argtypes = argtypes
            fn.restype = C.c_int
        L.batotp_hip_last_error.restype = C.c_char_p
        L.batotp_hip_last_error.argtypes = []
        self.symbols = sorted(list(sig) + ["batotp_hip_last_error"])

    def check(self, rc: int, what: str):
        if rc != 0:
            msg = self.lib.batotp_hip_last_error()
            raise BatotpError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")

    def builtin_serial_model(self, robot_type: int) -> SerialModel:
        m = SerialModel()
        self.check(self.lib.batotp_hip_builtin_serial_model(robot_type, C.byref(m)), "builtin_serial_model")
        return m

    def toolchain(self):
        """(built with, flat loop validated with)"""
        a, b = C.create_string_buffer(256), C.create_string_buffer(256)
        self.check(self.lib.batotp_hip_toolchain(a, b, 256), "toolchain")
        return a.value.decode(), b.value.decode()

    def device_count(self) -> int:
        n = C.c_int(0)
        self.lib.batotp_hip_device_count(C.byref(n))
        return n.value


# debug aid for test runs (tests/conftest.py, BATOTP_TEST_POISON=1): every context fills its workspaces with 0xFF bytes before use
DEFAULT_POISON = False


class Context:
    def __init__(self, library: Library, device: int = 0):
        self.library = library
        self.handle = C.c_void_p()
        library.check(library.lib.batotp_hip_ctx_create(device, C.byref(self.handle)), "batotp_hip_ctx_create")
        if DEFAULT_POISON:
            self.set_poison(True)

    def set_resample_trace(self, on: bool):
        """one-path resample calls keep a checksum of every intermediate stage (Resampled.trace)"""
        self.library.check(self.library.lib.batotp_hip_set_resample_trace(self.handle, 1 if on else 0), "set_resample_trace")

    def set_poison(self, on: bool):
        """debug aid: workspaces and batch arrays are filled with 0xFF bytes before use (include/batotp_hip.h)"""
        self.library.check(self.library.lib.batotp_hip_set_poison(self.handle, 1 if on else 0), "set_poison")

    def close(self):
        if self.handle:
            self.library.lib.batotp_hip_ctx_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_sweep_group(self, lanes: int):
        self.library.check(self.library.lib.batotp_hip_set_sweep_group(self.handle, lanes), "set_sweep_group")

    def set_overlap(self, on: bool):
        self.library.check(self.library.lib.batotp_hip_set_overlap(self.handle, 1 if on else 0), "set_overlap")

    def set_sweep_hold(self, reverse: int, forward: int):
        self.library.check(self.library.lib.batotp_hip_set_sweep_hold(self.handle, reverse, forward), "set_sweep_hold")

    def set_flat_form(self, form: int):
        """1 = k_sweep8 (default), 0 = the flat instantiation of the general kernel (include/batotp_hip.h)"""
        self.library.check(self.library.lib.batotp_hip_set_flat_form(self.handle, form), "set_flat_form")

    def set_sweep_prefetch(self, reverse: int, forward: int):
        self.library.check(self.library.lib.batotp_hip_set_sweep_prefetch(self.handle, reverse, forward), "set_sweep_prefetch")

    def set_cert_hold(self, hold: int):
        """k_sweep8, reverse sweep: hold of the certificate phase (-1 automatic, 0 off, 1..8); include/batotp_hip.h"""
        self.library.check(self.library.lib.batotp_hip_set_cert_hold(self.handle, int(hold)), "set_cert_hold")

    def set_fast_forward(self, on):
        """certified fast-forward of the bisection (include/batotp_hip.h); never changes a result"""
        self.library.check(self.library.lib.batotp_hip_set_fast_forward(self.handle, int(on)), "set_fast_forward")

    def set_workspace_budget(self, resample_bytes: int = 0, output_bytes: int = 0):
        """scratch bytes per chunk of paths of the resampler / the output stage (0 = from the free device memory)"""
        self.library.check(self.library.lib.batotp_hip_set_workspace_budget(self.handle, int(resample_bytes), int(output_bytes)), "set_workspace_budget")

    def set_path_order(self, mode: int):
        """ragged batches: 1 = the sweeps take the paths longest first (default), 0 = in the order given"""
        self.library.check(self.library.lib.batotp_hip_set_path_order(self.handle, int(mode)), "set_path_order")

    def set_k3_form(self, form: int):
        """per-knot evaluation of velocity / acceleration-only problems: 1 = k_pointwise_va (default), 0 = the general kernel"""
        self.library.check(self.library.lib.batotp_hip_set_k3_form(self.handle, int(form)), "set_k3_form")

    def set_spline_tiles(self, on):
        """True / False, or -1 for the automatic choice (tiles for small batches)"""
        self.library.check(self.library.lib.batotp_hip_set_spline_tiles(self.handle, -1 if on == -1 else int(on)), "set_spline_tiles")   # (2: the single-pass kernel for pairs)

    def flat_loop_status(self) -> int:
        """1 = the automatic loop choice uses the flat loop; negative: why not (include/batotp_hip.h)"""
        st = C.c_int32(0)
        self.library.check(self.library.lib.batotp_hip_flat_loop_status(self.handle, C.byref(st)), "flat_loop_status")
        return st.value

    def set_paths_per_wave(self, n: int):
        self.library.check(self.library.lib.batotp_hip_set_paths_per_wave(self.handle, n), "set_paths_per_wave")

    def synchronize(self):
        self.library.check(self.library.lib.batotp_hip_synchronize(self.handle), "synchronize")

    def trim(self):
        """release the workspaces cached between resample calls (invalidates live Resampled objects)"""
        self.library.check(self.library.lib.batotp_hip_ctx_trim(self.handle), "ctx_trim")

    def fp64_kat(self, a: np.ndarray, b: np.ndarray):
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        q, r, p = np.empty_like(a), np.empty_like(a), np.empty_like(a)
        self.library.check(self.library.lib.batotp_hip_fp64_kat(self.handle, a.size, _dptr(a), _dptr(b), _dptr(q), _dptr(r), _dptr(p)), "fp64_kat")
        return q, r, p


class Resampled:
    """Uniform-s knots of a set of taught paths, produced by batotp_hip_resample and resident on the
    device (feed them to Batch.upload_knots_device)."""

    def __init__(self, ctx: Context, prm: ResampleParams, x_list: Sequence[np.ndarray], sres_in: Sequence[float]):
        self.ctx, self.lib, self.L = ctx, ctx.library.lib, ctx.library
        self.n_paths = len(x_list)
        n_ch_in = prm.n_joints + prm.n_cart
        # poses (path type BOTH, 6 rows) leave the resampler as position + quaternion: one row more
        self.n_ch = n_ch_in + (1 if (prm.path_type == PATH_BOTH and prm.n_cart == 6) else 0)
        n_in = np.ascontiguousarray([x.shape[1] for x in x_list], dtype=np.int64)
        for x in x_list:
            assert x.shape[0] == n_ch_in
        flat = np.ascontiguousarray(np.concatenate([np.ascontiguousarray(x, dtype=np.float64).ravel() for x in x_list]))
        sr = np.ascontiguousarray(sres_in, dtype=np.float64)
        self.handle = C.c_void_p()
        self.L.check(self.lib.batotp_hip_resample(ctx.handle, C.byref(prm), self.n_paths, n_in.ctypes.data_as(C.POINTER(C.c_int64)),
                                                  _dptr(flat), _dptr(sr), C.byref(self.handle)), "batotp_hip_resample")
        self.n_knots = np.zeros(self.n_paths, dtype=np.int64)
        self.sres = np.zeros(self.n_paths, dtype=np.float64)
        self.status = np.zeros(self.n_paths, dtype=np.uint32)
        self.L.check(self.lib.batotp_hip_resampled_info(self.handle, self.n_knots.ctypes.data_as(C.POINTER(C.c_int64)), _dptr(self.sres),
                                                        self.status.ctypes.data_as(C.POINTER(C.c_uint32))), "resampled_info")

    def auto(self):
        """(integ_res[n_paths], s_weights[n_paths][3], scale_type[n_paths]) the automatic integration resolution left"""
        ir = np.zeros(self.n_paths); sw = np.zeros((self.n_paths, 3)); st = np.zeros(self.n_paths, dtype=np.int32)
        self.L.check(self.lib.batotp_hip_resampled_auto(self.handle, _dptr(ir), _dptr(sw), st.ctypes.data_as(C.POINTER(C.c_int32))), "resampled_auto")
        return ir, sw, st

    def close(self):
        if self.handle:
            self.lib.batotp_hip_resampled_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_ptr(self) -> int:
        ptr, cnt = C.c_void_p(), C.c_int64(0)
        self.L.check(self.lib.batotp_hip_resampled_knots_device(self.handle, C.byref(ptr), C.byref(cnt)), "resampled_knots_device")
        return ptr.value

    def knots(self, path: int) -> np.ndarray:
        y = np.empty((self.n_ch, int(self.n_knots[path])), dtype=np.float64)
        self.L.check(self.lib.batotp_hip_resampled_download(self.handle, path, _dptr(y)), "resampled_download")
        return y

    def ms(self) -> float:
        v = C.c_float(0)
        self.L.check(self.lib.batotp_hip_resampled_ms(self.handle, C.byref(v)), "resampled_ms")
        return float(v.value)

    def trace(self) -> np.ndarray:
        """checksums of the eight intermediate stages of a traced one-path call (include/batotp_hip.h)"""
        out = np.zeros(8, dtype=np.uint64)
        self.L.check(self.lib.batotp_hip_resampled_trace(self.handle, out.ctypes.data_as(C.POINTER(C.c_uint64))), "resampled_trace")
        return out

    def trace_data(self, stage: int) -> np.ndarray:
        """the array of a traced stage that is kept on the host (2, 3); empty for the others"""
        n = C.c_int64(0)
        self.L.check(self.lib.batotp_hip_resampled_trace_data(self.handle, stage, None, 0, C.byref(n)), "resampled_trace_data")
        out = np.empty(int(n.value), dtype=np.float64)
        if n.value:
            self.L.check(self.lib.batotp_hip_resampled_trace_data(self.handle, stage, _dptr(out), n.value, C.byref(n)), "resampled_trace_data")
        return out

    def checksums(self) -> np.ndarray:
        """order-independent 64-bit checksum of every path's knots, computed where the knots are (include/batotp_hip.h)"""
        out = np.zeros(self.n_paths, dtype=np.uint64)
        self.L.check(self.lib.batotp_hip_resampled_checksums(self.handle, out.ctypes.data_as(C.POINTER(C.c_uint64))), "resampled_checksums")
        return out


class Output:
    """Constant-time output trajectories of a range of paths of a batch (batotp_hip_output), resident on the device."""

    def __init__(self, batch: "Batch", prm: OutputParams, path0: int, n_paths: int):
        self.lib, self.L = batch.lib, batch.L
        self.n_paths, self.n_joints = n_paths, prm.n_joints
        self.handle = C.c_void_p()
        self.L.check(self.lib.batotp_hip_output(batch.handle, C.byref(prm), path0, n_paths, C.byref(self.handle)), "batotp_hip_output")
        self.n_pts = np.zeros(n_paths, dtype=np.int64)
        self.sres = np.zeros(n_paths, dtype=np.float64)
        self.L.check(self.lib.batotp_hip_output_info(self.handle, self.n_pts.ctypes.data_as(C.POINTER(C.c_int64)), _dptr(self.sres)), "output_info")
        a, b_, c = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        self.L.check(self.lib.batotp_hip_output_channels(self.handle, C.byref(a), C.byref(b_), C.byref(c)), "output_channels")
        self.n_theta, self.n_cart, self.n_trq = a.value, b_.value, c.value
        self.n_rows = a.value + b_.value + c.value

    def close(self):
        if self.handle:
            self.lib.batotp_hip_output_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def rows(self, k: int) -> np.ndarray:
        """[n_theta + n_cart + n_trq][n_pts] of path k of the range"""
        out = np.empty((self.n_rows, int(self.n_pts[k])), dtype=np.float64)
        if out.size:
            self.L.check(self.lib.batotp_hip_output_download(self.handle, k, _dptr(out)), "output_download")
        return out

    def all_rows(self):
        """every path of the range from one device-to-host copy: list of [n_rows][n_pts] arrays"""
        flat = np.empty(int(self.n_pts.sum()) * self.n_rows, dtype=np.float64)
        if flat.size:
            self.L.check(self.lib.batotp_hip_output_download_all(self.handle, _dptr(flat)), "output_download_all")
        out, at = [], 0
        for n in self.n_pts:
            out.append(flat[at: at + int(n) * self.n_rows].reshape(self.n_rows, int(n)))
            at += int(n) * self.n_rows
        return out

    def theta(self, k: int) -> np.ndarray:
        return self.rows(k)[: self.n_theta]

    def ms(self) -> float:
        v = C.c_float(0)
        self.L.check(self.lib.batotp_hip_output_ms(self.handle, C.byref(v)), "output_ms")
        return float(v.value)


def sdiv_kat(ctx: "Context", a: np.ndarray, b: np.ndarray):
    """(a / b as k_sweep8 divides by a shared reciprocal, which operand pairs went through the reciprocal)"""
    a = np.ascontiguousarray(a, dtype=np.float64); b = np.ascontiguousarray(b, dtype=np.float64)
    q = np.empty_like(a); w = np.empty(a.size, dtype=np.int32)
    ctx.library.check(ctx.library.lib.batotp_hip_sdiv_kat(ctx.handle, a.size, _dptr(a), _dptr(b), _dptr(q),
                                                          w.ctypes.data_as(C.POINTER(C.c_int32))), "sdiv_kat")
    return q, w.astype(bool)


def out_segmax_kat(ctx: "Context", segs):
    """running maxima of the raw segment indices of several paths (findInterpSegs' cursor), through the output stage's launch function"""
    n1 = np.ascontiguousarray([len(x) for x in segs], dtype=np.int32)
    flat = np.ascontiguousarray(np.concatenate([np.asarray(x, dtype=np.int32) for x in segs]), dtype=np.int32)
    ctx.library.check(ctx.library.lib.batotp_hip_out_segmax_kat(ctx.handle, len(segs), n1.ctypes.data_as(C.POINTER(C.c_int32)),
                                                                flat.ctypes.data_as(C.POINTER(C.c_int32))), "out_segmax_kat")
    return np.split(flat, np.cumsum(n1)[:-1])


def spline_lanes_kat(ctx: "Context", y: np.ndarray):
    """second derivatives of one series by the wavefront-per-series solve (+ its fallback) and by the lane-per-series kernel;
    returns (sol, sol_seq, redone)"""
    y = np.ascontiguousarray(y, dtype=np.float64)
    sol, seq = np.empty_like(y), np.empty_like(y)
    redone = C.c_int32(0)
    ctx.library.check(ctx.library.lib.batotp_hip_spline_lanes_kat(ctx.handle, y.size, _dptr(y), _dptr(sol), _dptr(seq), C.byref(redone)),
                      "spline_lanes_kat")
    return sol, seq, int(redone.value)


def div6_kat(ctx: "Context", a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float64)
    q = np.empty_like(a)
    ctx.library.check(ctx.library.lib.batotp_hip_div6_kat(ctx.handle, a.size, _dptr(a), _dptr(q)), "div6_kat")
    return q


class Batch:
    """B independent paths resident on one device."""

    def __init__(self, ctx: Context, prob: Problem, n_knots: Sequence[int], max_steps: int):
        self.ctx, self.lib, self.L = ctx, ctx.library.lib, ctx.library
        self.prob = prob
        self.n_knots = np.ascontiguousarray(n_knots, dtype=np.int64)
        self.n_paths = int(self.n_knots.size)
        self.max_steps = int(max_steps)
        self.handle = C.c_void_p()
        self.L.check(self.lib.batotp_hip_batch_create(ctx.handle, C.byref(prob), self.n_paths,
                                                      self.n_knots.ctypes.data_as(C.POINTER(C.c_int64)), self.max_steps,
                                                      C.byref(self.handle)), "batotp_hip_batch_create")

    def close(self):
        if self.handle:
            self.lib.batotp_hip_batch_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- uploads -----------------------------------------------------------------------------
    def upload_knots(self, path0: int, y_list: Sequence[np.ndarray], sres: Sequence[float]):
        """y_list[k]: [n_joints+n_cart][N_k] knot values of path path0+k."""
        flat = np.ascontiguousarray(np.concatenate([np.ascontiguousarray(y, dtype=np.float64).ravel() for y in y_list]))
        sr = np.ascontiguousarray(sres, dtype=np.float64)
        self.L.check(self.lib.batotp_hip_upload_knots(self.handle, path0, len(y_list), _dptr(flat), _dptr(sr)), "upload_knots")

    def upload_knots_device(self, path0: int, n: int, dev_ptr: int, sres: Sequence[float]):
        sr = np.ascontiguousarray(sres, dtype=np.float64)
        self.L.check(self.lib.batotp_hip_upload_knots_device(self.handle, path0, n, C.c_void_p(dev_ptr), _dptr(sr)), "upload_knots_device")

    def upload_knots_device_rows(self, path0: int, n: int, dev_ptr: int, src_rows: int, sres: Sequence[float]):
        """paths [path0, path0 + n) from a device block whose paths carry src_rows rows each: the first n_joints + n_cart are taken"""
        sr = np.ascontiguousarray(sres, dtype=np.float64)
        self.L.check(self.lib.batotp_hip_upload_knots_device_rows(self.handle, path0, n, C.c_void_p(dev_ptr), int(src_rows), _dptr(sr)),
                     "upload_knots_device_rows")

    def upload_rr_trig(self, path: int, trig: np.ndarray):
        t = np.ascontiguousarray(trig, dtype=np.float64)
        self.L.check(self.lib.batotp_hip_upload_rr_trig(self.handle, path, _dptr(t)), "upload_rr_trig")

    def set_serial_model(self, model: SerialModel):
        self.L.check(self.lib.batotp_hip_set_serial_model(self.handle, C.byref(model)), "set_serial_model")

    def upload_joint_trig(self, path: int, trig: np.ndarray):
        """trig: [2*n_joints][N] cosines, then sines, of the joint angles (radians) at the knot samples"""
        t = np.ascontiguousarray(trig, dtype=np.float64)
        self.L.check(self.lib.batotp_hip_upload_joint_trig(self.handle, path, _dptr(t)), "upload_joint_trig")

    def upload_path_sites(self, path: int, sites: np.ndarray, vfact: float, afact: float, parallel_now: int):
        s = np.ascontiguousarray(sites, dtype=np.float64)
        self.L.check(self.lib.batotp_hip_upload_path_sites(self.handle, path, _dptr(s), vfact, afact, parallel_now), "upload_path_sites")

    def upload_coeffs(self, path: int, channel: int, c: np.ndarray):
        a = np.ascontiguousarray(c, dtype=np.float64)
        self.L.check(self.lib.batotp_hip_upload_coeffs(self.handle, path, channel, _dptr(a)), "upload_coeffs")

    def upload_curve(self, path: int, s: np.ndarray, sdot: np.ndarray):
        a = np.ascontiguousarray(s, dtype=np.float64)
        b = np.ascontiguousarray(sdot, dtype=np.float64)
        self.L.check(self.lib.batotp_hip_upload_curve(self.handle, path, _dptr(a), _dptr(b), a.size), "upload_curve")

    def upload_forward_curve(self, path: int, s: np.ndarray, sdot: np.ndarray, t_total: float):
        a = np.ascontiguousarray(s, dtype=np.float64)
        b = np.ascontiguousarray(sdot, dtype=np.float64)
        self.L.check(self.lib.batotp_hip_upload_forward_curve(self.handle, path, _dptr(a), _dptr(b), a.size, float(t_total)), "upload_forward_curve")

    def set_path_integ_res(self, path0: int, integ_res):
        v = np.ascontiguousarray(integ_res, dtype=np.float64)
        self.L.check(self.lib.batotp_hip_set_path_integ_res(self.handle, path0, v.size, _dptr(v)), "set_path_integ_res")

    # ---- hot path ----------------------------------------------------------------------------
    def precompute(self, stage: int = 0):
        self.L.check(self.lib.batotp_hip_precompute(self.handle, stage), "precompute")

    def pointwise_mvc(self):
        self.L.check(self.lib.batotp_hip_pointwise_mvc(self.handle), "pointwise_mvc")

    def sweep(self, direction: int):
        self.L.check(self.lib.batotp_hip_sweep(self.handle, direction), "sweep")

    def optimize(self):
        self.L.check(self.lib.batotp_hip_optimize(self.handle), "optimize")

    def spline_tile_fallbacks(self) -> int:
        n = C.c_int32(0)
        self.L.check(self.lib.batotp_hip_spline_tile_fallbacks(self.handle, C.byref(n)), "spline_tile_fallbacks")
        return n.value

    def last_sweep_launch(self, direction: int):
        """(lanes per path, paths per wavefront, hold of the flat loop or -1 for the nested loops) of the last launch"""
        lanes, ppw, hold = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        self.L.check(self.lib.batotp_hip_last_sweep_launch(self.handle, direction, C.byref(lanes), C.byref(ppw), C.byref(hold)), "last_sweep_launch")
        return lanes.value, ppw.value, hold.value

    # ---- results -----------------------------------------------------------------------------
    def results(self) -> np.ndarray:
        out = np.zeros(self.n_paths, dtype=RESULT_DTYPE)
        self.L.check(self.lib.batotp_hip_get_results(self.handle, out.ctypes.data_as(C.c_void_p)), "get_results")
        return out

    def curve(self, path: int, which: int):
        n = C.c_int64(0)
        self.L.check(self.lib.batotp_hip_download_curve(self.handle, path, which, None, None, 0, C.byref(n)), "download_curve")
        s = np.empty(max(n.value, 0)); sd = np.empty(max(n.value, 0))
        if n.value > 0:
            self.L.check(self.lib.batotp_hip_download_curve(self.handle, path, which, _dptr(s), _dptr(sd), n.value, C.byref(n)), "download_curve")
        return s, sd

    def coeffs(self, path: int, channel: int) -> np.ndarray:
        out = np.empty((4, int(self.n_knots[path])))
        self.L.check(self.lib.batotp_hip_download_coeffs(self.handle, path, channel, _dptr(out)), "download_coeffs")
        return out

    def samples(self, path: int, channel: int) -> np.ndarray:
        out = np.empty((3, int(self.n_knots[path])))
        self.L.check(self.lib.batotp_hip_download_samples(self.handle, path, channel, _dptr(out)), "download_samples")
        return out

    def dyn(self, path: int, k: int, row: int) -> np.ndarray:
        out = np.empty(int(self.n_knots[path]))
        self.L.check(self.lib.batotp_hip_download_dyn(self.handle, path, k, row, _dptr(out)), "download_dyn")
        return out

    def mvc(self, path: int):
        n = int(self.n_knots[path])
        a, b, c = np.empty(n), np.empty(n), np.empty(n)
        self.L.check(self.lib.batotp_hip_download_mvc(self.handle, path, _dptr(a), _dptr(b), _dptr(c)), "download_mvc")
        return a, b, c

    def pack_curves(self, which: int, path0: int, n_paths: int, dst_ptr: int, dst_points: int) -> int:
        """curves of paths [path0, path0 + n_paths) packed as (s, sdot) pairs into memory of the library's device at dst_ptr;
        returns the number of pairs (call with dst_ptr = 0, dst_points = 0 to size the buffer when no curve is empty ...
        sizes also follow from results()['n_fwd' / 'n_rev'])"""
        total = C.c_int64(0)
        self.L.check(self.lib.batotp_hip_pack_curves(self.handle, which, path0, n_paths, C.c_void_p(dst_ptr), dst_points, C.byref(total)), "pack_curves")
        return int(total.value)

    def results_device_ptr(self):
        ptr, nbytes = C.c_void_p(), C.c_int64(0)
        self.L.check(self.lib.batotp_hip_results_device_ptr(self.handle, C.byref(ptr), C.byref(nbytes)), "results_device_ptr")
        return ptr.value, nbytes.value

    def kernel_ms(self, which: int) -> float:
        ms = C.c_float(0)
        self.L.check(self.lib.batotp_hip_last_kernel_ms(self.handle, which, C.byref(ms)), "last_kernel_ms")
        return float(ms.value)

    def nbytes(self) -> int:
        n = C.c_int64(0)
        self.L.check(self.lib.batotp_hip_batch_bytes(self.handle, C.byref(n)), "batch_bytes")
        return n.value


_hip_library: Optional[Library] = None


def load_hip() -> Library:
    """The product library.  Raises BatotpError when it has not been built."""
    global _hip_library
    if _hip_library is None:
        _hip_library = Library(HIP_LIB_PATH)
    return _hip_library
