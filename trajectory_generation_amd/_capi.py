"""ctypes binding of libtg_mpc.so (include/tg_mpc.h).

The library is the product: there is no CPU fallback.  Importing this module without the
built HIP library raises ImportError with the build command.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TG_MPC_LIB") or os.path.join(_HERE, "lib", "libtg_mpc.so")

TG_INFTY = 1e30
TG_OK, TG_E_ARG, TG_E_HIP, TG_E_STATE, TG_E_NUMERIC = 0, -1, -2, -3, -4
TG_SOLVED, TG_MAX_ITER, TG_PRIMAL_INFEASIBLE, TG_UNSOLVED = 1, -2, -3, -10
TG_SOLVED_INACCURATE, TG_PRIMAL_INFEASIBLE_INACCURATE = 2, 3   # OSQP's verdicts of its tenfold-tolerance test at max_iter
TG_FAM_6DOF, TG_FAM_XY, TG_FAM_Z, TG_FAM_YAW = 0, 1, 2, 3
(TG_MAT_P, TG_MAT_A, TG_MAT_AD, TG_MAT_BD, TG_MAT_Q, TG_MAT_R, TG_MAT_KINV, TG_VEC_L, TG_VEC_U,
 TG_VEC_RHO) = range(10)


class Mpc6Params(C.Structure):
    _fields_ = [("N", C.c_int), ("dt", C.c_double), ("state_weight", C.c_double),
                ("input_weight", C.c_double), ("smooth_input_weight", C.c_double),
                ("enable_control_smoothing", C.c_int), ("min_altitude", C.c_double),
                ("max_vel", C.c_double * 3), ("max_accel", C.c_double * 3)]


class Mpc12Params(C.Structure):
    _fields_ = [("N", C.c_int), ("dt", C.c_double), ("enable_control_smoothing", C.c_int),
                ("xy_state_weight", C.c_double), ("z_state_weight", C.c_double),
                ("yaw_state_weight", C.c_double), ("xy_input_weight", C.c_double),
                ("z_input_weight", C.c_double), ("yaw_input_weight", C.c_double),
                ("xy_smooth_input_weight", C.c_double), ("z_smooth_input_weight", C.c_double),
                ("yaw_smooth_input_weight", C.c_double),
                ("xy_max_vel", C.c_double), ("z_max_vel", C.c_double), ("yaw_max_vel", C.c_double),
                ("xy_max_accel", C.c_double), ("z_max_accel", C.c_double),
                ("yaw_max_accel", C.c_double), ("xy_max_jerk", C.c_double),
                ("z_max_jerk", C.c_double), ("yaw_max_jerk", C.c_double)]


class SolverOpts(C.Structure):
    _fields_ = [("rho", C.c_double), ("sigma", C.c_double), ("alpha", C.c_double),
                ("eps_abs", C.c_double), ("eps_rel", C.c_double), ("max_iter", C.c_int),
                ("check_termination", C.c_int), ("rho_scale", C.c_double * 4),
                ("eps_prim_inf", C.c_double), ("check_fine", C.c_int), ("grid_cap", C.c_int), ("lowrank", C.c_int), ("hard_first", C.c_int),
                ("phase_iters", C.c_int * 2), ("phase_scale", (C.c_double * 4) * 2), ("overlap", C.c_int)]


# every symbol include/tg_mpc.h declares: name -> (restype, argtypes)
_vp, _dp, _ip, _i, _d = C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double
SYMBOLS = {
    "tg_default_opts": (None, [C.POINTER(SolverOpts)]),
    "tg_last_error": (C.c_char_p, []),
    "tg_device_count": (_i, []),
    "tg_mpc6_create": (_i, [C.POINTER(Mpc6Params), C.POINTER(SolverOpts), _i, C.POINTER(_vp)]),
    "tg_mpc6_destroy": (None, [_vp]),
    "tg_mpc6_solve_batch": (_i, [_vp, _i, _i, _dp, _dp, _dp, _dp, _ip, _ip, _i, _vp]),
    "tg_mpc6_solve_host": (_i, [_vp, _i, _dp, _dp, _dp, _dp, _ip, _ip, _i]),
    "tg_mpc6_assemble_batch": (_i, [_vp, _i, _i, _dp, _dp, _dp, _dp, _dp, _vp]),
    "tg_mpc6_assemble_host": (_i, [_vp, _i, _dp, _dp, _dp, _dp, _dp]),
    "tg_mpc6_reset_warm": (_i, [_vp]),
    "tg_mpc6_dims": (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "tg_mpc6_get": (_i, [_vp, _i, _dp]),
    "tg_mpc6_rho": (_d, [_vp]),
    "tg_mpc6_kernel_info": (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "tg_mpc12_create": (_i, [C.POINTER(Mpc12Params), C.POINTER(SolverOpts), _dp, _i, C.POINTER(_vp)]),
    "tg_mpc12_destroy": (None, [_vp]),
    "tg_mpc12_solve_batch": (_i, [_vp, _i, _i, _dp, _dp, _dp, _dp, _ip, _ip, _i, _vp]),
    "tg_mpc12_solve_host": (_i, [_vp, _i, _dp, _dp, _dp, _dp, _ip, _ip, _i]),
    "tg_mpc12_solve_batch_static": (_i, [_vp, _i, _i, _dp, _dp, _dp, _dp, _ip, _ip, _vp]),
    "tg_mpc12_static_scratch_bytes": (C.c_ulonglong, [_i, _i]),
    "tg_mpc12_solve_stage": (_i, [_vp, _i, _i, _i, _dp, _dp, _dp, _dp, _ip, _ip, _vp]),
    "tg_mpc12_intermediates": (_i, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i)]),
    "tg_mpc12_assemble_stage": (_i, [_vp, _i, _i, _i, _dp, _dp, _dp, _dp, _dp, _dp, _vp]),
    "tg_mpc12_assemble_host": (_i, [_vp, _i, _i, _dp, _dp, _dp, _dp, _dp, _dp]),
    "tg_mpc12_reset_warm": (_i, [_vp]),
    "tg_mpc12_dims": (_i, [_vp, _i, C.POINTER(_i), C.POINTER(_i)]),
    "tg_mpc12_get": (_i, [_vp, _i, _i, _dp]),
    "tg_mpc12_rho": (_d, [_vp, _i]),
    "tg_mpc12_kernel_info": (_i, [_vp, _i, C.POINTER(_i), C.POINTER(_i)]),
    "tg_mpc6_plant_step": (_i, [_vp, _i, _i, _dp, _dp, _ip, _vp]),
    "tg_mpc12_plant_step": (_i, [_vp, _i, _i, _dp, _dp, _ip, _vp]),
    "tg_mpc6_get_opts": (_i, [_vp, C.POINTER(SolverOpts)]),
    "tg_mpc6_set_opts": (_i, [_vp, C.POINTER(SolverOpts)]),
    "tg_mpc6_last_residuals": (_i, [_vp, _i, _dp, _dp]),
    "tg_mpc12_get_opts": (_i, [_vp, _i, C.POINTER(SolverOpts)]),
    "tg_mpc12_set_opts": (_i, [_vp, C.POINTER(SolverOpts)]),
    "tg_mpc12_last_residuals": (_i, [_vp, _i, _i, _dp, _dp]),
    "tg_ref_shift": (_i, [_i, _i, _i, _i, _dp, _vp]),
    "tg_copy_to_host": (_i, [_vp, _vp, C.c_ulonglong]),
    "tg_timing_enable": (None, [_i]),
    "tg_timing_read": (_i, [C.POINTER(_d), C.POINTER(_i), _i]),
}

_lib = None


def load():
    """Load libtg_mpc.so and declare its prototypes.  No fallback: raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "trajectory_generation_amd: %s is missing; build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C trajectory_generation_amd/csrc` (needs hipcc, target gfx950)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class TgError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("tg_mpc error %d: %s" % (code, msg))
        self.code = code


def check(rc):
    if rc != TG_OK:
        raise TgError(rc, load().tg_last_error().decode("utf-8", "replace"))
