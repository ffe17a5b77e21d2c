"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (see oracle/mpc_oracle.h).  Imported by
tests/, by __graft_entry__.smoke() and by the cpu_baseline leg of bench.py; never by
the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

INFTY = 1e30
FAM_6DOF, FAM_XY, FAM_Z, FAM_YAW = 0, 1, 2, 3
SOLVED, MAX_ITER, PRIMAL_INFEASIBLE, DUAL_INFEASIBLE, UNSOLVED = 1, -2, -3, -4, -10


def build(force=False):
    """Compile oracle/liboracle.so with gcc (CPU only)."""
    srcs = [os.path.join(_HERE, f) for f in ("osqp_restate.c", "mpc_qp.c", "mpc_oracle.h")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def use_native():
    """For timing the CPU baseline on the machine it runs on: the same sources built with
    -march=native into oracle/_native/ (the portable liboracle.so has to run on whatever machine the
    tree travels to), loaded in place of it.  Call before anything else of this module is used.
    Returns the flags used, or None when the build failed and the portable library stays."""
    global _LIB_PATH, _lib
    flags = ["-O3", "-march=native", "-fPIC", "-std=gnu11", "-fno-fast-math", "-ffp-contract=off"]
    out_dir = os.path.join(_HERE, "_native")
    out = os.path.join(out_dir, "liboracle_native.so")
    try:
        os.makedirs(out_dir, exist_ok=True)
        subprocess.check_call([os.environ.get("CC", "gcc")] + flags + ["-shared", "-o", out, os.path.join(_HERE, "osqp_restate.c"),
                               os.path.join(_HERE, "mpc_qp.c"), "-lm"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except (OSError, subprocess.CalledProcessError):
        return None
    _LIB_PATH, _lib = out, None
    return " ".join(flags)


class Settings(C.Structure):
    _fields_ = [("rho", C.c_double), ("sigma", C.c_double), ("alpha", C.c_double),
                ("eps_abs", C.c_double), ("eps_rel", C.c_double),
                ("eps_prim_inf", C.c_double), ("eps_dual_inf", C.c_double),
                ("max_iter", C.c_int), ("check_termination", C.c_int), ("scaling", C.c_int),
                ("adaptive_rho", C.c_int), ("adaptive_rho_interval", C.c_int),
                ("adaptive_rho_tolerance", C.c_double), ("warm_start", C.c_int),
                ("scaled_termination", C.c_int), ("check_fine", C.c_int), ("linsys", C.c_int)]


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


_lib = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _pi(a):
    return None if a is None else a.ctypes.data_as(_ip)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.osqp_setup.restype = C.c_void_p
        L.osqp_setup.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, C.POINTER(Settings)]
        L.osqp_cleanup.argtypes = [C.c_void_p]
        L.osqp_update_bounds.argtypes = [C.c_void_p, _dp, _dp]
        L.osqp_update_lin_cost.argtypes = [C.c_void_p, _dp]
        L.osqp_warm_start.argtypes = [C.c_void_p, _dp, _dp]
        L.osqp_solve.argtypes = [C.c_void_p]
        L.osqp_get_solution.argtypes = [C.c_void_p, _dp, _dp]
        L.osqp_get_iter.argtypes = [C.c_void_p]
        L.osqp_get_rho_updates.argtypes = [C.c_void_p]
        L.osqp_get_rho.argtypes = [C.c_void_p]
        L.osqp_get_rho.restype = C.c_double
        L.osqp_get_residuals.argtypes = [C.c_void_p, _dp, _dp]
        L.osqp_get_scaling.argtypes = [C.c_void_p, _dp, _dp, _dp]
        L.osqp_get_rho_vec.argtypes = [C.c_void_p, _dp]
        L.qp_dims.argtypes = [C.c_int, C.c_int, _ip, _ip, _ip, _ip]
        L.qp6_build.argtypes = [C.POINTER(Mpc6Params), _dp, _dp, _dp, _dp, _dp]
        L.qp12_build.argtypes = [C.POINTER(Mpc12Params), C.c_int, _dp, _dp, _dp, _dp, _dp]
        L.model6.argtypes = [C.POINTER(Mpc6Params), _dp, _dp, _dp, _dp]
        L.model12.argtypes = [C.POINTER(Mpc12Params), C.c_int, _dp, _dp, _dp, _dp]
        L.qp6_update.argtypes = [C.POINTER(Mpc6Params), _dp, _dp, _dp, _dp, _dp]
        L.qp12_update_z.argtypes = [C.POINTER(Mpc12Params), _dp, _dp, _dp, _dp, _dp]
        L.qp12_update_yaw.argtypes = [C.POINTER(Mpc12Params), _dp, _dp, _dp, _dp, _dp]
        L.qp12_xy_bounds_from_z.argtypes = [C.POINTER(Mpc12Params), _dp, _dp, _dp]
        L.qp12_update_xy.argtypes = [C.POINTER(Mpc12Params), _dp, _dp, _dp, _dp, _dp, _dp, _dp]
        L.qp12_yaw_ref_from_xy.argtypes = [C.c_int, _dp, _dp, _dp]
        L.qp12_split_state.argtypes = [_dp, _dp, _dp, _dp]
        L.qp12_split_ref.argtypes = [C.c_int, _dp, _dp, _dp, _dp]
        L.qp12_merge.argtypes = [C.c_int, _dp, _dp, _dp, _dp, _dp]
        L.mpc6_create.restype = C.c_void_p
        L.mpc6_create.argtypes = [C.POINTER(Mpc6Params), C.POINTER(Settings)]
        L.mpc6_destroy.argtypes = [C.c_void_p]
        L.mpc6_cold_start.argtypes = [C.c_void_p]
        L.mpc6_loop.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _ip, _ip]
        L.mpc12_create.restype = C.c_void_p
        L.mpc12_create.argtypes = [C.POINTER(Mpc12Params), C.POINTER(Settings)]
        L.mpc12_destroy.argtypes = [C.c_void_p]
        L.mpc12_loop.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _ip, _ip, _dp]
        L.mpc6_batch.argtypes = [C.POINTER(Mpc6Params), C.POINTER(Settings), C.c_int, _dp, _dp,
                                 _dp, _dp, _ip, _ip]
        L.mpc12_batch.argtypes = [C.POINTER(Mpc12Params), C.POINTER(Settings), C.c_int, _dp, _dp,
                                  _dp, _dp, _ip, _ip]
        L.mpc12_create3.restype = C.c_void_p
        L.mpc12_create3.argtypes = [C.POINTER(Mpc12Params)] + [C.POINTER(Settings)] * 3
        L.mpc12_batch3.argtypes = [C.POINTER(Mpc12Params)] + [C.POINTER(Settings)] * 3 + [C.c_int, _dp, _dp, _dp, _dp, _ip, _ip]
        L.osqp_set_rho_scale.argtypes = [C.c_void_p, _dp]
        L.mpc_set_rho_class_scale.argtypes = [C.c_double] * 4
        L.mpc_set_rho_class_scale_family.argtypes = [C.c_int] + [C.c_double] * 4
        L.mpc_set_rho_phases_family.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)]
        L.synth_uniform.restype = C.c_double
        L.synth_uniform.argtypes = [C.c_ulonglong, C.c_ulonglong, C.c_ulonglong]
        L.synth_mpc6.argtypes = [C.c_ulonglong, C.c_int, C.c_double, C.c_int, C.c_int, _dp, _dp]
        L.synth_mpc12.argtypes = [C.c_ulonglong, C.c_int, C.c_double, C.c_int, C.c_int, _dp, _dp]
        _lib = L
    return _lib


# ------------------------------------------------------------------ settings / params
def default_settings(**kw):
    s = Settings()
    lib().osqp_default_settings(C.byref(s))
    for k, v in kw.items():
        setattr(s, k, v)
    return s


def fixed_rho_settings(rho, **kw):
    s = Settings()
    lib().osqp_fixed_rho_settings(C.byref(s), C.c_double(rho))
    for k, v in kw.items():
        setattr(s, k, v)
    return s


def mpc6_params(N=20, dt=0.1, state_weight=7000.0, input_weight=500.0, smooth_input_weight=10.0,
                enable_control_smoothing=False, min_altitude=-1.0, max_vel=(10.0, 10.0, 10.0),
                max_accel=(5.0, 5.0, 5.0)):
    p = Mpc6Params()
    p.N, p.dt = N, dt
    p.state_weight, p.input_weight, p.smooth_input_weight = state_weight, input_weight, smooth_input_weight
    p.enable_control_smoothing = int(enable_control_smoothing)
    p.min_altitude = min_altitude
    p.max_vel[:] = max_vel
    p.max_accel[:] = max_accel
    return p


def mpc12_params(N=20, dt=0.1, enable_control_smoothing=False,
                 state_weight=(5000.0, 7000.0, 1000.0), input_weight=(200.0, 200.0, 100.0),
                 smooth_input_weight=(100.0, 100.0, 100.0), max_vel=(12.0, 7.0, 2.0),
                 max_accel=(6.0, 10.0, 10.0), max_jerk=(30.0, 40.0, 10.0)):
    """Triples are (xy, z, yaw); defaults are config/mpc_12state.yaml:23-53."""
    p = Mpc12Params()
    p.N, p.dt, p.enable_control_smoothing = N, dt, int(enable_control_smoothing)
    p.xy_state_weight, p.z_state_weight, p.yaw_state_weight = state_weight
    p.xy_input_weight, p.z_input_weight, p.yaw_input_weight = input_weight
    p.xy_smooth_input_weight, p.z_smooth_input_weight, p.yaw_smooth_input_weight = smooth_input_weight
    p.xy_max_vel, p.z_max_vel, p.yaw_max_vel = max_vel
    p.xy_max_accel, p.z_max_accel, p.yaw_max_accel = max_accel
    p.xy_max_jerk, p.z_max_jerk, p.yaw_max_jerk = max_jerk
    return p


def set_rho_class_scale(pos=1.0, vel=1.0, acc=1.0, inp=1.0):
    """Per-class multipliers of rho used by Mpc6 / Mpc12 / mpc*_batch objects created afterwards.
    Also switches the step-size schedule off (match_engine sets both)."""
    lib().mpc_set_rho_class_scale(pos, vel, acc, inp)
    set_rho_phases(None)


def match_rho_classes(stages):
    """Read the per-class multipliers back from per-row rho vectors and make them the ones the
    oracle uses.  stages: iterable of (rho_vec, rho, nx, N[, stage]) in the reference's row ordering
    (dynamics | state box | input box | mixed); stage = "xy" / "z" / "yaw" names the family of a
    12-state stage (each keeps its own set), without it the set applies to every family.  Rows with
    infinite bounds carry rho_min and say nothing about their class; such a class keeps multiplier 1."""
    fams = {"xy": FAM_XY, "z": FAM_Z, "yaw": FAM_YAW}
    cls = [1.0, 1.0, 1.0, 1.0]
    for st in stages:
        rho_vec, rho, nx, N = st[:4]
        cls = [1.0, 1.0, 1.0, 1.0]
        Nx = nx * (N + 1)
        rows = {0: Nx + 0, 3: 2 * Nx}
        if nx == 6 and len(rho_vec) == 2 * Nx + 3 * N:      # 6-DOF: p(3), v(3); only p_z is bounded
            rows[0], rows[1] = Nx + 2, Nx + 3
        else:                                                 # chains p, v, a
            rows[1], rows[2] = Nx + 1, Nx + 2
        for k, r in rows.items():
            f = rho_vec[r] / rho
            if f > 1e-3:
                cls[k] = float(f)
        if len(st) > 4:
            lib().mpc_set_rho_class_scale_family(fams[st[4]], *cls)
        else:
            set_rho_class_scale(*cls)
    return tuple(cls)


def set_rho_phases(family=None, starts=(), classes=()):
    """Step-size schedule of the solvers created afterwards: phases 1.. start at iterations `starts`
    with absolute class multipliers `classes` (one 4-tuple each); family None = every family.
    No starts: schedule off."""
    n = 1 + len(starts)
    st = (C.c_int * 3)(0, *[int(v) for v in starts])
    cl = (C.c_double * 12)(*([0.0] * 4 + [float(v) for c in classes for v in c]))
    for f in ((FAM_6DOF, FAM_XY, FAM_Z, FAM_YAW) if family is None else (family,)):
        lib().mpc_set_rho_phases_family(f, n, st, cl)


def match_engine(eng):
    """Make the oracle's fixed-rho mode run the step sizes of an engine handle (BatchedMPC6 /
    BatchedMPC12): the class multipliers of the first operator and the schedule of the later ones
    (tg_solver_opts::phase_iters / phase_scale as the handle resolved them)."""
    cls = match_rho_classes(eng.rho_rows())
    fams = {"xy": FAM_XY, "z": FAM_Z, "yaw": FAM_YAW}
    stages = [None] if not hasattr(eng, "STAGES") else ["xy", "z", "yaw"]
    for st in stages:
        o = eng.get_opts() if st is None else eng.get_opts(st)
        base = [float(v) for v in o.rho_scale]
        starts = [int(o.phase_iters[p]) for p in range(2) if o.phase_iters[p] > 0]
        classes = [[base[k] * float(o.phase_scale[p][k]) for k in range(4)] for p in range(len(starts))]
        # the engine moves an instance at the first check at or after the iteration asked for
        chk = int(o.check_termination)
        starts = [(v + chk - 1) // chk * chk for v in starts]
        set_rho_phases(None if st is None else fams[st], starts, classes)
    return cls


def dims(family, N):
    n, m, nx, nu = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    lib().qp_dims(family, N, C.byref(n), C.byref(m), C.byref(nx), C.byref(nu))
    return n.value, m.value, nx.value, nu.value


# ------------------------------------------------------------------ QP construction
def build_qp(params, family=FAM_6DOF):
    """Returns dict(P, A, q, l, u, n, m) for the init-time problem (x0 = 0, ref = 0)."""
    n, m, nx, nu = dims(family, params.N)
    P = np.zeros((n, n)); A = np.zeros((m, n)); q = np.zeros(n); l = np.zeros(m); u = np.zeros(m)
    if family == FAM_6DOF:
        lib().qp6_build(C.byref(params), _p(P), _p(A), _p(q), _p(l), _p(u))
    else:
        lib().qp12_build(C.byref(params), family, _p(P), _p(A), _p(q), _p(l), _p(u))
    return dict(P=P, A=A, q=q, l=l, u=u, n=n, m=m, nx=nx, nu=nu, N=params.N)


def model(params, family=FAM_6DOF):
    _, _, nx, nu = dims(family, params.N)
    Ad = np.zeros((nx, nx)); Bd = np.zeros((nx, nu)); Q = np.zeros((nx, nx)); R = np.zeros((nu, nu))
    if family == FAM_6DOF:
        lib().model6(C.byref(params), _p(Ad), _p(Bd), _p(Q), _p(R))
    else:
        lib().model12(C.byref(params), family, _p(Ad), _p(Bd), _p(Q), _p(R))
    return Ad, Bd, Q, R


def qp6_vectors(params, x0, ref):
    n, m, _, _ = dims(FAM_6DOF, params.N)
    x0 = np.ascontiguousarray(x0, dtype=np.float64); ref = np.ascontiguousarray(ref, dtype=np.float64)
    q = np.zeros(n); l = np.zeros(m); u = np.zeros(m)
    lib().qp6_update(C.byref(params), _p(x0), _p(ref), _p(q), _p(l), _p(u))
    return q, l, u


def qp12_vectors(params, family, s0, ref, xy_max=None, mixed_max=None):
    n, m, _, _ = dims(family, params.N)
    s0 = np.ascontiguousarray(s0, dtype=np.float64); ref = np.ascontiguousarray(ref, dtype=np.float64)
    q = np.zeros(n); l = np.zeros(m); u = np.zeros(m)
    if family == FAM_Z:
        lib().qp12_update_z(C.byref(params), _p(s0), _p(ref), _p(q), _p(l), _p(u))
    elif family == FAM_YAW:
        lib().qp12_update_yaw(C.byref(params), _p(s0), _p(ref), _p(q), _p(l), _p(u))
    else:
        xy_max = np.ascontiguousarray(xy_max, dtype=np.float64)
        mixed_max = np.ascontiguousarray(mixed_max, dtype=np.float64)
        lib().qp12_update_xy(C.byref(params), _p(s0), _p(ref), _p(xy_max), _p(mixed_max), _p(q),
                             _p(l), _p(u))
    return q, l, u


def xy_bounds_from_z(params, z_x_opt):
    z_x_opt = np.ascontiguousarray(z_x_opt, dtype=np.float64)
    xm = np.zeros(6 * params.N); mm = np.zeros(6 * params.N)
    lib().qp12_xy_bounds_from_z(C.byref(params), _p(z_x_opt), _p(xm), _p(mm))
    return xm, mm


def yaw_ref_from_xy(N, xyref, xy_x_opt):
    xyref = np.ascontiguousarray(xyref, dtype=np.float64)
    xy_x_opt = np.ascontiguousarray(xy_x_opt, dtype=np.float64)
    out = np.zeros(3 * (N + 1))
    lib().qp12_yaw_ref_from_xy(N, _p(xyref), _p(xy_x_opt), _p(out))
    return out


def split_ref12(N, ref12):
    ref12 = np.ascontiguousarray(ref12, dtype=np.float64)
    a = np.zeros(6 * (N + 1)); b = np.zeros(3 * (N + 1)); c = np.zeros(3 * (N + 1))
    lib().qp12_split_ref(N, _p(ref12), _p(a), _p(b), _p(c))
    return a, b, c


# ------------------------------------------------------------------ generic OSQP restatement
class Osqp:
    def __init__(self, P, A, q, l, u, settings):
        self.n, self.m = P.shape[0], A.shape[0]
        P = np.ascontiguousarray(P, dtype=np.float64); A = np.ascontiguousarray(A, dtype=np.float64)
        self._h = lib().osqp_setup(self.n, self.m, _p(P), _p(A), _p(np.ascontiguousarray(q)),
                                   _p(np.ascontiguousarray(l)), _p(np.ascontiguousarray(u)),
                                   C.byref(settings))
        if not self._h:
            raise RuntimeError("osqp_setup failed (reduced KKT matrix not positive definite)")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().osqp_cleanup(self._h)
            self._h = None

    def update(self, q=None, l=None, u=None):
        if l is not None:
            rc = lib().osqp_update_bounds(self._h, _p(np.ascontiguousarray(l)), _p(np.ascontiguousarray(u)))
            if rc:
                raise ValueError("update_bounds rc=%d" % rc)
        if q is not None:
            lib().osqp_update_lin_cost(self._h, _p(np.ascontiguousarray(q)))

    def set_rho_scale(self, scale):
        rc = lib().osqp_set_rho_scale(self._h, _p(np.ascontiguousarray(scale, dtype=np.float64)))
        if rc:
            raise RuntimeError("refactorisation failed")

    def warm_start(self, x, y):
        lib().osqp_warm_start(self._h, _p(np.ascontiguousarray(x)), _p(np.ascontiguousarray(y)))

    def solve(self):
        st = lib().osqp_solve(self._h)
        x = np.zeros(self.n); y = np.zeros(self.m)
        lib().osqp_get_solution(self._h, _p(x), _p(y))
        pri, dua = C.c_double(), C.c_double()
        lib().osqp_get_residuals(self._h, C.byref(pri), C.byref(dua))
        return dict(status=st, x=x, y=y, iter=lib().osqp_get_iter(self._h), pri_res=pri.value,
                    dua_res=dua.value, rho_updates=lib().osqp_get_rho_updates(self._h),
                    rho=lib().osqp_get_rho(self._h))

    def scaling(self):
        D = np.zeros(self.n); E = np.zeros(self.m); c = C.c_double()
        lib().osqp_get_scaling(self._h, _p(D), _p(E), C.byref(c))
        return D, E, c.value

    def rho_vec(self):
        r = np.zeros(self.m)
        lib().osqp_get_rho_vec(self._h, _p(r))
        return r


# ------------------------------------------------------------------ whole mpcLoop
def mpc6_batch(params, settings, x0, ref):
    """x0: [B,6], ref: [B,6(N+1)] (instance-major).  Every instance starts cold."""
    x0 = np.ascontiguousarray(x0, dtype=np.float64); ref = np.ascontiguousarray(ref, dtype=np.float64)
    B, N = x0.shape[0], params.N
    X = np.zeros((B, 6 * (N + 1))); U = np.zeros((B, 3 * N))
    it = np.zeros(B, dtype=np.int32); st = np.zeros(B, dtype=np.int32)
    ok = lib().mpc6_batch(C.byref(params), C.byref(settings), B, _p(x0), _p(ref), _p(X), _p(U),
                          _pi(it), _pi(st))
    return dict(X=X, U=U, iters=it, status=st, n_ok=ok)


def _three(settings):
    """settings: one Settings for all stages, or dict(xy=, z=, yaw=)."""
    if isinstance(settings, dict):
        return settings["xy"], settings["z"], settings["yaw"]
    return settings, settings, settings


def mpc12_batch(params, settings, x12, ref12):
    sxy, sz, syaw = _three(settings)
    x12 = np.ascontiguousarray(x12, dtype=np.float64); ref12 = np.ascontiguousarray(ref12, dtype=np.float64)
    B, N = x12.shape[0], params.N
    X = np.zeros((B, 12 * (N + 1))); U = np.zeros((B, 4 * N))
    it = np.zeros((B, 3), dtype=np.int32); st = np.zeros((B, 3), dtype=np.int32)
    ok = lib().mpc12_batch3(C.byref(params), C.byref(sxy), C.byref(sz), C.byref(syaw), B, _p(x12),
                            _p(ref12), _p(X), _p(U), _pi(it), _pi(st))
    return dict(X=X, U=U, iters=it, status=st, n_ok=ok)


class Mpc6:
    """Stateful single-instance oracle mirroring MPC::mpcLoop (warm start kept between calls)."""

    def __init__(self, params, settings):
        self.params = params
        self._h = lib().mpc6_create(C.byref(params), C.byref(settings))
        if not self._h:
            raise RuntimeError("mpc6_create failed")

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                lib().mpc6_destroy(self._h)
            except TypeError:        # interpreter shutdown: the module's globals are already gone
                pass
            self._h = None

    def loop(self, x0, ref):
        N = self.params.N
        x0 = np.ascontiguousarray(x0, dtype=np.float64); ref = np.ascontiguousarray(ref, dtype=np.float64)
        X = np.zeros(6 * (N + 1)); U = np.zeros(3 * N)
        it, st = C.c_int(), C.c_int()
        ok = lib().mpc6_loop(self._h, _p(x0), _p(ref), _p(X), _p(U), C.byref(it), C.byref(st))
        return dict(ok=bool(ok), X=X, U=U, iters=it.value, status=st.value)


class Mpc12:
    def __init__(self, params, settings):
        self.params = params
        sxy, sz, syaw = _three(settings)
        self._h = lib().mpc12_create3(C.byref(params), C.byref(sxy), C.byref(sz), C.byref(syaw))
        if not self._h:
            raise RuntimeError("mpc12_create failed")

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                lib().mpc12_destroy(self._h)
            except TypeError:
                pass
            self._h = None

    def loop(self, x12, ref12):
        N = self.params.N
        x12 = np.ascontiguousarray(x12, dtype=np.float64); ref12 = np.ascontiguousarray(ref12, dtype=np.float64)
        X = np.zeros(12 * (N + 1)); U = np.zeros(4 * N)
        it = np.zeros(3, dtype=np.int32); st = np.zeros(3, dtype=np.int32)
        ninter = (4 * N + 3) * 2 + 12 * N + (8 * N + 6) + 3 * (N + 1)
        inter = np.zeros(ninter)
        ok = lib().mpc12_loop(self._h, _p(x12), _p(ref12), _p(X), _p(U), _pi(it), _pi(st), _p(inter))
        o = 0
        parts = {}
        for name, k in (("z_sol", 4 * N + 3), ("xy_max", 6 * N), ("mixed_max", 6 * N),
                        ("xy_sol", 8 * N + 6), ("yaw_ref", 3 * (N + 1)), ("yaw_sol", 4 * N + 3)):
            parts[name] = inter[o:o + k].copy(); o += k
        return dict(ok=bool(ok), X=X, U=U, iters=it, status=st, **parts)


# ------------------------------------------------------------------ synthetic inputs
def synth_mpc6(seed, N, dt, B, first=0):
    x0 = np.zeros((B, 6)); ref = np.zeros((B, 6 * (N + 1)))
    lib().synth_mpc6(seed, N, dt, B, first, _p(x0), _p(ref))
    return x0, ref


def synth_mpc12(seed, N, dt, B, first=0):
    x = np.zeros((B, 12)); ref = np.zeros((B, 12 * (N + 1)))
    lib().synth_mpc12(seed, N, dt, B, first, _p(x), _p(ref))
    return x, ref
