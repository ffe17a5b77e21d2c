"""Batched engines over the C ABI: device-resident structure-of-arrays in, trajectories out.

Device arrays are torch CUDA tensors of dtype float64 / int32 laid out [rows, ld]
(instance-minor).  torch is used for device memory and streams only.
"""
import ctypes as C

import numpy as np

from . import _capi as capi


def _torch():
    import torch
    return torch


def make_opts(rho=0.0, sigma=1e-6, alpha=1.6, eps_abs=1e-3, eps_rel=1e-3, max_iter=4000,
              check_termination=25, rho_scale=(0.0, 0.0, 0.0, 0.0), eps_prim_inf=1e-4, check_fine=0, grid_cap=0, lowrank=0, hard_first=0,
              phase_iters=(0, 0), phase_scale=((0.0,) * 4, (0.0,) * 4), overlap=0):
    """rho <= 0 and rho_scale entries of 0 select the engine's built-in choices; rho_scale is
    (position, velocity, acceleration, input) multipliers of rho on the box rows;
    eps_prim_inf = 0 switches the primal infeasibility test off; check_fine > 0: after the first
    check (iteration check_termination) test every check_fine iterations, 0: OSQP's fixed interval;
    grid_cap: most workgroups a solve launches (0: what the GPU holds at once, < 0: one per tile of
    16 instances); a smaller grid refills its instance slots from the rest of the batch;
    lowrank: K^-1 as diagonal tiles + low-rank rest (0: where it pays, 1: wherever possible, -1: dense).
    hard_first: a refilled grid takes the likely slow instances first (1), the batch in its order (-1),
    or whichever pays (0: the former for horizons longer than 30; with a step-size schedule only for batches of at most 49152).
    phase_iters / phase_scale: the step-size schedule -- an instance still running at the first check
    at or after phase_iters[p] continues under an operator whose class multipliers are rho_scale x
    phase_scale[p] (0: built-in choice, phase_iters < 0: no such phase).
    overlap (12-state chain): 1 = the yaw stage of the vehicles done with XY runs beside the XY stage's continuation
    launches (shorter latency of a batch that runs alone), 0 = stages strictly in turn; same results."""
    o = capi.SolverOpts()
    capi.load().tg_default_opts(C.byref(o))
    o.rho_scale[:] = [float(v) for v in rho_scale]
    o.rho, o.sigma, o.alpha = rho, sigma, alpha
    o.eps_abs, o.eps_rel, o.max_iter, o.check_termination = eps_abs, eps_rel, max_iter, check_termination
    o.eps_prim_inf = eps_prim_inf
    o.check_fine = check_fine
    o.grid_cap = grid_cap
    o.lowrank = lowrank
    o.hard_first = hard_first
    for p in range(2):
        o.phase_iters[p] = int(phase_iters[p])
        for k in range(4):
            o.phase_scale[p][k] = float(phase_scale[p][k])
    o.overlap = int(overlap)
    return o


def mpc6_params(N=20, dt=0.1, state_weight=7000.0, input_weight=500.0, smooth_input_weight=10.0,
                enable_control_smoothing=False, min_altitude=-1.0, max_vel=(10.0, 10.0, 10.0),
                max_accel=(5.0, 5.0, 5.0)):
    """Defaults: the reference's test harness (test/test_mpc_6dof.cpp:16-53) with smoothing as
    in config/mpc.yaml:22."""
    p = capi.Mpc6Params()
    p.N, p.dt = int(N), float(dt)
    p.state_weight, p.input_weight, p.smooth_input_weight = state_weight, input_weight, smooth_input_weight
    p.enable_control_smoothing = int(bool(enable_control_smoothing))
    p.min_altitude = min_altitude
    p.max_vel[:] = [float(v) for v in max_vel]
    p.max_accel[:] = [float(v) for v in max_accel]
    return p


def mpc12_params(N=20, dt=0.1, enable_control_smoothing=False,
                 state_weight=(5000.0, 7000.0, 1000.0), input_weight=(200.0, 200.0, 100.0),
                 smooth_input_weight=(100.0, 100.0, 100.0), max_vel=(12.0, 7.0, 2.0),
                 max_accel=(6.0, 10.0, 10.0), max_jerk=(30.0, 40.0, 10.0)):
    """Triples are (xy, z, yaw); defaults are config/mpc_12state.yaml:23-53."""
    p = capi.Mpc12Params()
    p.N, p.dt, p.enable_control_smoothing = int(N), float(dt), int(bool(enable_control_smoothing))
    p.xy_state_weight, p.z_state_weight, p.yaw_state_weight = state_weight
    p.xy_input_weight, p.z_input_weight, p.yaw_input_weight = input_weight
    p.xy_smooth_input_weight, p.z_smooth_input_weight, p.yaw_smooth_input_weight = smooth_input_weight
    p.xy_max_vel, p.z_max_vel, p.yaw_max_vel = max_vel
    p.xy_max_accel, p.z_max_accel, p.yaw_max_accel = max_accel
    p.xy_max_jerk, p.z_max_jerk, p.yaw_max_jerk = max_jerk
    return p


def _ptr(tensor):
    return None if tensor is None else C.c_void_p(tensor.data_ptr())


def _check_dev(tensor, rows, ld, dtype, name):
    torch = _torch()
    if not (isinstance(tensor, torch.Tensor) and tensor.is_cuda and tensor.dtype == dtype
            and tensor.is_contiguous() and tuple(tensor.shape) == (rows, ld)):
        raise ValueError("%s must be a contiguous cuda %s tensor of shape (%d, %d)" % (name, dtype, rows, ld))


def _stream_ptr(stream):
    if stream is None:
        return C.c_void_p(_torch().cuda.current_stream().cuda_stream)
    return C.c_void_p(stream.cuda_stream)


class BatchedMPC6:
    """B independent MPC::mpcLoop calls per launch (src/mpc_6dof.cpp:616-649)."""

    nx, nu = 6, 3

    def __init__(self, params=None, opts=None, device=0):
        self.lib = capi.load()
        self.params = params if params is not None else mpc6_params()
        self.opts = opts if opts is not None else make_opts()
        self.device = device
        h = C.c_void_p()
        capi.check(self.lib.tg_mpc6_create(C.byref(self.params), C.byref(self.opts), device, C.byref(h)))
        self._h = h
        n, m = C.c_int(), C.c_int()
        capi.check(self.lib.tg_mpc6_dims(self._h, C.byref(n), C.byref(m)))
        self.n, self.m, self.N = n.value, m.value, self.params.N

    def close(self):
        if getattr(self, "_h", None):
            self.lib.tg_mpc6_destroy(self._h)
            self._h = None

    __del__ = close

    @property
    def rho(self):
        return self.lib.tg_mpc6_rho(self._h)

    def kernel_info(self):
        """(MFMA instructions per ADMM iteration of a 16-instance tile, low-rank form in use)."""
        n, lr = C.c_int(), C.c_int()
        capi.check(self.lib.tg_mpc6_kernel_info(self._h, C.byref(n), C.byref(lr)))
        return n.value, bool(lr.value)

    def rho_rows(self):
        """[(per-row rho vector, rho, nx, N)]: the step sizes the solver was built with."""
        return [(self.get(capi.TG_VEC_RHO), self.rho, 6, self.N)]

    def get(self, which):
        N = self.N
        shape = {capi.TG_MAT_P: (self.n, self.n), capi.TG_MAT_A: (self.m, self.n),
                 capi.TG_MAT_KINV: (self.n, self.n), capi.TG_MAT_AD: (6, 6), capi.TG_MAT_BD: (6, 3),
                 capi.TG_MAT_Q: (6, 6), capi.TG_MAT_R: (3, 3), capi.TG_VEC_L: (self.m,),
                 capi.TG_VEC_U: (self.m,), capi.TG_VEC_RHO: (self.m,)}[which]
        out = np.zeros(shape)
        capi.check(self.lib.tg_mpc6_get(self._h, which, out.ctypes.data_as(C.c_void_p)))
        return out

    def alloc(self, B, ld=None):
        """Device arrays for a batch: dict(x0, ref, X, U, status, iters)."""
        torch = _torch()
        ld = ld or (B + 15) // 16 * 16
        dev = torch.device("cuda", self.device)
        f = lambda r: torch.zeros((r, ld), dtype=torch.float64, device=dev)
        return dict(x0=f(6), ref=f(6 * (self.N + 1)), X=f(6 * (self.N + 1)), U=f(3 * self.N),
                    status=torch.zeros(ld, dtype=torch.int32, device=dev),
                    iters=torch.zeros(ld, dtype=torch.int32, device=dev), ld=ld, B=B)

    def solve(self, buf, warm=False, stream=None, B=None):
        B = buf["B"] if B is None else B
        ld, N = buf["ld"], self.N
        torch = _torch()
        _check_dev(buf["x0"], 6, ld, torch.float64, "x0")
        _check_dev(buf["ref"], 6 * (N + 1), ld, torch.float64, "ref")
        _check_dev(buf["X"], 6 * (N + 1), ld, torch.float64, "X")
        _check_dev(buf["U"], 3 * N, ld, torch.float64, "U")
        capi.check(self.lib.tg_mpc6_solve_batch(self._h, B, ld, _ptr(buf["x0"]), _ptr(buf["ref"]),
                                                _ptr(buf["X"]), _ptr(buf["U"]), _ptr(buf["status"]),
                                                _ptr(buf["iters"]), int(warm), _stream_ptr(stream)))

    def assemble(self, buf, B=None, stream=None):
        """Full q [n, ld], l [m, ld], u [m, ld] of every instance (reference ordering)."""
        torch = _torch()
        B = buf["B"] if B is None else B
        ld = buf["ld"]
        dev = buf["x0"].device
        q = torch.zeros((self.n, ld), dtype=torch.float64, device=dev)
        l = torch.zeros((self.m, ld), dtype=torch.float64, device=dev)
        u = torch.zeros((self.m, ld), dtype=torch.float64, device=dev)
        capi.check(self.lib.tg_mpc6_assemble_batch(self._h, B, ld, _ptr(buf["x0"]), _ptr(buf["ref"]),
                                                   _ptr(q), _ptr(l), _ptr(u), _stream_ptr(stream)))
        return q, l, u

    def solve_host(self, x0, ref, warm=False):
        """x0 [B, 6], ref [B, 6(N+1)] numpy (instance-major, the reference's vectors)."""
        x0 = np.ascontiguousarray(x0, dtype=np.float64).reshape(-1, 6)
        B, N = x0.shape[0], self.N
        ref = np.ascontiguousarray(ref, dtype=np.float64).reshape(B, 6 * (N + 1))
        X = np.zeros((B, 6 * (N + 1))); U = np.zeros((B, 3 * N))
        st = np.zeros(B, dtype=np.int32); it = np.zeros(B, dtype=np.int32)
        vp = lambda a_: a_.ctypes.data_as(C.c_void_p)
        capi.check(self.lib.tg_mpc6_solve_host(self._h, B, vp(x0), vp(ref), vp(X), vp(U), vp(st), vp(it),
                                               int(warm)))
        return dict(X=X, U=U, status=st, iters=it)

    def reset_warm(self):
        capi.check(self.lib.tg_mpc6_reset_warm(self._h))

    NX, NU = 6, 3

    def plant_step(self, buf, stream=None, B=None, gated=True):
        """x0 <- A x0 + B u(0) with the U of the last solve (test_mpc_6dof.cpp:115-122).  gated:
        an instance whose solve failed keeps following the last plan it received (its U moves one
        step up); ungated, row 0 of U is applied whatever the status."""
        B = buf["B"] if B is None else B
        capi.check(self.lib.tg_mpc6_plant_step(self._h, B, buf["ld"], _ptr(buf["x0"]), _ptr(buf["U"]),
                                               _ptr(buf["status"]) if gated else None, _stream_ptr(stream)))

    def get_opts(self):
        o = capi.SolverOpts()
        capi.check(self.lib.tg_mpc6_get_opts(self._h, C.byref(o)))
        return o

    def set_opts(self, opts):
        capi.check(self.lib.tg_mpc6_set_opts(self._h, C.byref(opts)))

    def last_residuals(self, B):
        """(primal, dual) residual of instances [0, B) at the check that ended the last solve."""
        pr, du = np.zeros(B), np.zeros(B)
        capi.check(self.lib.tg_mpc6_last_residuals(self._h, B, pr.ctypes.data_as(C.c_void_p), du.ctypes.data_as(C.c_void_p)))
        return pr, du

    def ref_shift(self, buf, stream=None, B=None):
        B = buf["B"] if B is None else B
        capi.check(self.lib.tg_ref_shift(B, buf["ld"], 6, self.N, _ptr(buf["ref"]), _stream_ptr(stream)))


class BatchedMPC12:
    """B independent MPC12STATE::mpcLoop calls per launch (src/mpc_12state.cpp:1374-1453)."""

    nx, nu = 12, 4
    STAGES = {"xy": capi.TG_FAM_XY, "z": capi.TG_FAM_Z, "yaw": capi.TG_FAM_YAW}

    def __init__(self, params=None, opts=None, rho_xyz=None, device=0):
        self.lib = capi.load()
        self.params = params if params is not None else mpc12_params()
        self.opts = opts if opts is not None else make_opts()
        self.device = device
        h = C.c_void_p()
        rp = None
        if rho_xyz is not None:
            arr = (C.c_double * 3)(*[float(v) for v in rho_xyz])
            rp = C.cast(arr, C.c_void_p)
        capi.check(self.lib.tg_mpc12_create(C.byref(self.params), C.byref(self.opts), rp, device, C.byref(h)))
        self._h = h
        self.N = self.params.N

    def close(self):
        if getattr(self, "_h", None):
            self.lib.tg_mpc12_destroy(self._h)
            self._h = None

    __del__ = close

    def dims(self, stage):
        n, m = C.c_int(), C.c_int()
        capi.check(self.lib.tg_mpc12_dims(self._h, self.STAGES[stage], C.byref(n), C.byref(m)))
        return n.value, m.value

    def rho(self, stage):
        return self.lib.tg_mpc12_rho(self._h, self.STAGES[stage])

    def kernel_info(self, stage):
        """(MFMA instructions per ADMM iteration of a 16-instance tile, low-rank form in use)."""
        n, lr = C.c_int(), C.c_int()
        capi.check(self.lib.tg_mpc12_kernel_info(self._h, self.STAGES[stage], C.byref(n), C.byref(lr)))
        return n.value, bool(lr.value)

    def rho_rows(self):
        N = self.params.N
        return [(self.get(s, capi.TG_VEC_RHO), self.rho(s), 6 if s == "xy" else 3, N, s) for s in ("xy", "z", "yaw")]

    def get(self, stage, which):
        n, m = self.dims(stage)
        nx, nu = (6, 2) if stage == "xy" else (3, 1)
        shape = {capi.TG_MAT_P: (n, n), capi.TG_MAT_A: (m, n), capi.TG_MAT_KINV: (n, n),
                 capi.TG_MAT_AD: (nx, nx), capi.TG_MAT_BD: (nx, nu), capi.TG_MAT_Q: (nx, nx),
                 capi.TG_MAT_R: (nu, nu), capi.TG_VEC_L: (m,), capi.TG_VEC_U: (m,),
                 capi.TG_VEC_RHO: (m,)}[which]
        out = np.zeros(shape)
        capi.check(self.lib.tg_mpc12_get(self._h, self.STAGES[stage], which, out.ctypes.data_as(C.c_void_p)))
        return out

    def alloc(self, B, ld=None):
        torch = _torch()
        ld = ld or (B + 15) // 16 * 16
        dev = torch.device("cuda", self.device)
        f = lambda r: torch.zeros((r, ld), dtype=torch.float64, device=dev)
        return dict(x0=f(12), ref=f(12 * (self.N + 1)), X=f(12 * (self.N + 1)), U=f(4 * self.N),
                    status=torch.zeros((3, ld), dtype=torch.int32, device=dev),
                    iters=torch.zeros((3, ld), dtype=torch.int32, device=dev), ld=ld, B=B)

    def solve(self, buf, warm=False, stream=None, B=None):
        B = buf["B"] if B is None else B
        ld, N = buf["ld"], self.N
        torch = _torch()
        _check_dev(buf["x0"], 12, ld, torch.float64, "x0")
        _check_dev(buf["ref"], 12 * (N + 1), ld, torch.float64, "ref")
        _check_dev(buf["X"], 12 * (N + 1), ld, torch.float64, "X")
        _check_dev(buf["U"], 4 * N, ld, torch.float64, "U")
        capi.check(self.lib.tg_mpc12_solve_batch(self._h, B, ld, _ptr(buf["x0"]), _ptr(buf["ref"]),
                                                 _ptr(buf["X"]), _ptr(buf["U"]), _ptr(buf["status"]),
                                                 _ptr(buf["iters"]), int(warm), _stream_ptr(stream)))

    def solve_static(self, buf, stream=None, B=None):
        """solve() for callers that cannot allocate device memory: cold start, inter-stage arrays in
        the library's static device scratch (32 MiB per GPU; larger batches raise TgError)."""
        B = buf["B"] if B is None else B
        ld, N = buf["ld"], self.N
        torch = _torch()
        _check_dev(buf["x0"], 12, ld, torch.float64, "x0")
        _check_dev(buf["ref"], 12 * (N + 1), ld, torch.float64, "ref")
        _check_dev(buf["X"], 12 * (N + 1), ld, torch.float64, "X")
        _check_dev(buf["U"], 4 * N, ld, torch.float64, "U")
        capi.check(self.lib.tg_mpc12_solve_batch_static(self._h, B, ld, _ptr(buf["x0"]), _ptr(buf["ref"]),
                                                        _ptr(buf["X"]), _ptr(buf["U"]), _ptr(buf["status"]),
                                                        _ptr(buf["iters"]), _stream_ptr(stream)))

    def solve_stage(self, stage, buf, stream=None, B=None):
        """One stage with the upstream solution already in buf['X'] (stage-by-stage parity)."""
        B = buf["B"] if B is None else B
        capi.check(self.lib.tg_mpc12_solve_stage(self._h, self.STAGES[stage], B, buf["ld"], _ptr(buf["x0"]),
                                                 _ptr(buf["ref"]), _ptr(buf["X"]), _ptr(buf["U"]),
                                                 _ptr(buf["status"]), _ptr(buf["iters"]), _stream_ptr(stream)))

    def intermediates(self, B):
        """xy_max [6N, B], mixed_max [6N, B], yaw_ref [N+1, B] of the last solve (copies)."""
        torch = _torch()
        a, b, c, ld = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int()
        capi.check(self.lib.tg_mpc12_intermediates(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(ld)))
        N, ldv = self.N, ld.value
        out = []
        for ptr, rows in ((a, 6 * N), (b, 6 * N), (c, N + 1)):
            host = np.zeros((rows, ldv))
            torch.cuda.synchronize()
            capi.check(self.lib.tg_copy_to_host(host.ctypes.data_as(C.c_void_p), ptr, host.nbytes))
            out.append(host[:, :B].copy())
        return out

    def assemble_stage(self, stage, buf, B=None, stream=None):
        torch = _torch()
        B = buf["B"] if B is None else B
        ld = buf["ld"]
        n, m = self.dims(stage)
        dev = buf["x0"].device
        q = torch.zeros((n, ld), dtype=torch.float64, device=dev)
        l = torch.zeros((m, ld), dtype=torch.float64, device=dev)
        u = torch.zeros((m, ld), dtype=torch.float64, device=dev)
        capi.check(self.lib.tg_mpc12_assemble_stage(self._h, self.STAGES[stage], B, ld, _ptr(buf["x0"]),
                                                    _ptr(buf["ref"]), _ptr(buf["X"]), _ptr(q), _ptr(l),
                                                    _ptr(u), _stream_ptr(stream)))
        return q, l, u

    def solve_host(self, x0, ref, warm=False):
        x0 = np.ascontiguousarray(x0, dtype=np.float64).reshape(-1, 12)
        B, N = x0.shape[0], self.N
        ref = np.ascontiguousarray(ref, dtype=np.float64).reshape(B, 12 * (N + 1))
        X = np.zeros((B, 12 * (N + 1))); U = np.zeros((B, 4 * N))
        st = np.zeros((B, 3), dtype=np.int32); it = np.zeros((B, 3), dtype=np.int32)
        vp = lambda a_: a_.ctypes.data_as(C.c_void_p)
        capi.check(self.lib.tg_mpc12_solve_host(self._h, B, vp(x0), vp(ref), vp(X), vp(U), vp(st), vp(it),
                                                int(warm)))
        return dict(X=X, U=U, status=st, iters=it)

    def reset_warm(self):
        capi.check(self.lib.tg_mpc12_reset_warm(self._h))

    NX, NU = 12, 4

    def plant_step(self, buf, stream=None, B=None, gated=True):
        """x0 <- A x0 + B u(0), A / B = getTransitionMatrix / getInputMatrix
        (test_mpc_12state.cpp:193-197, mpc_12state.cpp:2001-2025); gated as for the 6-DOF engine."""
        B = buf["B"] if B is None else B
        capi.check(self.lib.tg_mpc12_plant_step(self._h, B, buf["ld"], _ptr(buf["x0"]), _ptr(buf["U"]),
                                                _ptr(buf["status"]) if gated else None, _stream_ptr(stream)))

    def get_opts(self, stage):
        o = capi.SolverOpts()
        capi.check(self.lib.tg_mpc12_get_opts(self._h, self.STAGES[stage], C.byref(o)))
        return o

    def set_opts(self, opts):
        capi.check(self.lib.tg_mpc12_set_opts(self._h, C.byref(opts)))

    def last_residuals(self, stage, B):
        pr, du = np.zeros(B), np.zeros(B)
        capi.check(self.lib.tg_mpc12_last_residuals(self._h, self.STAGES[stage], B, pr.ctypes.data_as(C.c_void_p),
                                                    du.ctypes.data_as(C.c_void_p)))
        return pr, du

    def ref_shift(self, buf, stream=None, B=None):
        B = buf["B"] if B is None else B
        capi.check(self.lib.tg_ref_shift(B, buf["ld"], 12, self.params.N, _ptr(buf["ref"]), _stream_ptr(stream)))


def phase_launches(opts, B=None):
    """Kernel launches one solve of a stage makes with these (resolved) options: one per operator of
    the step-size schedule that can come into use -- boundaries are rounded up to the check interval,
    must lie below max_iter, and two that coincide count once."""
    chk, bounds = int(opts.check_termination), []
    for p in range(2):
        it = int(opts.phase_iters[p])
        if it <= 0:
            break
        b = (it + chk - 1) // chk * chk
        if b >= int(opts.max_iter):
            break
        if not bounds or b > bounds[-1]:
            bounds.append(b)
    return 1 + len(bounds)


def timing_enable(on=True):
    capi.load().tg_timing_enable(int(on))


def timing_read(reset=True):
    ms, n = C.c_double(), C.c_int()
    capi.load().tg_timing_read(C.byref(ms), C.byref(n), int(reset))
    return ms.value, n.value
