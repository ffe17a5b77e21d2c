"""Python mirror of the reference's `class MPC12STATE`
(include/trajectory_generation/mpc_12state.hpp:492-683, src/mpc_12state.cpp) on top of the MI355X
engine: same method names, argument meaning, bool returns and call order; Eigen vectors become
numpy arrays.  Every solve runs on the GPU through libtg_mpc.so (B = 1)."""
import numpy as np

from . import _capi as capi
from .engine import BatchedMPC12, make_opts, mpc12_params

NUM_OF_STATES = 12  # mpc_12state.hpp:86
NUM_OF_INPUTS = 4   # mpc_12state.hpp:87

# CSV header of saveMPCSolutionsToFile (mpc_12state.cpp:2113), consumed by scripts/plot_trajectories.py
_STATE_NAMES = ["x", "vx", "ax", "y", "vy", "ay", "z", "vz", "az", "yaw", "vyaw", "ayaw"]


def _err(msg):
    print("\033[1;31m[ERROR] %s\033[0m" % msg, flush=True)


def _warn(msg):
    print("\033[1;33m[INFO] %s\033[0m" % msg, flush=True)


def _info(msg):
    print("\033[1;32m[INFO] %s\033[0m" % msg, flush=True)


def _nonneg_setter(attr, label, default):
    def setter(self, v):
        if v < 0:
            _err("%s %0.3f < 0. Setting default of %s" % (label, v, default))
            setattr(self, attr, float(default))
            return False
        setattr(self, attr, float(v))
        return True
    return setter


def _weight_setter(attr, label):
    def setter(self, w):
        if w < 0.0:
            _err("%s = %f < 0" % (label, w))
            return False
        setattr(self, attr, float(w))
        return True
    return setter


class MPC12STATE:
    def __init__(self, device=0, solver_opts=None, rho_xyz=None):
        # constructor defaults, mpc_12state.cpp:41-64
        self._debug = False
        self._dt = 0.05
        self._state_received = False
        self._received_refTraj = False
        self._alt_above_target = 1.0
        self._mpcWindow = 20
        self._xy_state_weight = self._z_state_weight = self._yaw_state_weight = 1.0
        self._xy_input_weight = self._z_input_weight = self._yaw_input_weight = 0.1
        self._xy_smooth_input_weight = self._z_smooth_input_weight = self._yaw_smooth_input_weight = 10.0
        self._enable_control_smoothing = False
        self._minAltitude = 1.0
        self._is_MPC_initialized = False
        for name in ("_xy_MaxVel", "_z_MaxVel", "_yaw_MaxVel", "_xy_MaxAccel", "_z_MaxAccel",
                     "_yaw_MaxAccel", "_xy_MaxJerk", "_z_MaxJerk", "_yaw_MaxJerk"):
            setattr(self, name, None)
        self._outputCSVFile = ""
        self._device, self._solver_opts, self._rho_xyz = device, solver_opts, rho_xyz
        self._engine = None
        self._current_state = np.zeros(12)
        self._referenceTraj = np.zeros(0)
        self._x_opt = np.zeros(0)
        self._u_opt = np.zeros(0)
        self.last_status = np.full(3, capi.TG_UNSOLVED)
        self.last_iters = np.zeros(3, dtype=int)

    # ------------------------------------------------------------ setters (mpc_12state.cpp:1571-1932)
    def setDt(self, dt):
        if dt > 0:
            self._dt = float(dt)
            return True
        _err("[MPC12STATE::setDt] dt < 0. Defaulting to 0.1")
        self._dt = 0.1
        return False

    def setDebug(self, d):
        self._debug = bool(d)

    def setMPCWindow(self, N):
        if N > 0:
            self._mpcWindow = int(N)
            return True
        _err("_mpcWindow < 0. Default to 10")
        self._mpcWindow = 10
        return False

    setXYStateWeight = _weight_setter("_xy_state_weight", "xy_state_weight")
    setZStateWeight = _weight_setter("_z_state_weight", "z_state_weight")
    setYawStateWeight = _weight_setter("_yaw_state_weight", "yaw_state_weight")
    setXYInputWeight = _weight_setter("_xy_input_weight", "_xy_input_weight")
    setZInputWeight = _weight_setter("_z_input_weight", "_z_input_weight")
    setYawInputWeight = _weight_setter("_yaw_input_weight", "_yaw_input_weight")
    setXYSmoothInputWeight = _weight_setter("_xy_smooth_input_weight", "_xy_smooth_input_weight")
    setZSmoothInputWeight = _weight_setter("_z_smooth_input_weight", "_z_smooth_input_weight")
    setYawSmoothInputWeight = _weight_setter("_yaw_smooth_input_weight", "_yaw_smooth_input_weight")

    # out-of-range limits install the reference's fall-back constants (mpc_12state.cpp:1817-1932)
    setXYMaxVel = _nonneg_setter("_xy_MaxVel", "maxXYVel", 10.0)
    setZMaxVel = _nonneg_setter("_z_MaxVel", "maxZVel", 10.0)
    setYawMaxVel = _nonneg_setter("_yaw_MaxVel", "maxYawVel", 100.0)
    setXYMaxAccel = _nonneg_setter("_xy_MaxAccel", "maxXYAccel", 5.0)
    setZMaxAccel = _nonneg_setter("_z_MaxAccel", "maxZAccel", 5.0)
    setYawMaxAccel = _nonneg_setter("_yaw_MaxAccel", "maxYawAccel", 100.0)
    setXYMaxJerk = _nonneg_setter("_xy_MaxJerk", "_xy_MaxJerk", 5.0)
    setZMaxJerk = _nonneg_setter("_z_MaxJerk", "_z_MaxJerk", 5.0)
    setYawMaxJerk = _nonneg_setter("_yaw_MaxJerk", "_yaw_MaxJerk", 100.0)

    def enableControlSmoothing(self, b):
        self._enable_control_smoothing = bool(b)

    def setAltAboveTarget(self, h):
        if h < 0:
            _err("_alt_above_target = %f < 0" % h)
            return False
        self._alt_above_target = float(h)
        return True

    def setMinimumAltitude(self, h):
        # accepted and, as in the reference, not used by any stage (mpc_12state.cpp:260-270)
        self._minAltitude = float(h)
        return True

    def setCurrentState(self, x):
        x = np.asarray(x, dtype=np.float64).ravel()
        if x.size != NUM_OF_STATES:
            _err("State vector size %d != %u" % (x.size, NUM_OF_STATES))
            return False
        self._current_state = x.copy()
        self._state_received = True
        return True

    def setReferenceTraj(self, v):
        # the argument's size is checked (the reference checks its own member, :1937)
        v = np.asarray(v, dtype=np.float64).ravel()
        want = NUM_OF_STATES * (self._mpcWindow + 1)
        if v.size != want:
            _err("[MPC12STATE::setReferenceTraj] input matrix size %d != %d" % (v.size, want))
            return False
        self._referenceTraj = v.copy()
        self._received_refTraj = True
        return True

    def setOutputFilePath(self, path):
        self._outputCSVFile = str(path)

    # ------------------------------------------------------------ initMPCProblem (:1282)
    def initMPCProblem(self):
        lim = [self._xy_MaxVel, self._z_MaxVel, self._yaw_MaxVel, self._xy_MaxAccel, self._z_MaxAccel,
               self._yaw_MaxAccel, self._xy_MaxJerk, self._z_MaxJerk, self._yaw_MaxJerk]
        if any(v is None for v in lim):
            _err("[MPC12STATE::initMPCProblem] all velocity / acceleration / jerk limits must be set first")
            return False
        p = mpc12_params(N=self._mpcWindow, dt=self._dt,
                         enable_control_smoothing=self._enable_control_smoothing,
                         state_weight=(self._xy_state_weight, self._z_state_weight, self._yaw_state_weight),
                         input_weight=(self._xy_input_weight, self._z_input_weight, self._yaw_input_weight),
                         smooth_input_weight=(self._xy_smooth_input_weight, self._z_smooth_input_weight,
                                              self._yaw_smooth_input_weight),
                         max_vel=lim[0:3], max_accel=lim[3:6], max_jerk=lim[6:9])
        try:
            self._engine = BatchedMPC12(p, self._solver_opts or make_opts(), self._rho_xyz, self._device)
        except capi.TgError as e:
            _err("[MPC12STATE::initMPCProblem] MPC initialization is not successful. %s" % e)
            return False
        N = self._mpcWindow
        self._current_state = np.zeros(12)
        self._referenceTraj = np.zeros(12 * (N + 1))
        self._x_opt = np.zeros(12 * (N + 1))
        self._u_opt = np.zeros(4 * N)
        self._received_refTraj = False     # initVariables, :1185-1186
        self._state_received = False
        self._is_MPC_initialized = True
        return True

    # ------------------------------------------------------------ mpcLoop (:1374)
    def mpcLoop(self):
        if not self._is_MPC_initialized:
            _warn("[MPC12STATE::mpcLoop] MPC controller is not initialized. Skipping MPC loop.")
            return False
        if not self._state_received:
            _warn("[MPC12STATE::mpcLoop] curent state is not received. Skipping MPC loop.")
            return False
        if not self._received_refTraj:
            _warn("[MPC12STATE::mpcLoop] Reference trajectory is not received. Skipping MPC loop.")
            return False
        r = self._engine.solve_host(self._current_state[None, :], self._referenceTraj[None, :], warm=False)
        self.last_status, self.last_iters = r["status"][0].copy(), r["iters"][0].copy()
        for k, name in enumerate(("Z", "XY", "Yaw")):
            if self.last_status[k] != capi.TG_SOLVED:
                _err("[MPC12STATE::mpcLoop] %s -  MPC solution is not found" % name)
                return False
        self._x_opt = r["X"][0].copy()
        self._u_opt = r["U"][0].copy()
        return True

    # the per-stage update entry points only refresh problem data in the reference
    # (:1357-1372); the engine rebuilds it inside every solve
    def updateXYMPC(self):
        return self._is_MPC_initialized

    def updateZMPC(self):
        return self._is_MPC_initialized

    def updateYawMPC(self):
        return self._is_MPC_initialized

    # ------------------------------------------------------------ getters (:1965-2098)
    def getOptimalStateTraj(self):
        return self._x_opt.copy()

    def getOptimalControlTraj(self):
        return self._u_opt.copy()

    def getNumOfStates(self):
        return NUM_OF_STATES

    def getNumOfInputs(self):
        return NUM_OF_INPUTS

    def getXYMaxVel(self): return self._xy_MaxVel
    def getZMaxVel(self): return self._z_MaxVel
    def getYawMaxVel(self): return self._yaw_MaxVel
    def getXYMaxAccel(self): return self._xy_MaxAccel
    def getZMaxAccel(self): return self._z_MaxAccel
    def getYawMaxAccel(self): return self._yaw_MaxAccel
    def getXYMaxJerk(self): return self._xy_MaxJerk
    def getZMaxJerk(self): return self._z_MaxJerk
    def getYawMaxJerk(self): return self._yaw_MaxJerk

    @staticmethod
    def _blkdiag(*mats):
        r = sum(m.shape[0] for m in mats); c = sum(m.shape[1] for m in mats)
        out = np.zeros((r, c)); i = j = 0
        for m in mats:
            out[i:i + m.shape[0], j:j + m.shape[1]] = m
            i += m.shape[0]; j += m.shape[1]
        return out

    def _stage(self, stage, which):
        return self._engine.get(stage, which)

    def getTransitionMatrix(self):
        return self._blkdiag(*[self._stage(s, capi.TG_MAT_AD) for s in ("xy", "z", "yaw")])

    def getInputMatrix(self):
        return self._blkdiag(*[self._stage(s, capi.TG_MAT_BD) for s in ("xy", "z", "yaw")])

    def getQ(self):
        return self._blkdiag(*[self._stage(s, capi.TG_MAT_Q) for s in ("xy", "z", "yaw")])

    def getR(self):
        # 4 x 4 block diagonal; the reference sizes this matrix from Q by mistake (:2051)
        return self._blkdiag(*[self._stage(s, capi.TG_MAT_R) for s in ("xy", "z", "yaw")])

    def getXYTransitionMatrix(self): return self._stage("xy", capi.TG_MAT_AD)
    def getZTransitionMatrix(self): return self._stage("z", capi.TG_MAT_AD)
    def getYawTransitionMatrix(self): return self._stage("yaw", capi.TG_MAT_AD)
    def getXYInputMatrix(self): return self._stage("xy", capi.TG_MAT_BD)
    def getZInputMatrix(self): return self._stage("z", capi.TG_MAT_BD)
    def getYawInputMatrix(self): return self._stage("yaw", capi.TG_MAT_BD)
    def getXYContraintsMatrix(self): return self._stage("xy", capi.TG_MAT_A)
    def getZContraintsMatrix(self): return self._stage("z", capi.TG_MAT_A)
    def getYawContraintsMatrix(self): return self._stage("yaw", capi.TG_MAT_A)
    def getXYHessianMatrix(self): return self._stage("xy", capi.TG_MAT_P)
    def getZHessianMatrix(self): return self._stage("z", capi.TG_MAT_P)
    def getYawHessianMatrix(self): return self._stage("yaw", capi.TG_MAT_P)
    def getXYQ(self): return self._stage("xy", capi.TG_MAT_Q)
    def getZQ(self): return self._stage("z", capi.TG_MAT_Q)
    def getYawQ(self): return self._stage("yaw", capi.TG_MAT_Q)
    def getXYR(self): return self._stage("xy", capi.TG_MAT_R)
    def getZR(self): return self._stage("z", capi.TG_MAT_R)
    def getYawR(self): return self._stage("yaw", capi.TG_MAT_R)

    def _stage_vectors(self, stage):
        """q, l, u of a stage for the current state / reference / last solution."""
        import torch
        e = self._engine
        buf = e.alloc(1)
        dev = buf["x0"].device
        buf["x0"][:, 0] = torch.from_numpy(self._current_state).to(dev)
        buf["ref"][:, 0] = torch.from_numpy(self._referenceTraj).to(dev)
        buf["X"][:, 0] = torch.from_numpy(self._x_opt).to(dev)
        q, l, u = e.assemble_stage(stage, buf)
        return q[:, 0].cpu().numpy(), l[:, 0].cpu().numpy(), u[:, 0].cpu().numpy()

    def getXYGradient(self): return self._stage_vectors("xy")[0]
    def getZGradient(self): return self._stage_vectors("z")[0]
    def getYawGradient(self): return self._stage_vectors("yaw")[0]
    def getGradient(self):
        return np.concatenate([self.getXYGradient(), self.getZGradient(), self.getYawGradient()])
    def getXYLowerBounds(self): return self._stage_vectors("xy")[1]
    def getZLowerBounds(self): return self._stage_vectors("z")[1]
    def getYawLowerBounds(self): return self._stage_vectors("yaw")[1]
    def getXYUpperBounds(self): return self._stage_vectors("xy")[2]
    def getZUpperBounds(self): return self._stage_vectors("z")[2]
    def getYawUpperBounds(self): return self._stage_vectors("yaw")[2]

    # ------------------------------------------------------------ saveMPCSolutionsToFile (:2107)
    _CSV_HEADER = ("time,x,v_x,a_x,y,v_y,a_y,z,v_z,a_z,yaw,v_yaw,a_yaw,j_x,j_y,j_z,j_yaw,des_x,des_vx,des_ax,"
                   "des_y,des_vy,des_ay,des_z,des_vz,des_az,des_yaw,des_v_yaw,des_a_yaw,xy_VelMin,xy_VelMax,"
                   "xy_AccelMin,xy_AccelMax,xy_JerkMin,xy_JerkMax,z_VelMin,z_VelMax,z_AccelMin,z_AccelMax,"
                   "z_JerkMin,z_JerkMax,yaw_VelMin,yaw_VelMax,yaw_AccelMin,yaw_AccelMax,yaw_JerkMin,yaw_JerkMax")

    def saveMPCSolutionsToFile(self):
        """Same file as the reference writes (column names are what scripts/plot_trajectories.py
        reads): a header, one row for the current state (29 fields, no limits), then N rows.  As
        in the reference, row i+1 carries time (i+1)*dt together with state block i of the optimal
        trajectory and reference block i+1 (mpc_12state.cpp:2132-2162)."""
        N, dt = self._mpcWindow, self._dt
        X, U, R = self._x_opt, self._u_opt, self._referenceTraj
        yaw_ref = np.zeros(N + 1)
        if self._engine is not None and self.last_status[2] != capi.TG_UNSOLVED:
            yaw_ref = self._engine.intermediates(1)[2][:, 0]
        lim = []
        for v, a, j in ((self._xy_MaxVel, self._xy_MaxAccel, self._xy_MaxJerk),
                        (self._z_MaxVel, self._z_MaxAccel, self._z_MaxJerk),
                        (self._yaw_MaxVel, self._yaw_MaxAccel, self._yaw_MaxJerk)):
            lim += [-v, v, -a, a, -j, j]
        fmt = lambda v: "%g" % v   # std::ostream default formatting (6 significant digits)
        try:
            with open(self._outputCSVFile, "w") as f:
                f.write(self._CSV_HEADER + "\n")
                f.write("0.0" + "".join("," + fmt(v) for v in self._current_state) + ",0.0" * NUM_OF_INPUTS
                        + "".join("," + fmt(v) for v in R[:12]) + "\n")
                for i in range(N):
                    row = ([(i + 1) * dt] + list(X[12 * i:12 * i + 12]) + list(U[4 * i:4 * i + 4])
                           + list(R[12 * (i + 1):12 * (i + 1) + 9]) + [yaw_ref[i + 1], 0.0, 0.0] + lim)
                    f.write(",".join(fmt(v) for v in row) + "\n")
            _info("[MPC12STATE::saveMPCSolutionsToFile] Saved MPC solutions to file: %s" % self._outputCSVFile)
        except OSError:
            _err("[MPCROS::saveMPCSolutionsToFile] Coudl not open file %s" % self._outputCSVFile)
