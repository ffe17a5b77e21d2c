"""Python mirror of the reference's `class MPC` (include/trajectory_generation/mpc_6dof.hpp:91-404,
src/mpc_6dof.cpp) on top of the MI355X engine: same method names, argument meaning, bool returns
and call-order contract; Eigen vectors become numpy arrays.  Every solve runs on the GPU through
libtg_mpc.so (B = 1)."""
import numpy as np

from . import _capi as capi
from .engine import BatchedMPC6, make_opts, mpc6_params

NUM_OF_STATES = 6   # mpc_6dof.hpp:77
NUM_OF_INPUTS = 3   # mpc_6dof.hpp:78


def _err(msg):
    print("\033[1;31m[ERROR] %s\033[0m" % msg, flush=True)


def _info(msg):
    print("\033[1;32m[INFO] %s\033[0m" % msg, flush=True)


def _warn(msg):
    print("\033[1;33m[INFO] %s\033[0m" % msg, flush=True)


class MPC:
    def __init__(self, device=0, solver_opts=None):
        # constructor defaults of the reference, mpc_6dof.cpp:36-52
        self._debug = True
        self._dt = 0.05
        self._state_received = False
        self._alt_above_target = 1.0
        self._mpcWindow = 20
        self._state_weight = 1.0
        self._input_weight = 0.1
        self._smooth_input_weight = 10.0
        self._enable_control_smoothing = True
        self._minAltitude = 1.0
        self._is_MPC_initialized = False
        self._maxVel = None
        self._maxAccel = None
        self._current_accel = np.zeros(3)
        self._outputCSVFile = ""
        self._device = device
        self._solver_opts = solver_opts
        self._engine = None
        self._current_state = np.zeros(6)
        self._referenceTraj = np.zeros(0)
        self._optimal_state_traj = np.zeros(0)
        self._optimal_control_traj = np.zeros(0)
        self._gradient = np.zeros(0)
        self._lowerBounds = np.zeros(0)
        self._upperBounds = np.zeros(0)
        self.last_status = capi.TG_UNSOLVED
        self.last_iters = 0

    # ------------------------------------------------------------ setters (mpc_6dof.cpp:60-227)
    def setDt(self, dt):
        if dt > 0:
            self._dt = float(dt)
            return True
        _err("[MPC::setDt] dt < 0. Defaulting to 0.1")
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

    def setStateWeight(self, w):
        if w < 0.0:
            _err("state_weight =%f < 0" % w)
            return False
        self._state_weight = float(w)
        return True

    def setInputWeight(self, w):
        if w < 0.0:
            _err("_input_weight = %f < 0" % w)
            return False
        self._input_weight = float(w)
        return True

    def setSmoothInputWeight(self, w):
        if w < 0.0:
            _err("_smooth_input_weight = %f < 0" % w)
            return False
        self._smooth_input_weight = float(w)
        return True

    def enableControlSmoothing(self, b):
        self._enable_control_smoothing = bool(b)

    def setAltAboveTarget(self, h):
        if h < 0:
            _err("_alt_above_target = %f < 0" % h)
            return False
        self._alt_above_target = float(h)
        return True

    def setMinimumAltitude(self, h):
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

    def setCurrentAccel(self, a):
        a = np.asarray(a, dtype=np.float64).ravel()
        if a.size != 3:
            _err("Inpur accelration vector size %d !=3" % a.size)
            return False
        self._current_accel = a.copy()
        return True

    def setMaxVel(self, v):
        if len(v) != 3:
            _err("maxVel vector size %d != 3" % len(v))
            return False
        self._maxVel = np.array(v, dtype=np.float64)
        return True

    def setMaxAccel(self, a):
        if len(a) != 3:
            _err("maxAccel vector size %d != 3" % len(a))
            return False
        self._maxAccel = np.array(a, dtype=np.float64)
        return True

    def setReferenceTraj(self, v):
        # The reference compares the size of its own member, not of the argument
        # (mpc_6dof.cpp:796); the argument is what is checked here.
        v = np.asarray(v, dtype=np.float64).ravel()
        want = NUM_OF_STATES * (self._mpcWindow + 1)
        if v.size != want:
            _err("[MPC::set_referenceTraj] input matrix size %d != %d" % (v.size, want))
            return False
        self._referenceTraj = v.copy()
        return True

    def setOutputFilePath(self, path):
        self._outputCSVFile = str(path)

    # ------------------------------------------------------------ initMPCProblem (mpc_6dof.cpp:552)
    def initMPCProblem(self):
        if self._maxVel is None or self._maxAccel is None:
            _err("[MPC::initMPCProblem] setMaxVel / setMaxAccel must be called first")
            return False
        p = mpc6_params(N=self._mpcWindow, dt=self._dt, state_weight=self._state_weight,
                        input_weight=self._input_weight, smooth_input_weight=self._smooth_input_weight,
                        enable_control_smoothing=self._enable_control_smoothing,
                        min_altitude=self._minAltitude, max_vel=self._maxVel, max_accel=self._maxAccel)
        try:
            self._engine = BatchedMPC6(p, self._solver_opts or make_opts(), self._device)
        except capi.TgError as e:
            _err("[MPC::initMPCProblem] MPC initialization is not successful. %s" % e)
            return False
        self._current_state = np.zeros(6)
        self._referenceTraj = np.zeros(NUM_OF_STATES * (self._mpcWindow + 1))
        self._refresh_vectors()
        self._is_MPC_initialized = True
        return True

    def _refresh_vectors(self):
        """castMPCToQPGradient / castMPCToQPConstraintBounds via the assembly kernel."""
        e = self._engine
        buf = e.alloc(1)
        import torch
        buf["x0"][:, 0] = torch.from_numpy(self._current_state).to(buf["x0"].device)
        buf["ref"][:, 0] = torch.from_numpy(self._referenceTraj).to(buf["ref"].device)
        q, l, u = e.assemble(buf)
        self._gradient = q[:, 0].cpu().numpy()
        self._lowerBounds = l[:, 0].cpu().numpy()
        self._upperBounds = u[:, 0].cpu().numpy()

    # ------------------------------------------------------------ updateMPC / mpcLoop
    def updateMPC(self):
        if not self._is_MPC_initialized:
            return False
        self._refresh_vectors()   # updateQP, mpc_6dof.cpp:584
        return True

    def mpcLoop(self):
        if not self._is_MPC_initialized:
            _warn("[MPC::mpcLoop] MPC controller is not initialized. Skipping MPC loop.")
            return False
        if not self._state_received:
            _warn("[MPC::mpcLoop] Drone state is not received yet. Skipping MPC loop.")
            return False
        self._refresh_vectors()
        r = self._engine.solve_host(self._current_state[None, :], self._referenceTraj[None, :], warm=True)
        self.last_status, self.last_iters = int(r["status"][0]), int(r["iters"][0])
        if self.last_status != capi.TG_SOLVED:
            _err("[MPC::mpcLoop] MPC solution is not found")
            return False
        self._optimal_state_traj = r["X"][0].copy()
        self._optimal_control_traj = r["U"][0].copy()
        return True

    # ------------------------------------------------------------ getters (mpc_6dof.cpp:805-885)
    def getOptimalStateTraj(self):
        return self._optimal_state_traj.copy()

    def getOptimalControlTraj(self):
        return self._optimal_control_traj.copy()

    def getNumOfStates(self):
        return NUM_OF_STATES

    def getNumOfInputs(self):
        return NUM_OF_INPUTS

    def getMaxAccel(self):
        return self._maxAccel.copy()

    def getMaxVel(self):
        return self._maxVel.copy()

    def getTransitionMatrix(self):
        return self._engine.get(capi.TG_MAT_AD)

    def getInputMatrix(self):
        return self._engine.get(capi.TG_MAT_BD)

    def getGradient(self):
        return self._gradient.copy()

    def getLowerBounds(self):
        return self._lowerBounds.copy()

    def getUpperBounds(self):
        return self._upperBounds.copy()

    def getContraintsMatrix(self):
        return self._engine.get(capi.TG_MAT_A)

    def getHessianMatrix(self):
        return self._engine.get(capi.TG_MAT_P)

    def getQ(self):
        return self._engine.get(capi.TG_MAT_Q)

    def getR(self):
        return self._engine.get(capi.TG_MAT_R)

    # ------------------------------------------------------------ saveMPCDataToFile (mpc_6dof.cpp:739)
    def saveMPCDataToFile(self):
        sep = "\n------------------------------------------\n"

        def mat(a):
            a = np.atleast_2d(a)
            return "\n".join("[" + ", ".join(repr(float(v)) for v in row) + "]" for row in a)

        def col(a):
            return "\n".join("[" + repr(float(v)) + "]" for v in np.ravel(a))

        try:
            with open(self._outputCSVFile, "w") as f:
                f.write("Initial state, x(0): \n" + "\n".join(repr(float(v)) for v in self._current_state) + sep)
                f.write(" A : \n" + mat(self.getTransitionMatrix()) + sep)
                f.write("B : \n" + mat(self.getInputMatrix()) + sep)
                f.write("Q : \n" + mat(self.getQ()) + sep)
                f.write("R : \n" + mat(self.getR()) + sep)
                f.write("Hessian matrix, P: \n" + mat(self.getHessianMatrix()) + sep)
                f.write("Constarints matrix, Ac: \n" + mat(self.getContraintsMatrix()) + sep)
                f.write("Lower bounds, l: \n" + col(self._lowerBounds) + sep)
                f.write("Upper bounds, l: \n" + col(self._upperBounds) + sep)
                f.write("gradient, q: \n" + col(self._gradient) + sep)
                f.write("Optimal state trajectory, X: \n" + col(self._optimal_state_traj) + sep)
                f.write("Optimal control trajectory, U: \n" + col(self._optimal_control_traj) + sep)
            _info("[MPC::saveMPCDataToFile] Saved MPC solutions to file: %s" % self._outputCSVFile)
        except OSError:
            _err("[MPCROS::saveMPCDataToFile] Coudl not open file %s" % self._outputCSVFile)
