/*
 * tg_mpc.h -- C ABI of the MI355X batched MPC engine (libtg_mpc.so).
 *
 * The reference (mzahana/trajectory_generation) has no FFI layer: its hot path sits
 * behind the public methods of two C++ classes,
 *     class MPC         include/trajectory_generation/mpc_6dof.hpp:282-404
 *     class MPC12STATE  include/trajectory_generation/mpc_12state.hpp:492-683
 * which build a horizon QP with Eigen and hand it to OSQP through osqp-eigen.  The
 * entry points below are what a binding of those two classes to this engine needs;
 * each one names the reference method(s) it replaces.  Plain C types only; every
 * buffer passed in is owned by the caller and is never freed by the library.
 *
 * Conventions
 *   - return value: 0 = success, negative = TG_E_* error (tg_last_error() has text)
 *   - "device" pointers are HIP device pointers on the handle's GPU
 *   - batched device arrays are structure-of-arrays, instance-minor:
 *         element (row r, instance b) lives at  ptr[r * ld + b],  ld >= B
 *     so that one wavefront reads 64 consecutive instances of one row
 *   - host-side batched arrays are instance-major (one reference-style vector per
 *     instance, back to back), exactly the vectors the reference's methods take
 *   - status per instance: 1 solved, -2 max_iter reached, -3 primal infeasible detected, and
 *     OSQP's two verdicts of its last test at max_iter with tenfold tolerances: 2 solved
 *     inaccurate, 3 primal infeasible inaccurate (OSQP status numbering; osqp-eigen's solve(),
 *     hence mpcLoop(), is true for 1 only)
 */
#ifndef TG_MPC_H
#define TG_MPC_H

#ifdef __cplusplus
extern "C" {
#endif

#define TG_INFTY 1e30 /* OsqpEigen::INFTY, mpc_6dof.cpp:306-308 */

enum { TG_OK = 0, TG_E_ARG = -1, TG_E_HIP = -2, TG_E_STATE = -3, TG_E_NUMERIC = -4 };
enum { TG_SOLVED = 1, TG_SOLVED_INACCURATE = 2, TG_PRIMAL_INFEASIBLE_INACCURATE = 3, TG_MAX_ITER = -2,
       TG_PRIMAL_INFEASIBLE = -3, TG_UNSOLVED = -10 };
enum { TG_FAM_6DOF = 0, TG_FAM_XY = 1, TG_FAM_Z = 2, TG_FAM_YAW = 3 };
enum { TG_MAT_P = 0, TG_MAT_A = 1, TG_MAT_AD = 2, TG_MAT_BD = 3, TG_MAT_Q = 4, TG_MAT_R = 5,
       TG_MAT_KINV = 6, TG_VEC_L = 7, TG_VEC_U = 8, TG_VEC_RHO = 9 };

/* MPC setters: setDt, setMPCWindow, setStateWeight, setInputWeight, setSmoothInputWeight,
 * enableControlSmoothing, setMinimumAltitude, setMaxVel, setMaxAccel
 * (mpc_6dof.cpp:60-227).  Frozen by tg_mpc6_create, as by MPC::initMPCProblem. */
typedef struct {
    int    N;
    double dt;
    double state_weight, input_weight, smooth_input_weight;
    int    enable_control_smoothing;
    double min_altitude;
    double max_vel[3], max_accel[3];
} tg_mpc6_params;

/* MPC12STATE setters (mpc_12state.cpp:1571-1932); triples are per stage */
typedef struct {
    int    N;
    double dt;
    int    enable_control_smoothing;
    double xy_state_weight, z_state_weight, yaw_state_weight;
    double xy_input_weight, z_input_weight, yaw_input_weight;
    double xy_smooth_input_weight, z_smooth_input_weight, yaw_smooth_input_weight;
    double xy_max_vel, z_max_vel, yaw_max_vel;
    double xy_max_accel, z_max_accel, yaw_max_accel;
    double xy_max_jerk, z_max_jerk, yaw_max_jerk;
} tg_mpc12_params;

/* Solver options.  The reference leaves every OSQP setting at its default except
 * warm start (mpc_6dof.cpp:527-528, mpc_12state.cpp:1024-1065); the engine runs the
 * same ADMM with OSQP's scaling and rho adaptation switched off and a fixed rho.
 * rho <= 0 selects the built-in per-family choice. */
typedef struct {
    double rho;               /* <=0: automatic */
    double sigma;             /* 1e-6 */
    double alpha;             /* 1.6  */
    double eps_abs, eps_rel;  /* 1e-3 */
    int    max_iter;          /* 4000 */
    int    check_termination; /* 25   */
    /* multipliers of rho on the box rows of positions, velocities, accelerations (12-state)
     * and inputs; the XY mixed rows use the velocity / acceleration entries.  OSQP has one rho
     * for all inequality rows and adapts it at run time; a fixed rho per class of row takes
     * the place of that adaptation (together with the schedule below).  0 = the built-in choice,
     * 1 = OSQP's uniform rho. */
    double rho_scale[4];
    /* OSQP's eps_prim_inf (1e-4): tolerance of the primal infeasibility certificate tested at
     * every termination check; 0 switches the test off.  Dual infeasibility cannot occur: the
     * inputs are boxed and the dynamics rows tie every state to them. */
    double eps_prim_inf;
    /* > 0: after the first termination check (at iteration check_termination) test every
     * check_fine iterations instead of every check_termination.  In a batch a tile of 16
     * instances runs until its slowest member passes the test, and most instances pass it a few
     * iterations after the first check; OSQP's fixed interval would make them all wait for the
     * next multiple of 25.  Every returned iterate passes the same eps_abs / eps_rel test.
     * Earlier stops mean iterates that sit closer to the tolerance.  Default 0 = OSQP's fixed
     * interval. */
    int    check_fine;
    /* Largest number of workgroups (tiles of 16 instance slots) a solve launches.  0: as many as
     * the GPU holds at once.  When the batch has more tiles than that, a slot whose instance has
     * terminated takes the next instance of the batch, so no slot waits for the slowest member of
     * its tile; every instance keeps its own iteration count and check schedule, and its result
     * does not depend on the batch around it.  Needs max_iter to be a multiple of
     * check_termination and check_termination a multiple of check_fine (at most 32 times it);
     * otherwise, or with a negative value, every tile gets a workgroup of its own. */
    int    grid_cap;
    /* How K^-1 is applied.  K is block tridiagonal, so outside its 16 x 16 diagonal tiles a tile
     * row of K^-1 has rank <= 2 x (integrator order [+ 1 with input smoothing]); the engine can
     * apply it as diagonal tile + low-rank rest, with less than half the matrix-core work of the
     * dense product on long horizons.  0: where it pays (horizons too long for register-resident
     * operators), 1: wherever the computed inverse has the structure to working precision,
     * -1: dense always.  The two forms agree to rounding (1e-15 relative). */
    int    lowrank;
    /* Order in which a grid whose slots are refilled (grid_cap above) takes the instances of the
     * batch.  1: the sixteenth of the batch whose reference strays farthest from where the initial
     * state coasts to is taken first -- those instances tend to need the most iterations, and
     * started last they would leave the GPU waiting for them; -1: batch order; 0: where it pays:
     * horizons longer than 30, on handles without a step-size schedule, and with one (phase_iters
     * below: an instance then leaves the first launch after phase_iters[0] iterations wherever it
     * started) for batches of at most 49152 instances.  An instance's result does not depend on
     * it. */
    int    hard_first;
    /* Step-size schedule, in place of OSQP's run-time rho adaptation.  A handle holds up to three
     * factorised operators; an instance that is still running at the first termination check at
     * or after iteration phase_iters[p] (p = 0, 1) continues under the next one, its iterates
     * (x, z, y) carried over -- any rho is a valid ADMM step, which is what OSQP's own adaptation
     * relies on.  Decided per instance, at the check; instances that move are gathered and solved
     * together, so the operator stays shared.  phase_scale[p] multiplies the first operator's
     * rho_scale for phase p + 1.  phase_iters: 0 = the built-in choice (100 and 200 iterations;
     * 6-DOF with N > 30: the step on the position rows x 8 and x 64; shorter horizons: one later
     * operator, position rows x 64 from iteration 150; 12-state Z and yaw stages: on the speed
     * rows x 10 and x 100, XY stage: on the acceleration rows x 3 and x 10),
     * < 0 = no such phase; phase_scale entries of 0 = built-in.  Frozen at create (the operators
     * are built there); tg_*_set_opts ignores both.  The entry point that allocates nothing runs
     * the first operator only. */
    int    phase_iters[2];
    double phase_scale[2][4];
    /* 12-state chain: 1 = when the XY stage has a step-size schedule, the yaw stage of the vehicles
     * whose XY problem the first XY launch solved runs on a second stream of the handle beside the
     * XY stage's continuation launches, and the vehicles those solve get their yaw stage
     * afterwards; 0 (default) = the three stages strictly one after the other.  Results are the
     * same bit for bit; the work is complete when the caller's stream is.  It shortens a batch
     * that runs alone (4096 vehicles, N = 20: 1.95 against 2.15 ms) and costs throughput when
     * several handles keep batches in flight (their streams then outnumber the hardware queues).
     * Frozen at create; ignored by the 6-DOF handles and by the entry point that allocates
     * nothing. */
    int    overlap;
} tg_solver_opts;

void tg_default_opts(tg_solver_opts *o);
const char *tg_last_error(void);
int tg_device_count(void);

/* ------------------------------------------------------------------ 6-DOF (class MPC) */
typedef struct tg_mpc6 tg_mpc6;

/* MPC::MPC + MPC::initMPCProblem (mpc_6dof.cpp:552-580): builds P, A, bounds, the
 * shared reduced-KKT inverse, and uploads them to GPU `device`. */
int tg_mpc6_create(const tg_mpc6_params *p, const tg_solver_opts *o, int device, tg_mpc6 **out);
void tg_mpc6_destroy(tg_mpc6 *h);

/* B independent MPC::mpcLoop calls (mpc_6dof.cpp:616-649): updateQP + solve +
 * extractSolution.  Device SoA arrays:
 *   x0  [6][ld]          setCurrentState   mpc_6dof.cpp:175
 *   ref [6(N+1)][ld]     setReferenceTraj  mpc_6dof.cpp:794
 *   X   [6(N+1)][ld]     getOptimalStateTraj   :805
 *   U   [3N][ld]         getOptimalControlTraj :811
 *   status[B], iters[B]  (may be NULL)
 * X and U of an instance are written only when its status is TG_SOLVED: mpcLoop returns
 * before extractSolution when solve() is false (mpc_6dof.cpp:639-647), so the columns of a
 * failed instance keep what they held (the plan of its last successful solve, if the
 * caller reuses the arrays; the host variants below zero them first).
 * warm != 0 continues from the iterates the handle keeps for slots [0,B) (the
 * reference's setWarmStart(true), mpc_6dof.cpp:528) and stores the new ones; the
 * first warm call of a handle starts cold.  stream: hipStream_t or NULL. */
int tg_mpc6_solve_batch(tg_mpc6 *h, int B, int ld, const double *d_x0, const double *d_ref,
                        double *d_X, double *d_U, int *d_status, int *d_iters, int warm,
                        void *stream);

/* same on host vectors, instance-major (x0[B][6], ref[B][6(N+1)], X[B][6(N+1)], U[B][3N]) */
int tg_mpc6_solve_host(tg_mpc6 *h, int B, const double *x0, const double *ref, double *X,
                       double *U, int *status, int *iters, int warm);

/* castMPCToQPGradient / castMPCToQPConstraintBounds for a batch (mpc_6dof.cpp:371,450):
 * the full per-instance q [n][ld], l [m][ld], u [m][ld] in the reference's ordering. */
int tg_mpc6_assemble_batch(tg_mpc6 *h, int B, int ld, const double *d_x0, const double *d_ref,
                           double *d_q, double *d_l, double *d_u, void *stream);

/* same on host vectors, instance-major (q[B][n], l[B][m], u[B][m]; any output may be NULL):
 * what getGradient / getLowerBounds / getUpperBounds (mpc_6dof.cpp:859-873) return after
 * updateQP for the given x0 and reference */
int tg_mpc6_assemble_host(tg_mpc6 *h, int B, const double *x0, const double *ref, double *q,
                          double *l, double *u);

/* drop the stored warm-start iterates */
int tg_mpc6_reset_warm(tg_mpc6 *h);
/* getters of the reference (getHessianMatrix, getContraintsMatrix, getTransitionMatrix,
 * getInputMatrix, getQ, getR, getLowerBounds, getUpperBounds  mpc_6dof.cpp:841-885);
 * row-major, sizes from tg_mpc6_dims; bounds are the init-time ones (x0 = 0) */
int tg_mpc6_dims(const tg_mpc6 *h, int *n, int *m);
int tg_mpc6_get(const tg_mpc6 *h, int which, double *out);
double tg_mpc6_rho(const tg_mpc6 *h);
/* how the solve kernel applies K^-1: v_mfma_f64_16x16x4_f64 instructions per ADMM iteration of a
 * tile of 16 instances (2048 flop each), and whether the low-rank form is in use (no counterpart
 * in the reference; for roofline accounting) */
int tg_mpc6_kernel_info(const tg_mpc6 *h, int *mfma_per_tile_iteration, int *lowrank);
/* Solver settings of a live handle (what osqp-eigen's solver.settings() exposes,
 * mpc_6dof.cpp:527-528): get returns the effective values (automatic rho / rho_scale resolved,
 * lowrank = the operator form in use); set changes alpha, eps_*, max_iter, check_termination,
 * check_fine and grid_cap for the following solves.  rho, sigma and rho_scale enter the
 * factorised operator and are frozen at create: asking for different ones returns TG_E_STATE. */
int tg_mpc6_get_opts(const tg_mpc6 *h, tg_solver_opts *out);
int tg_mpc6_set_opts(tg_mpc6 *h, const tg_solver_opts *o);
/* Unscaled primal / dual residual (OSQP's info->pri_res, dua_res) of instances [0, B) at the
 * termination check that ended the handle's last solve; host arrays of B doubles, either may be
 * NULL.  Synchronises the device. */
int tg_mpc6_last_residuals(tg_mpc6 *h, int B, double *prim, double *dual);

/* ------------------------------------------------------------------ 12-state (class MPC12STATE) */
typedef struct tg_mpc12 tg_mpc12;

/* MPC12STATE::initMPCProblem (mpc_12state.cpp:1282-1350); opts apply to all three
 * stages, rho_xyz (may be NULL) overrides the per-stage automatic rho {xy, z, yaw} */
int tg_mpc12_create(const tg_mpc12_params *p, const tg_solver_opts *o, const double *rho_xyz,
                    int device, tg_mpc12 **out);
void tg_mpc12_destroy(tg_mpc12 *h);

/* B independent MPC12STATE::mpcLoop calls (mpc_12state.cpp:1374-1453):
 * Z solve -> computeXYBounds -> XY solve -> computeYawRefTrajectory -> Yaw solve -> merge.
 *   x0  [12][ld], ref [12(N+1)][ld], X [12(N+1)][ld], U [4N][ld]
 *   status[3][ld], iters[3][ld]   rows = {z, xy, yaw} stage   (may be NULL)
 * X and U of an instance are written only when all three stages end TG_SOLVED -- whole columns
 * or nothing: mpcLoop fills _x_opt / _u_opt in extractSolution, after the third stage
 * (mpc_12state.cpp:1405-1449, 1528-1551), so an instance whose XY or yaw stage fails keeps the
 * whole plan of its last successful call, Z rows included.
 * warm: as above (a deviation from the reference, which cold-starts all three
 * solvers, mpc_12state.cpp:1025,1045,1065; used by closed-loop roll-outs). */
int tg_mpc12_solve_batch(tg_mpc12 *h, int B, int ld, const double *d_x0, const double *d_ref,
                         double *d_X, double *d_U, int *d_status, int *d_iters, int warm,
                         void *stream);
int tg_mpc12_solve_host(tg_mpc12 *h, int B, const double *x0, const double *ref, double *X,
                        double *U, int *status, int *iters, int warm);

/* No counterpart in the reference: tg_mpc12_solve_batch for callers that cannot allocate device
 * memory at call time -- code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc.  Same three stages, same kernels and bit-identical X / U / status / iters on a handle
 * whose step-size schedule is off (phase_iters = {-1, -1}; with a schedule this entry point runs
 * every instance under the first operator to the end, since the hand-over between operators needs
 * per-instance storage it has no room for).  Cold start only: there is no warm argument, no
 * residual record and no kernel timing.  The arrays that
 * pass between the stages (xy_max, mixed_max, yaw_ref), the stages' solutions until all three have
 * solved, the stage status rows when d_status is NULL and the kernels' queue counters live in a
 * static device array of 32 MiB that every GPU the library runs on holds until the process ends,
 * instead of in buffers the handle allocates.  The call makes no hipMalloc / hipFree.  Limit:
 * tg_mpc12_static_scratch_bytes(N, ld) = 256 + roundup(12 ld, 256) + 8 (29 N + 13) ld must not
 * exceed 32 MiB (N = 20: ld <= 7055; N = 50: ld <= 2863); larger batches are refused with
 * TG_E_ARG.  All calls on one GPU share that GPU's array: two of them overlapping in time
 * (different streams or threads, any handles) race; calls on one stream, or on different GPUs, do
 * not.  tg_mpc12_intermediates does not see its arrays. */
int tg_mpc12_solve_batch_static(tg_mpc12 *h, int B, int ld, const double *d_x0, const double *d_ref,
                                double *d_X, double *d_U, int *d_status, int *d_iters, void *stream);
unsigned long long tg_mpc12_static_scratch_bytes(int N, int ld);

/* single stages with the upstream solution injected, for stage-by-stage checks:
 * stage = TG_FAM_Z / TG_FAM_XY / TG_FAM_YAW.  For XY, X must already hold the z rows
 * (6..8 of every step); for YAW, the xy rows (0..5). */
int tg_mpc12_solve_stage(tg_mpc12 *h, int stage, int B, int ld, const double *d_x0,
                         const double *d_ref, double *d_X, double *d_U, int *d_status,
                         int *d_iters, void *stream);
/* intermediate products of the last tg_mpc12_solve_* call on this handle (device, SoA with
 * the ld of that call): xy_max [6N], mixed_max [6N] (computeXYBounds :724), yaw_ref [N+1]
 * (computeYawRefTrajectory :525) */
int tg_mpc12_intermediates(tg_mpc12 *h, const double **d_xy_max, const double **d_mixed_max,
                           const double **d_yaw_ref, int *ld);
int tg_mpc12_assemble_stage(tg_mpc12 *h, int stage, int B, int ld, const double *d_x0,
                            const double *d_ref, const double *d_X, double *d_q, double *d_l,
                            double *d_u, void *stream);
/* host vectors, instance-major; X (the upstream solution, [B][12(N+1)]) may be NULL for Z */
int tg_mpc12_assemble_host(tg_mpc12 *h, int stage, int B, const double *x0, const double *ref,
                           const double *X, double *q, double *l, double *u);
int tg_mpc12_reset_warm(tg_mpc12 *h);
int tg_mpc12_dims(const tg_mpc12 *h, int stage, int *n, int *m);
int tg_mpc12_get(const tg_mpc12 *h, int stage, int which, double *out);
double tg_mpc12_rho(const tg_mpc12 *h, int stage);
int tg_mpc12_kernel_info(const tg_mpc12 *h, int stage, int *mfma_per_tile_iteration, int *lowrank);
/* as for the 6-DOF handle; get is per stage (their rho and rho_scale differ), set applies to all
 * three.  set ignores rho and rho_scale (per stage, frozen at create), so that what get returned
 * for any stage can be modified and handed back; only a different sigma is refused. */
int tg_mpc12_get_opts(const tg_mpc12 *h, int stage, tg_solver_opts *out);
int tg_mpc12_set_opts(tg_mpc12 *h, const tg_solver_opts *o);
int tg_mpc12_last_residuals(tg_mpc12 *h, int stage, int B, double *prim, double *dual);

/* ------------------------------------------------------------------ closed-loop roll-out */
/* Plant update with the first control of the horizon, x0 <- A x0 + B u(0), A and B being
 * getTransitionMatrix / getInputMatrix: the rule both test mains roll the solution out with
 * (test_mpc_6dof.cpp:115-122, test_mpc_12state.cpp:193-197; mpc_12state.cpp:2001-2025).
 * d_x0 [nx][ld] is updated in place, d_U is the U array of the last solve.
 * d_status: the status array of that solve ([ld], 12-state [3][ld]) or NULL.  An instance whose
 * solve did not reach TG_SOLVED in every stage got no new plan -- mpcLoop returned false and the
 * reference's node publishes nothing (mpc_ros.cpp:428-432, mpc_12state_ros.cpp:519-527) -- so it
 * keeps following the last plan it received: its U rows move one step up (zeros enter at the end:
 * beyond the plan it coasts) before row 0 is applied.  With NULL every instance counts as solved. */
int tg_mpc6_plant_step(tg_mpc6 *h, int B, int ld, double *d_x0, double *d_U, const int *d_status, void *stream);
int tg_mpc12_plant_step(tg_mpc12 *h, int B, int ld, double *d_x0, double *d_U, const int *d_status, void *stream);
/* Receding horizon on a reference array ref [nx(N+1)][ld]: block i+1 moves to block i, the last
 * block is continued linearly.  What mpc_ros.cpp / mpc_12state_ros.cpp get from their reference
 * topic at every step, produced on the device. */
int tg_ref_shift(int B, int ld, int nx, int N, double *d_ref, void *stream);

/* device -> host copy of `bytes` bytes on the current device (for callers without a HIP
 * runtime of their own, e.g. a ctypes binding reading tg_mpc12_intermediates) */
int tg_copy_to_host(void *dst, const void *d_src, unsigned long long bytes);

/* ------------------------------------------------------------------ kernel timing hook */
/* Total duration (ms) and number of the ADMM kernel launches made while timing was enabled,
 * since the last call with reset != 0, over all handles of the process; measured with HIP events
 * on each launch's stream.  Used by the benchmark's roofline line.  Thread-safe. */
void tg_timing_enable(int on);
int tg_timing_read(double *admm_ms_total, int *admm_launches, int reset);

#ifdef __cplusplus
}
#endif
#endif
