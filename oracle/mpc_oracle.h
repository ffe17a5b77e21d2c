/*
 * CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Nothing in the shipped product path may
 * link, import or call this code; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py do.
 *
 * PARITY UNPINNED: the reference (mzahana/trajectory_generation) delegates the QP
 * solve to OSQP (git commit 25b6b39) through osqp-eigen v0.8.0, neither of which is
 * vendored in the reference tree nor installed in this image, and the reference's
 * own tests (test/test_mpc_6dof.cpp, test/test_mpc_12state.cpp) hold no assertions
 * and no golden vectors.  This oracle therefore restates
 *   (1) the horizon-QP construction exactly as written in the reference
 *       (src/mpc_6dof.cpp:243-519, src/mpc_12state.cpp:74-1014), and
 *   (2) the published OSQP algorithm (Stellato et al., "OSQP: an operator splitting
 *       solver for quadratic programs", Math. Prog. Comp. 12, 2020, Algorithm 1 and
 *       sections 3.4, 5.1, 5.2) with the solver's documented default settings,
 * and is anchored by solver-independent optimality certificates (KKT residuals of a
 * primal-dual active-set refinement, see oracle/kkt_certificate.py).
 */
#ifndef MPC_ORACLE_H
#define MPC_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_INFTY 1e30 /* OsqpEigen::INFTY == OSQP_INFTY (mpc_6dof.cpp:306-308) */

/* ------------------------------------------------------------------ OSQP restatement */

typedef struct {
    double rho;                    /* 0.1   */
    double sigma;                  /* 1e-6  */
    double alpha;                  /* 1.6   */
    double eps_abs;                /* 1e-3  */
    double eps_rel;                /* 1e-3  */
    double eps_prim_inf;           /* 1e-4  */
    double eps_dual_inf;           /* 1e-4  */
    int    max_iter;               /* 4000  */
    int    check_termination;      /* 25    */
    int    scaling;                /* 10 Ruiz passes; 0 disables */
    int    adaptive_rho;           /* 1     */
    int    adaptive_rho_interval;  /* OSQP 0.6 default 0 = wall-clock based; the oracle
                                      needs a deterministic value and uses 50 when 0 is
                                      passed (the fixed default of OSQP >= 1.0) */
    double adaptive_rho_tolerance; /* 5     */
    int    warm_start;             /* reference: 1 for MPC (mpc_6dof.cpp:528), 0 for
                                      MPC12STATE (mpc_12state.cpp:1025,1045,1065) */
    int    scaled_termination;     /* 0     */
    int    check_fine;             /* NOT an OSQP setting: > 0 = after the first check (at
                                      check_termination) test every check_fine iterations, so the
                                      checker can follow the engine's check schedule; 0 = OSQP's
                                      fixed interval */
    int    linsys;                 /* NOT an OSQP setting: 0 = dense Cholesky of the reduced matrix
                                      P + sigma I + A' rho A (what the parity tests compare the
                                      engine with), 1 = sparse LDL' of the quasi-definite KKT
                                      matrix, the form of OSQP's default linear-system solver and
                                      of the CPU baseline timing */
} osqp_settings;

enum {
    ORACLE_SOLVED = 1,
    ORACLE_MAX_ITER = -2,
    ORACLE_PRIMAL_INFEASIBLE = -3,
    ORACLE_DUAL_INFEASIBLE = -4,
    ORACLE_UNSOLVED = -10
};

typedef struct osqp_work osqp_work;

void osqp_default_settings(osqp_settings *s);
/* the configuration the MI355X engine runs: scaling = 0, adaptive_rho = 0, rho given */
void osqp_fixed_rho_settings(osqp_settings *s, double rho);

/* P: n x n dense row-major (symmetric, full), A: m x n dense row-major.  Dense input
 * mirrors the reference, which builds dense Eigen matrices and calls sparseView()
 * (mpc_6dof.cpp:532-533). */
osqp_work *osqp_setup(int n, int m, const double *P, const double *A, const double *q,
                      const double *l, const double *u, const osqp_settings *s);
void osqp_cleanup(osqp_work *w);
int  osqp_update_bounds(osqp_work *w, const double *l, const double *u); /* 0 ok */
int  osqp_update_lin_cost(osqp_work *w, const double *q);
/* Extension used by the engine's fixed-rho mode: rho of inequality row i becomes
 * scale[i] * rho (OSQP itself only distinguishes equality / inequality / unbounded rows). */
int  osqp_set_rho_scale(osqp_work *w, const double *scale);
/* Step-size schedule of the engine's fixed-rho mode (tg_solver_opts::phase_iters / phase_scale):
 * nphase <= 3 phases; phase p >= 1 takes over at the first termination check at or after
 * iteration start[p] for an instance still running, with per-row multipliers scale[p * m + i] on
 * the inequality rows; x, z, y carry over.  nphase <= 1 switches the schedule off. */
int  osqp_set_rho_phases(osqp_work *w, int nphase, const int *start, const double *scale);
void osqp_warm_start(osqp_work *w, const double *x, const double *y);
int  osqp_solve(osqp_work *w); /* returns status */
void osqp_get_solution(const osqp_work *w, double *x, double *y);
int  osqp_get_iter(const osqp_work *w);
int  osqp_get_rho_updates(const osqp_work *w);
double osqp_get_rho(const osqp_work *w);
void osqp_get_residuals(const osqp_work *w, double *pri, double *dua);
void osqp_get_scaling(const osqp_work *w, double *D, double *E, double *c);
void osqp_get_rho_vec(const osqp_work *w, double *rho_vec);

/* ------------------------------------------------------------------ QP construction */

typedef struct {
    int    N;              /* _mpcWindow */
    double dt;
    double state_weight, input_weight, smooth_input_weight;
    int    enable_control_smoothing;
    double min_altitude;
    double max_vel[3], max_accel[3];
} mpc6_params;

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
} mpc12_params;

enum { FAM_6DOF = 0, FAM_XY = 1, FAM_Z = 2, FAM_YAW = 3 };

/* sizes of the QP of one family at horizon N */
void qp_dims(int family, int N, int *n, int *m, int *nx, int *nu);

/* Dense P (n*n), A (m*n) and the init-time q, l, u (x0 = 0, ref = 0) */
void qp6_build(const mpc6_params *p, double *P, double *A, double *q, double *l, double *u);
void qp12_build(const mpc12_params *p, int family, double *P, double *A, double *q,
                double *l, double *u);

/* model matrices, row-major */
void model6(const mpc6_params *p, double *Ad /*6x6*/, double *Bd /*6x3*/, double *Q, double *R);
void model12(const mpc12_params *p, int family, double *Ad, double *Bd, double *Q, double *R);

/* per-step updates */
void qp6_update(const mpc6_params *p, const double *x0 /*6*/, const double *ref /*6(N+1)*/,
                double *q, double *l, double *u);
void qp12_split_state(const double *x12, double *xy0, double *z0, double *yaw0);
void qp12_split_ref(int N, const double *ref12, double *xyref, double *zref, double *yawref);
void qp12_update_z(const mpc12_params *p, const double *z0, const double *zref, double *q,
                   double *l, double *u);
void qp12_xy_bounds_from_z(const mpc12_params *p, const double *z_x_opt /*3(N+1)*/,
                           double *xy_max /*6N*/, double *mixed_max /*6N*/);
void qp12_update_xy(const mpc12_params *p, const double *xy0, const double *xyref,
                    const double *xy_max, const double *mixed_max, double *q, double *l,
                    double *u);
void qp12_yaw_ref_from_xy(int N, const double *xyref, const double *xy_x_opt, double *yawref);
void qp12_update_yaw(const mpc12_params *p, const double *yaw0, const double *yawref,
                     double *q, double *l, double *u);
void qp12_merge(int N, const double *xy_sol, const double *z_sol, const double *yaw_sol,
                double *x_opt /*12(N+1)*/, double *u_opt /*4N*/);

/* Per-class rho multipliers applied by mpc6_create / mpc12_create to the box rows of
 * positions, velocities, accelerations (12-state) and inputs, and to the XY mixed rows
 * (velocity class for rows 0-3, acceleration class for rows 4-5).  Default {1,1,1,1}. */
void mpc_set_rho_class_scale(double pos, double vel, double acc, double input);
/* the same for one family (FAM_*) only: the stages of the 12-state chain may differ */
void mpc_set_rho_class_scale_family(int family, double pos, double vel, double acc, double input);
/* Step-size schedule for the solvers mpc6_create / mpc12_create make afterwards: nphase (1 .. 3)
 * phases for one family; start[p] and the class multipliers cls[p][4] (absolute, like
 * mpc_set_rho_class_scale) of phases 1 .. nphase - 1; phase 0 uses the family's class scale. */
void mpc_set_rho_phases_family(int family, int nphase, const int *start, const double *cls);

/* ------------------------------------------------------------------ whole mpcLoop   */

typedef struct mpc6_oracle mpc6_oracle;
mpc6_oracle *mpc6_create(const mpc6_params *p, const osqp_settings *s);
void mpc6_destroy(mpc6_oracle *o);
/* one MPC::mpcLoop (mpc_6dof.cpp:616-649): returns 1 when OSQP reports "solved" */
int mpc6_loop(mpc6_oracle *o, const double *x0, const double *ref, double *X /*6(N+1)*/,
              double *U /*3N*/, int *iters, int *status);
void mpc6_cold_start(mpc6_oracle *o);

typedef struct mpc12_oracle mpc12_oracle;
mpc12_oracle *mpc12_create(const mpc12_params *p, const osqp_settings *s);
/* per-stage solver settings (the engine picks one fixed rho per stage) */
mpc12_oracle *mpc12_create3(const mpc12_params *p, const osqp_settings *s_xy,
                            const osqp_settings *s_z, const osqp_settings *s_yaw);
void mpc12_destroy(mpc12_oracle *o);
/* one MPC12STATE::mpcLoop (mpc_12state.cpp:1374-1453).  iters/status: [z, xy, yaw].
 * inter (optional, may be NULL) receives the intermediates
 *   [ z_sol (4N+3) | xy_max (6N) | mixed_max (6N) | xy_sol (8N+6) | yaw_ref (3(N+1)) | yaw_sol (4N+3) ] */
int mpc12_loop(mpc12_oracle *o, const double *x12, const double *ref12, double *X, double *U,
               int *iters, int *status, double *inter);

/* batch helpers used by the CPU baseline timing (OpenMP-free, one instance at a time) */
int mpc6_batch(const mpc6_params *p, const osqp_settings *s, int B, const double *x0 /*B x 6*/,
               const double *ref /*B x 6(N+1)*/, double *X, double *U, int *iters, int *status);
int mpc12_batch(const mpc12_params *p, const osqp_settings *s, int B, const double *x12,
                const double *ref12, double *X, double *U, int *iters, int *status);

int mpc12_batch3(const mpc12_params *p, const osqp_settings *s_xy, const osqp_settings *s_z,
                 const osqp_settings *s_yaw, int B, const double *x12, const double *ref12,
                 double *X, double *U, int *iters, int *status);

/* ------------------------------------------------------------------ synthetic inputs */
/* counter-based generator shared with the Python side (splitmix64 of seed, stream, ctr) */
double synth_uniform(unsigned long long seed, unsigned long long stream,
                     unsigned long long ctr);
void synth_mpc6(unsigned long long seed, int N, double dt, int B, int first, double *x0,
                double *ref);
void synth_mpc12(unsigned long long seed, int N, double dt, int B, int first, double *x12,
                 double *ref12);

#ifdef __cplusplus
}
#endif
#endif
