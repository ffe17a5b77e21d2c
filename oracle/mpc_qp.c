/*
 * CPU ORACLE -- TEST INFRASTRUCTURE ONLY (see mpc_oracle.h).  PARITY UNPINNED.
 *
 * Restatement of the horizon-QP construction of the reference:
 *   class MPC        src/mpc_6dof.cpp:243-519, 552-580, 616-718
 *   class MPC12STATE src/mpc_12state.cpp:74-1014, 1282-1350, 1374-1551, 1786-1814, 1935-1963
 * Variable order  [x(0); ...; x(N); u(0); ...; u(N-1)]
 * Row order       [dynamics nx(N+1) | state box nx(N+1) | input box nu N | (XY) mixed 6N]
 */
#include "mpc_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

void qp_dims(int family, int N, int *n, int *m, int *nx, int *nu)
{
    int sx, su, mixed = 0;
    switch (family) {
    case FAM_6DOF: sx = 6; su = 3; break;             /* mpc_6dof.hpp:77-78 */
    case FAM_XY:   sx = 6; su = 2; mixed = 6; break;  /* mpc_12state.hpp:102-105 */
    default:       sx = 3; su = 1; break;             /* mpc_12state.hpp:114-122 */
    }
    if (n) *n = sx * (N + 1) + su * N;
    if (m) *m = 2 * sx * (N + 1) + su * N + mixed * N;
    if (nx) *nx = sx;
    if (nu) *nu = su;
}

/* ---------------------------------------------------------------- models */
void model6(const mpc6_params *p, double *Ad, double *Bd, double *Q, double *R)
{
    /* setTransitionMatrix mpc_6dof.cpp:274, setInputMatrix :287, setQ :244, setR :260 */
    memset(Ad, 0, sizeof(double) * 36); memset(Bd, 0, sizeof(double) * 18);
    memset(Q, 0, sizeof(double) * 36);  memset(R, 0, sizeof(double) * 9);
    for (int i = 0; i < 6; i++) Ad[i * 6 + i] = 1.0;
    for (int i = 0; i < 3; i++) {
        Ad[i * 6 + 3 + i] = p->dt;          /* position integrates velocity; no dt^2/2 term */
        Bd[(3 + i) * 3 + i] = p->dt;
        Q[i * 6 + i] = p->state_weight;     /* positions only */
        R[i * 3 + i] = p->input_weight;
    }
}

static void triple_integrator(double dt, double *A3)
{
    /* set{XY,Z,Yaw}TransitionMatrix mpc_12state.cpp:161-200 */
    double a[9] = {1, dt, dt * dt / 2.0, 0, 1, dt, 0, 0, 1};
    memcpy(A3, a, sizeof(a));
}

void model12(const mpc12_params *p, int family, double *Ad, double *Bd, double *Q, double *R)
{
    double A3[9];
    triple_integrator(p->dt, A3);
    if (family == FAM_XY) {
        memset(Ad, 0, sizeof(double) * 36); memset(Bd, 0, sizeof(double) * 12);
        memset(Q, 0, sizeof(double) * 36);  memset(R, 0, sizeof(double) * 4);
        for (int b = 0; b < 2; b++)
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) Ad[(3 * b + i) * 6 + 3 * b + j] = A3[i * 3 + j];
        Bd[2 * 2 + 0] = p->dt; Bd[5 * 2 + 1] = p->dt;          /* :203-212 */
        Q[0] = p->xy_state_weight; Q[3 * 6 + 3] = p->xy_state_weight; /* :75-88 */
        R[0] = R[3] = p->xy_input_weight;                      /* :121-131 */
    } else {
        double w = family == FAM_Z ? p->z_state_weight : p->yaw_state_weight;
        double r = family == FAM_Z ? p->z_input_weight : p->yaw_input_weight;
        memcpy(Ad, A3, sizeof(A3));
        Bd[0] = 0; Bd[1] = 0; Bd[2] = p->dt;                   /* :215-236 */
        memset(Q, 0, sizeof(double) * 9); Q[0] = w;            /* :91-118 */
        R[0] = r;                                              /* :134-157 */
    }
}

/* ---------------------------------------------------------------- generic chain QP */
/* castMPCToQPHessian (mpc_6dof.cpp:329, mpc_12state.cpp:314/355/396) and
 * castMPCToQPConstraintMatrix (mpc_6dof.cpp:409, mpc_12state.cpp:571/641/683) */
static void chain_build(int nx, int nu, int N, const double *Ad, const double *Bd, const double *Q,
                        const double *R, int smoothing, double smooth_w, int mixed, double *P,
                        double *A)
{
    int Nx = nx * (N + 1), Nu = nu * N, n = Nx + Nu, m = 2 * Nx + Nu + mixed * N;
    memset(P, 0, sizeof(double) * (size_t)n * n);
    memset(A, 0, sizeof(double) * (size_t)m * n);
    for (int i = 0; i <= N; i++)
        for (int a = 0; a < nx; a++)
            for (int b = 0; b < nx; b++) P[(long)(nx * i + a) * n + nx * i + b] = Q[a * nx + b];
    for (int i = 0; i < N; i++)
        for (int a = 0; a < nu; a++)
            for (int b = 0; b < nu; b++)
                P[(long)(Nx + nu * i + a) * n + Nx + nu * i + b] = R[a * nu + b];
    if (smoothing) {
        /* U_diff' * S * U_diff with U_diff the (N-1) x N block first-difference operator:
         * +s on both diagonal blocks touched by a difference, -s on the two off-diagonal */
        for (int i = 0; i < N - 1; i++)
            for (int a = 0; a < nu; a++) {
                long r0 = Nx + nu * i + a, r1 = Nx + nu * (i + 1) + a;
                P[r0 * n + r0] += smooth_w; P[r1 * n + r1] += smooth_w;
                P[r0 * n + r1] -= smooth_w; P[r1 * n + r0] -= smooth_w;
            }
    }
    for (int r = 0; r < Nx; r++) A[(long)r * n + r] = -1.0;
    for (int i = 1; i <= N; i++) {
        for (int a = 0; a < nx; a++) {
            for (int b = 0; b < nx; b++)
                A[(long)(nx * i + a) * n + nx * (i - 1) + b] = Ad[a * nx + b];
            for (int b = 0; b < nu; b++)
                A[(long)(nx * i + a) * n + Nx + nu * (i - 1) + b] = Bd[a * nu + b];
        }
    }
    for (int r = 0; r < Nx; r++) A[(long)(Nx + r) * n + r] = 1.0;
    for (int r = 0; r < Nu; r++) A[(long)(2 * Nx + r) * n + Nx + r] = 1.0;
    if (mixed) {
        /* mpc_12state.cpp:600-630 ; acts on state i+1, i = 0..N-1 */
        const double s3 = sqrt(3.0) / 2.0, s2 = sqrt(2.0) / 2.0;
        for (int i = 0; i < N; i++) {
            long r = 2 * Nx + Nu + 6 * i;
            long cvx = nx * (i + 1) + 1, cvy = nx * (i + 1) + 4;
            long cax = nx * (i + 1) + 2, cay = nx * (i + 1) + 5;
            A[(r + 0) * n + cvx] = s3;   A[(r + 0) * n + cvy] = 0.5;
            A[(r + 1) * n + cvx] = s3;   A[(r + 1) * n + cvy] = -0.5;
            A[(r + 2) * n + cvx] = 0.5;  A[(r + 2) * n + cvy] = s3;
            A[(r + 3) * n + cvx] = -0.5; A[(r + 3) * n + cvy] = s3;
            A[(r + 4) * n + cax] = s2;   A[(r + 4) * n + cay] = s2;
            A[(r + 5) * n + cax] = s2;   A[(r + 5) * n + cay] = -s2;
        }
    }
}

/* ---------------------------------------------------------------- 6-DOF */
static void bounds6(const mpc6_params *p, double *xmin, double *xmax, double *umin, double *umax)
{
    /* setStateBounds mpc_6dof.cpp:299, setControlBounds :320 */
    xmin[0] = -ORACLE_INFTY; xmax[0] = ORACLE_INFTY;
    xmin[1] = -ORACLE_INFTY; xmax[1] = ORACLE_INFTY;
    xmin[2] = p->min_altitude; xmax[2] = ORACLE_INFTY;
    for (int i = 0; i < 3; i++) {
        xmin[3 + i] = -p->max_vel[i]; xmax[3 + i] = p->max_vel[i];
        umin[i] = -p->max_accel[i];   umax[i] = p->max_accel[i];
    }
}

void qp6_update(const mpc6_params *p, const double *x0, const double *ref, double *q, double *l,
                double *u)
{
    int N = p->N, Nx = 6 * (N + 1), Nu = 3 * N;
    double xmin[6], xmax[6], umin[3], umax[3];
    bounds6(p, xmin, xmax, umin, umax);
    /* castMPCToQPConstraintBounds :450 / updateQPConstraintsBounds :505 */
    for (int r = 0; r < Nx; r++) l[r] = u[r] = 0.0;
    for (int a = 0; a < 6; a++) { l[a] = -x0[a]; u[a] = -x0[a]; }
    for (int i = 0; i <= N; i++)
        for (int a = 0; a < 6; a++) { l[Nx + 6 * i + a] = xmin[a]; u[Nx + 6 * i + a] = xmax[a]; }
    for (int i = 0; i < N; i++)
        for (int a = 0; a < 3; a++) {
            l[2 * Nx + 3 * i + a] = umin[a]; u[2 * Nx + 3 * i + a] = umax[a];
        }
    /* castMPCToQPGradient :371 / updateQPGradientVector :393 : q_i = -Q ref_i */
    memset(q, 0, sizeof(double) * (size_t)(Nx + Nu));
    for (int i = 0; i <= N; i++)
        for (int a = 0; a < 3; a++) q[6 * i + a] = -1.0 * p->state_weight * ref[6 * i + a];
}

void qp6_build(const mpc6_params *p, double *P, double *A, double *q, double *l, double *u)
{
    double Ad[36], Bd[18], Q[36], R[9];
    model6(p, Ad, Bd, Q, R);
    chain_build(6, 3, p->N, Ad, Bd, Q, R, p->enable_control_smoothing, p->smooth_input_weight, 0,
                P, A);
    /* initMPCProblem :560-566 : x0 = 0, ref = 0 */
    double x0[6] = {0};
    double *ref = (double *)calloc((size_t)6 * (p->N + 1), sizeof(double));
    qp6_update(p, x0, ref, q, l, u);
    free(ref);
}

/* ---------------------------------------------------------------- 12-state */
void qp12_split_state(const double *x12, double *xy0, double *z0, double *yaw0)
{
    /* setCurrentState mpc_12state.cpp:1786-1814 */
    memcpy(xy0, x12, sizeof(double) * 6);
    memcpy(z0, x12 + 6, sizeof(double) * 3);
    memcpy(yaw0, x12 + 9, sizeof(double) * 3);
}

void qp12_split_ref(int N, const double *ref12, double *xyref, double *zref, double *yawref)
{
    /* setReferenceTraj mpc_12state.cpp:1935-1963 */
    for (int i = 0; i <= N; i++) {
        memcpy(xyref + 6 * i, ref12 + 12 * i, sizeof(double) * 6);
        memcpy(zref + 3 * i, ref12 + 12 * i + 6, sizeof(double) * 3);
        memcpy(yawref + 3 * i, ref12 + 12 * i + 9, sizeof(double) * 3);
    }
}

static void single_axis_update(int N, double w, double vmax, double amax, double pmax, double jmax,
                               const double *s0, const double *ref, double *q, double *l, double *u)
{
    /* cast{Z,Yaw}MPCToQPConstraintBounds mpc_12state.cpp:839/889, update :983/1000,
     * set{Z,Yaw}StateBounds :260/273, set{Z,Yaw}ControlBounds :296/305 */
    int Nx = 3 * (N + 1), Nu = N;
    for (int r = 0; r < Nx; r++) l[r] = u[r] = 0.0;
    for (int a = 0; a < 3; a++) { l[a] = -s0[a]; u[a] = -s0[a]; }
    for (int i = 0; i <= N; i++) {
        l[Nx + 3 * i + 0] = -pmax; u[Nx + 3 * i + 0] = pmax;
        l[Nx + 3 * i + 1] = -vmax; u[Nx + 3 * i + 1] = vmax;
        l[Nx + 3 * i + 2] = -amax; u[Nx + 3 * i + 2] = amax;
    }
    for (int i = 0; i < N; i++) { l[2 * Nx + i] = -jmax; u[2 * Nx + i] = jmax; }
    memset(q, 0, sizeof(double) * (size_t)(Nx + Nu));
    for (int i = 0; i <= N; i++) q[3 * i] = -1.0 * w * ref[3 * i];
}

void qp12_update_z(const mpc12_params *p, const double *z0, const double *zref, double *q,
                   double *l, double *u)
{
    single_axis_update(p->N, p->z_state_weight, p->z_max_vel, p->z_max_accel, ORACLE_INFTY,
                       p->z_max_jerk, z0, zref, q, l, u);
}

void qp12_update_yaw(const mpc12_params *p, const double *yaw0, const double *yawref, double *q,
                     double *l, double *u)
{
    single_axis_update(p->N, p->yaw_state_weight, p->yaw_max_vel, p->yaw_max_accel, 2.0 * M_PI,
                       p->yaw_max_jerk, yaw0, yawref, q, l, u);
}

void qp12_xy_bounds_from_z(const mpc12_params *p, const double *z_x_opt, double *xy_max,
                           double *mixed_max)
{
    /* computeXYBounds mpc_12state.cpp:724-791 */
    int N = p->N;
    for (int i = 1; i <= N; i++) {
        double *b = xy_max + 6 * (i - 1), *mx = mixed_max + 6 * (i - 1);
        double vb;
        double v_zt = z_x_opt[3 * i + 1];
        if (v_zt < 0) {
            vb = p->xy_max_vel;
        } else {
            double a_zt = z_x_opt[3 * i + 2];
            double d = a_zt / p->z_max_accel;
            vb = p->xy_max_vel * 0.9;
            if ((1 - d * d) >= 0) vb = p->xy_max_vel * sqrt(1 - d * d);
        }
        b[0] = ORACLE_INFTY; b[1] = vb; b[2] = p->xy_max_accel;
        b[3] = ORACLE_INFTY; b[4] = vb; b[5] = p->xy_max_accel;
        mx[0] = mx[1] = mx[2] = mx[3] = vb;
        mx[4] = mx[5] = p->xy_max_accel;
    }
}

void qp12_update_xy(const mpc12_params *p, const double *xy0, const double *xyref,
                    const double *xy_max, const double *mixed_max, double *q, double *l, double *u)
{
    /* castXYMPCToQPConstraintBounds mpc_12state.cpp:794 / updateXYQPConstraintsBounds :939 */
    int N = p->N, Nx = 6 * (N + 1), Nu = 2 * N;
    for (int r = 0; r < Nx; r++) l[r] = u[r] = 0.0;
    for (int a = 0; a < 6; a++) { l[a] = -xy0[a]; u[a] = -xy0[a]; }
    double b0[6] = {ORACLE_INFTY, p->xy_max_vel, p->xy_max_accel,
                    ORACLE_INFTY, p->xy_max_vel, p->xy_max_accel};
    for (int a = 0; a < 6; a++) { l[Nx + a] = -b0[a]; u[Nx + a] = b0[a]; }
    for (int r = 0; r < 6 * N; r++) { l[Nx + 6 + r] = -1.0 * xy_max[r]; u[Nx + 6 + r] = xy_max[r]; }
    for (int r = 0; r < Nu; r++) { l[2 * Nx + r] = -1.0 * p->xy_max_jerk; u[2 * Nx + r] = p->xy_max_jerk; }
    for (int r = 0; r < 6 * N; r++) {
        l[2 * Nx + Nu + r] = -1.0 * mixed_max[r]; u[2 * Nx + Nu + r] = mixed_max[r];
    }
    /* updateXYQPGradientVector :495 */
    memset(q, 0, sizeof(double) * (size_t)(Nx + Nu));
    for (int i = 0; i <= N; i++) {
        q[6 * i + 0] = -1.0 * p->xy_state_weight * xyref[6 * i + 0];
        q[6 * i + 3] = -1.0 * p->xy_state_weight * xyref[6 * i + 3];
    }
}

void qp12_yaw_ref_from_xy(int N, const double *xyref, const double *xy_x_opt, double *yawref)
{
    /* computeYawRefTrajectory mpc_12state.cpp:525-552 */
    memset(yawref, 0, sizeof(double) * (size_t)3 * (N + 1));
    for (int i = 0; i <= N; i++)
        yawref[3 * i] = atan2(xyref[6 * i + 3] - xy_x_opt[6 * i + 3], xyref[6 * i] - xy_x_opt[6 * i]);
}

void qp12_merge(int N, const double *xy_sol, const double *z_sol, const double *yaw_sol,
                double *x_opt, double *u_opt)
{
    /* extract{XY,Z,Yaw}Solution :1456-1525, extractSolution :1528-1551 */
    const double *xyu = xy_sol + 6 * (N + 1), *zu = z_sol + 3 * (N + 1), *yu = yaw_sol + 3 * (N + 1);
    for (int i = 0; i <= N; i++) {
        memcpy(x_opt + 12 * i, xy_sol + 6 * i, sizeof(double) * 6);
        memcpy(x_opt + 12 * i + 6, z_sol + 3 * i, sizeof(double) * 3);
        memcpy(x_opt + 12 * i + 9, yaw_sol + 3 * i, sizeof(double) * 3);
        if (i < N) {
            u_opt[4 * i + 0] = xyu[2 * i]; u_opt[4 * i + 1] = xyu[2 * i + 1];
            u_opt[4 * i + 2] = zu[i];      u_opt[4 * i + 3] = yu[i];
        }
    }
}

void qp12_build(const mpc12_params *p, int family, double *P, double *A, double *q, double *l,
                double *u)
{
    /* initMPCProblem mpc_12state.cpp:1282-1350 */
    int N = p->N;
    if (family == FAM_XY) {
        double Ad[36], Bd[12], Q[36], R[4];
        model12(p, FAM_XY, Ad, Bd, Q, R);
        chain_build(6, 2, N, Ad, Bd, Q, R, p->enable_control_smoothing, p->xy_smooth_input_weight,
                    6, P, A);
        /* computeXYBounds on the all-zero _z_x_opt of initVariables (:1149, :1295) */
        double *zz = (double *)calloc((size_t)3 * (N + 1), sizeof(double));
        double *xm = (double *)calloc((size_t)6 * N + 1, sizeof(double));
        double *mm = (double *)calloc((size_t)6 * N + 1, sizeof(double));
        double *ref = (double *)calloc((size_t)6 * (N + 1), sizeof(double));
        double x0[6] = {0};
        qp12_xy_bounds_from_z(p, zz, xm, mm);
        qp12_update_xy(p, x0, ref, xm, mm, q, l, u);
        free(zz); free(xm); free(mm); free(ref);
    } else {
        double Ad[9], Bd[3], Q[9], R[1];
        model12(p, family, Ad, Bd, Q, R);
        double sw = family == FAM_Z ? p->z_smooth_input_weight : p->yaw_smooth_input_weight;
        chain_build(3, 1, N, Ad, Bd, Q, R, p->enable_control_smoothing, sw, 0, P, A);
        double s0[3] = {0};
        double *ref = (double *)calloc((size_t)3 * (N + 1), sizeof(double));
        if (family == FAM_Z) qp12_update_z(p, s0, ref, q, l, u);
        else qp12_update_yaw(p, s0, ref, q, l, u);
        free(ref);
    }
}

/* ---------------------------------------------------------------- per-class rho scale */
/* one set per family: the stages of the 12-state chain may carry different ones */
static double g_cls_fam[4][4] = {{1.0, 1.0, 1.0, 1.0}, {1.0, 1.0, 1.0, 1.0}, {1.0, 1.0, 1.0, 1.0}, {1.0, 1.0, 1.0, 1.0}};
void mpc_set_rho_class_scale_family(int family, double pos, double vel, double acc, double input)
{
    if (family < 0 || family > 3) return;
    g_cls_fam[family][0] = pos; g_cls_fam[family][1] = vel; g_cls_fam[family][2] = acc; g_cls_fam[family][3] = input;
}
void mpc_set_rho_class_scale(double pos, double vel, double acc, double input)
{
    for (int f = 0; f < 4; f++) mpc_set_rho_class_scale_family(f, pos, vel, acc, input);
}

/* schedule of later phases per family: start iteration and absolute class multipliers */
static int g_nphase[4] = {1, 1, 1, 1};
static int g_phase_start[4][3];
static double g_phase_cls[4][3][4];
void mpc_set_rho_phases_family(int family, int nphase, const int *start, const double *cls)
{
    if (family < 0 || family > 3) return;
    if (nphase < 1 || nphase > 3) nphase = 1;
    g_nphase[family] = nphase;
    for (int p = 1; p < nphase; p++) {
        g_phase_start[family][p] = start[p];
        for (int k = 0; k < 4; k++) g_phase_cls[family][p][k] = cls[p * 4 + k];
    }
}

/* per-row multipliers of one set of class multipliers */
static void class_rows(int family, int N, const double *g_cls, double *sc)
{
    int n, m, nx, nu;
    qp_dims(family, N, &n, &m, &nx, &nu);
    int Nx = nx * (N + 1), Nu = nu * N;
    for (int r = 0; r < m; r++) sc[r] = 1.0;
    for (int i = 0; i <= N; i++)
        for (int a = 0; a < nx; a++) {
            int cls = family == FAM_6DOF ? (a < 3 ? 0 : 1) : a % 3;
            sc[Nx + nx * i + a] = g_cls[cls];
        }
    for (int r = 0; r < Nu; r++) sc[2 * Nx + r] = g_cls[3];
    for (int r = 2 * Nx + Nu; r < m; r++) sc[r] = g_cls[(r - 2 * Nx - Nu) % 6 < 4 ? 1 : 2];
}

static void apply_class_scale(osqp_work *w, int family, int N)
{
    const double *g_cls = g_cls_fam[family];
    const int np = g_nphase[family];
    if (np <= 1 && g_cls[0] == 1.0 && g_cls[1] == 1.0 && g_cls[2] == 1.0 && g_cls[3] == 1.0) return;
    int n, m, nx, nu;
    qp_dims(family, N, &n, &m, &nx, &nu);
    double *sc = (double *)calloc((size_t)m * 3, sizeof(double));
    class_rows(family, N, g_cls, sc);
    if (np <= 1) osqp_set_rho_scale(w, sc);
    else {
        for (int p = 1; p < np; p++) class_rows(family, N, g_phase_cls[family][p], sc + (size_t)p * m);
        osqp_set_rho_phases(w, np, g_phase_start[family], sc);
    }
    free(sc);
}

/* ---------------------------------------------------------------- mpcLoop, 6-DOF */
struct mpc6_oracle {
    mpc6_params p;
    int n, m;
    osqp_work *w;
    double *q, *l, *u, *sol;
};

mpc6_oracle *mpc6_create(const mpc6_params *p, const osqp_settings *s)
{
    mpc6_oracle *o = (mpc6_oracle *)calloc(1, sizeof(*o));
    o->p = *p;
    qp_dims(FAM_6DOF, p->N, &o->n, &o->m, NULL, NULL);
    double *P = (double *)calloc((size_t)o->n * o->n, sizeof(double));
    double *A = (double *)calloc((size_t)o->m * o->n, sizeof(double));
    o->q = (double *)calloc(o->n, sizeof(double));
    o->l = (double *)calloc(o->m, sizeof(double));
    o->u = (double *)calloc(o->m, sizeof(double));
    o->sol = (double *)calloc(o->n, sizeof(double));
    qp6_build(p, P, A, o->q, o->l, o->u);
    o->w = osqp_setup(o->n, o->m, P, A, o->q, o->l, o->u, s);
    free(P); free(A);
    if (!o->w) { mpc6_destroy(o); return NULL; }
    apply_class_scale(o->w, FAM_6DOF, p->N);
    return o;
}

void mpc6_destroy(mpc6_oracle *o)
{
    if (!o) return;
    osqp_cleanup(o->w);
    free(o->q); free(o->l); free(o->u); free(o->sol);
    free(o);
}

void mpc6_cold_start(mpc6_oracle *o)
{
    double *zx = (double *)calloc(o->n, sizeof(double)), *zy = (double *)calloc(o->m, sizeof(double));
    int ws = 1;
    osqp_warm_start(o->w, zx, zy);
    (void)ws;
    free(zx); free(zy);
}

int mpc6_loop(mpc6_oracle *o, const double *x0, const double *ref, double *X, double *U,
              int *iters, int *status)
{
    int N = o->p.N;
    qp6_update(&o->p, x0, ref, o->q, o->l, o->u);         /* updateQP :584 */
    if (osqp_update_bounds(o->w, o->l, o->u)) return 0;
    osqp_update_lin_cost(o->w, o->q);
    int st = osqp_solve(o->w);                            /* :639 */
    if (iters) *iters = osqp_get_iter(o->w);
    if (status) *status = st;
    if (st != ORACLE_SOLVED) return 0;                    /* solve() true only when "solved" */
    osqp_get_solution(o->w, o->sol, NULL);                /* extractSolution :653 */
    memcpy(X, o->sol, sizeof(double) * (size_t)6 * (N + 1));
    memcpy(U, o->sol + 6 * (N + 1), sizeof(double) * (size_t)3 * N);
    return 1;
}

int mpc6_batch(const mpc6_params *p, const osqp_settings *s, int B, const double *x0,
               const double *ref, double *X, double *U, int *iters, int *status)
{
    mpc6_oracle *o = mpc6_create(p, s);
    if (!o) return -1;
    int N = p->N, ok = 0;
    for (int b = 0; b < B; b++) {
        /* instances of a batch are unrelated problems: every one starts cold */
        if (s->warm_start) mpc6_cold_start(o);
        ok += mpc6_loop(o, x0 + 6 * b, ref + (long)6 * (N + 1) * b, X + (long)6 * (N + 1) * b,
                        U + (long)3 * N * b, iters ? iters + b : NULL, status ? status + b : NULL);
    }
    mpc6_destroy(o);
    return ok;
}

/* ---------------------------------------------------------------- mpcLoop, 12-state */
struct mpc12_oracle {
    mpc12_params p;
    int n[4], m[4];
    osqp_work *w[4]; /* indexed by family; [0] unused */
    double *q[4], *l[4], *u[4], *sol[4];
    double *xyref, *zref, *yawref, *xy_max, *mixed_max;
};

mpc12_oracle *mpc12_create(const mpc12_params *p, const osqp_settings *s)
{
    return mpc12_create3(p, s, s, s);
}

mpc12_oracle *mpc12_create3(const mpc12_params *p, const osqp_settings *s_xy,
                            const osqp_settings *s_z, const osqp_settings *s_yaw)
{
    const osqp_settings *ss[4] = {NULL, s_xy, s_z, s_yaw};
    mpc12_oracle *o = (mpc12_oracle *)calloc(1, sizeof(*o));
    o->p = *p;
    int N = p->N;
    for (int f = FAM_XY; f <= FAM_YAW; f++) {
        qp_dims(f, N, &o->n[f], &o->m[f], NULL, NULL);
        double *P = (double *)calloc((size_t)o->n[f] * o->n[f], sizeof(double));
        double *A = (double *)calloc((size_t)o->m[f] * o->n[f], sizeof(double));
        o->q[f] = (double *)calloc(o->n[f], sizeof(double));
        o->l[f] = (double *)calloc(o->m[f], sizeof(double));
        o->u[f] = (double *)calloc(o->m[f], sizeof(double));
        o->sol[f] = (double *)calloc(o->n[f], sizeof(double));
        qp12_build(p, f, P, A, o->q[f], o->l[f], o->u[f]);
        o->w[f] = osqp_setup(o->n[f], o->m[f], P, A, o->q[f], o->l[f], o->u[f], ss[f]);
        free(P); free(A);
        if (!o->w[f]) { mpc12_destroy(o); return NULL; }
        apply_class_scale(o->w[f], f, N);
    }
    o->xyref = (double *)calloc((size_t)6 * (N + 1), sizeof(double));
    o->zref = (double *)calloc((size_t)3 * (N + 1), sizeof(double));
    o->yawref = (double *)calloc((size_t)3 * (N + 1), sizeof(double));
    o->xy_max = (double *)calloc((size_t)6 * N + 1, sizeof(double));
    o->mixed_max = (double *)calloc((size_t)6 * N + 1, sizeof(double));
    return o;
}

void mpc12_destroy(mpc12_oracle *o)
{
    if (!o) return;
    for (int f = FAM_XY; f <= FAM_YAW; f++) {
        osqp_cleanup(o->w[f]);
        free(o->q[f]); free(o->l[f]); free(o->u[f]); free(o->sol[f]);
    }
    free(o->xyref); free(o->zref); free(o->yawref); free(o->xy_max); free(o->mixed_max);
    free(o);
}

int mpc12_loop(mpc12_oracle *o, const double *x12, const double *ref12, double *X, double *U,
               int *iters, int *status, double *inter)
{
    const mpc12_params *p = &o->p;
    int N = p->N;
    double xy0[6], z0[3], yaw0[3];
    int it[3] = {0, 0, 0}, st[3] = {ORACLE_UNSOLVED, ORACLE_UNSOLVED, ORACLE_UNSOLVED};
    int ok = 0;
    qp12_split_state(x12, xy0, z0, yaw0);
    qp12_split_ref(N, ref12, o->xyref, o->zref, o->yawref);
    /* stage 1: Z  (:1395-1411) */
    qp12_update_z(p, z0, o->zref, o->q[FAM_Z], o->l[FAM_Z], o->u[FAM_Z]);
    if (osqp_update_bounds(o->w[FAM_Z], o->l[FAM_Z], o->u[FAM_Z])) goto done;
    osqp_update_lin_cost(o->w[FAM_Z], o->q[FAM_Z]);
    st[0] = osqp_solve(o->w[FAM_Z]); it[0] = osqp_get_iter(o->w[FAM_Z]);
    if (st[0] != ORACLE_SOLVED) goto done;
    osqp_get_solution(o->w[FAM_Z], o->sol[FAM_Z], NULL);
    /* stage 2: XY (:1413-1428) */
    qp12_xy_bounds_from_z(p, o->sol[FAM_Z], o->xy_max, o->mixed_max);
    qp12_update_xy(p, xy0, o->xyref, o->xy_max, o->mixed_max, o->q[FAM_XY], o->l[FAM_XY],
                   o->u[FAM_XY]);
    if (osqp_update_bounds(o->w[FAM_XY], o->l[FAM_XY], o->u[FAM_XY])) goto done;
    osqp_update_lin_cost(o->w[FAM_XY], o->q[FAM_XY]);
    st[1] = osqp_solve(o->w[FAM_XY]); it[1] = osqp_get_iter(o->w[FAM_XY]);
    if (st[1] != ORACLE_SOLVED) goto done;
    osqp_get_solution(o->w[FAM_XY], o->sol[FAM_XY], NULL);
    /* stage 3: Yaw (:1430-1445); bounds first, then the reference from the XY solution */
    qp12_yaw_ref_from_xy(N, o->xyref, o->sol[FAM_XY], o->yawref);
    qp12_update_yaw(p, yaw0, o->yawref, o->q[FAM_YAW], o->l[FAM_YAW], o->u[FAM_YAW]);
    if (osqp_update_bounds(o->w[FAM_YAW], o->l[FAM_YAW], o->u[FAM_YAW])) goto done;
    osqp_update_lin_cost(o->w[FAM_YAW], o->q[FAM_YAW]);
    st[2] = osqp_solve(o->w[FAM_YAW]); it[2] = osqp_get_iter(o->w[FAM_YAW]);
    if (st[2] != ORACLE_SOLVED) goto done;
    osqp_get_solution(o->w[FAM_YAW], o->sol[FAM_YAW], NULL);
    qp12_merge(N, o->sol[FAM_XY], o->sol[FAM_Z], o->sol[FAM_YAW], X, U);
    if (inter) {
        double *d = inter;
        memcpy(d, o->sol[FAM_Z], sizeof(double) * o->n[FAM_Z]); d += o->n[FAM_Z];
        memcpy(d, o->xy_max, sizeof(double) * 6 * N); d += 6 * N;
        memcpy(d, o->mixed_max, sizeof(double) * 6 * N); d += 6 * N;
        memcpy(d, o->sol[FAM_XY], sizeof(double) * o->n[FAM_XY]); d += o->n[FAM_XY];
        memcpy(d, o->yawref, sizeof(double) * 3 * (N + 1)); d += 3 * (N + 1);
        memcpy(d, o->sol[FAM_YAW], sizeof(double) * o->n[FAM_YAW]);
    }
    ok = 1;
done:
    if (iters) { iters[0] = it[0]; iters[1] = it[1]; iters[2] = it[2]; }
    if (status) { status[0] = st[0]; status[1] = st[1]; status[2] = st[2]; }
    return ok;
}

int mpc12_batch(const mpc12_params *p, const osqp_settings *s, int B, const double *x12,
                const double *ref12, double *X, double *U, int *iters, int *status)
{
    return mpc12_batch3(p, s, s, s, B, x12, ref12, X, U, iters, status);
}

int mpc12_batch3(const mpc12_params *p, const osqp_settings *s_xy, const osqp_settings *s_z,
                 const osqp_settings *s_yaw, int B, const double *x12, const double *ref12,
                 double *X, double *U, int *iters, int *status)
{
    const osqp_settings *s = s_xy;
    mpc12_oracle *o = mpc12_create3(p, s_xy, s_z, s_yaw);
    if (!o) return -1;
    int N = p->N, ok = 0;
    for (int b = 0; b < B; b++) {
        if (s->warm_start) {
            for (int f = FAM_XY; f <= FAM_YAW; f++) {
                double *zx = (double *)calloc(o->n[f], sizeof(double));
                double *zy = (double *)calloc(o->m[f], sizeof(double));
                osqp_warm_start(o->w[f], zx, zy);
                free(zx); free(zy);
            }
        }
        ok += mpc12_loop(o, x12 + 12 * b, ref12 + (long)12 * (N + 1) * b,
                         X + (long)12 * (N + 1) * b, U + (long)4 * N * b,
                         iters ? iters + 3 * b : NULL, status ? status + 3 * b : NULL, NULL);
    }
    mpc12_destroy(o);
    return ok;
}

/* ---------------------------------------------------------------- synthetic inputs */
static unsigned long long splitmix64(unsigned long long x)
{
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

double synth_uniform(unsigned long long seed, unsigned long long stream, unsigned long long ctr)
{
    unsigned long long h = splitmix64(seed ^ splitmix64(stream * 0xD1342543DE82EF95ULL + ctr));
    return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}

static double uni(unsigned long long seed, unsigned long long k, int c, double lo, double hi)
{
    return lo + (hi - lo) * synth_uniform(seed, k, (unsigned long long)c);
}

void synth_mpc6(unsigned long long seed, int N, double dt, int B, int first, double *x0,
                double *ref)
{
    for (int b = 0; b < B; b++) {
        unsigned long long k = (unsigned long long)(first + b);
        double p0[3], v0[3], pt[3], vt[3];
        p0[0] = uni(seed, k, 0, -50, 50); p0[1] = uni(seed, k, 1, -50, 50);
        p0[2] = uni(seed, k, 2, 5, 30);
        for (int a = 0; a < 3; a++) v0[a] = uni(seed, k, 3 + a, -3, 3);
        for (int a = 0; a < 3; a++) pt[a] = p0[a] + uni(seed, k, 6 + a, -8, 8);
        if (pt[2] < 2.0) pt[2] = 2.0;
        for (int a = 0; a < 3; a++) vt[a] = uni(seed, k, 9 + a, -3, 3);
        for (int a = 0; a < 3; a++) { x0[6 * b + a] = p0[a]; x0[6 * b + 3 + a] = v0[a]; }
        double *r = ref + (long)6 * (N + 1) * b;
        for (int i = 0; i <= N; i++)
            for (int a = 0; a < 3; a++) {
                r[6 * i + a] = pt[a] + vt[a] * dt * i;
                r[6 * i + 3 + a] = vt[a];
            }
    }
}

void synth_mpc12(unsigned long long seed, int N, double dt, int B, int first, double *x12,
                 double *ref12)
{
    for (int b = 0; b < B; b++) {
        unsigned long long k = (unsigned long long)(first + b);
        double *x = x12 + 12 * b;
        x[0] = uni(seed, k, 0, -50, 50); x[1] = uni(seed, k, 1, -3, 3); x[2] = uni(seed, k, 2, -1, 1);
        x[3] = uni(seed, k, 3, -50, 50); x[4] = uni(seed, k, 4, -3, 3); x[5] = uni(seed, k, 5, -1, 1);
        x[6] = uni(seed, k, 6, 5, 30);   x[7] = uni(seed, k, 7, -2, 2); x[8] = uni(seed, k, 8, -1, 1);
        x[9] = uni(seed, k, 9, -M_PI, M_PI); x[10] = uni(seed, k, 10, -0.5, 0.5); x[11] = 0.0;
        double pt[3], vt[3];
        pt[0] = x[0] + uni(seed, k, 11, -8, 8); pt[1] = x[3] + uni(seed, k, 12, -8, 8);
        pt[2] = x[6] + uni(seed, k, 13, -3, 3);
        for (int a = 0; a < 3; a++) vt[a] = uni(seed, k, 14 + a, -3, 3);
        double *r = ref12 + (long)12 * (N + 1) * b;
        for (int i = 0; i <= N; i++) {
            for (int a = 0; a < 3; a++) {
                r[12 * i + 3 * a + 0] = pt[a] + vt[a] * dt * i;
                r[12 * i + 3 * a + 1] = vt[a];
                r[12 * i + 3 * a + 2] = 0.0;
            }
            r[12 * i + 9] = r[12 * i + 10] = r[12 * i + 11] = 0.0;
        }
    }
}
