/* Optional pin of the CPU oracle against the real OSQP library (test infrastructure, not product).
 *
 * The reference solves its QPs through osqp-eigen v0.8.0 on OSQP commit 25b6b39 (setup.sh:32-56)
 * with the library's default settings except warm starting on and verbosity off
 * (mpc_6dof.cpp:525-526, mpc_12state.cpp initQPSolver).  Neither library is on this image, so this
 * file is only compiled when the Makefile beside it finds osqp.h; it has never been compiled or
 * run here.  It takes the dense P and A the oracle builds, hands them to OSQP as CSC (upper
 * triangle of P, as osqp-eigen does) and returns solution, status and iteration count, so that
 * tests/test_real_osqp_optional.py can compare oracle/osqp_restate.c with the real thing.
 *
 * Both C APIs are covered: 0.6.x (OSQPWorkspace, c_int / c_float; what 25b6b39 exposes) and 1.x
 * (OSQPSolver, OSQPInt / OSQPFloat).
 */
#include <stdlib.h>
#include <string.h>
#include "osqp.h"

#if defined(OSQP_VERSION_MAJOR) || defined(OSQP_API_TYPES_H)
#define REAL_OSQP_V1 1
typedef OSQPInt ro_int;
typedef OSQPFloat ro_float;
#else
#define REAL_OSQP_V1 0
typedef c_int ro_int;
typedef c_float ro_float;
#endif

typedef struct {
    ro_int n, m;
    ro_int *Pp, *Pi, *Ap, *Ai;
    ro_float *Px, *Ax, *q, *l, *u;
#if REAL_OSQP_V1
    OSQPSolver *solver;
    OSQPCscMatrix *P, *A;
    OSQPSettings *settings;
#else
    OSQPWorkspace *work;
    OSQPSettings *settings;
    OSQPData *data;
#endif
} real_osqp;

/* dense row-major [rows][cols] -> CSC of the entries that are not exactly zero; upper != 0 keeps j >= i only */
static void to_csc(const double *D, int rows, int cols, int upper, ro_int **pp, ro_int **pi, ro_float **px)
{
    size_t nnz = 0;
    for (int j = 0; j < cols; j++)
        for (int i = 0; i < (upper ? j + 1 : rows); i++)
            if (D[(size_t)i * cols + j] != 0.0) nnz++;
    ro_int *p = malloc(sizeof(ro_int) * (cols + 1)), *ix = malloc(sizeof(ro_int) * (nnz ? nnz : 1));
    ro_float *x = malloc(sizeof(ro_float) * (nnz ? nnz : 1));
    size_t k = 0;
    for (int j = 0; j < cols; j++) {
        p[j] = (ro_int)k;
        for (int i = 0; i < (upper ? j + 1 : rows); i++)
            if (D[(size_t)i * cols + j] != 0.0) { ix[k] = i; x[k] = (ro_float)D[(size_t)i * cols + j]; k++; }
    }
    p[cols] = (ro_int)k;
    *pp = p; *pi = ix; *px = x;
}

static ro_float *dup_vec(const double *v, int n)
{
    ro_float *o = malloc(sizeof(ro_float) * (n ? n : 1));
    for (int i = 0; i < n; i++) o[i] = (ro_float)v[i];
    return o;
}

void real_osqp_free(real_osqp *h);

/* OSQP defaults + warm start on, verbose off; eps / max_iter <= 0 keep the library's defaults */
real_osqp *real_osqp_setup(int n, int m, const double *P, const double *A, const double *q, const double *l,
                           const double *u, double eps_abs, double eps_rel, int max_iter)
{
    real_osqp *h = calloc(1, sizeof(real_osqp));
    h->n = n; h->m = m;
    to_csc(P, n, n, 1, &h->Pp, &h->Pi, &h->Px);
    to_csc(A, m, n, 0, &h->Ap, &h->Ai, &h->Ax);
    h->q = dup_vec(q, n); h->l = dup_vec(l, m); h->u = dup_vec(u, m);
    h->settings = malloc(sizeof(OSQPSettings));
    osqp_set_default_settings(h->settings);
    h->settings->verbose = 0;
#if REAL_OSQP_V1
    h->settings->warm_starting = 1;
#else
    h->settings->warm_start = 1;
#endif
    if (eps_abs > 0) h->settings->eps_abs = eps_abs;
    if (eps_rel > 0) h->settings->eps_rel = eps_rel;
    if (max_iter > 0) h->settings->max_iter = max_iter;
#if REAL_OSQP_V1
    h->P = malloc(sizeof(OSQPCscMatrix)); h->A = malloc(sizeof(OSQPCscMatrix));
    csc_set_data(h->P, n, n, h->Pp[n], h->Px, h->Pi, h->Pp);
    csc_set_data(h->A, m, n, h->Ap[n], h->Ax, h->Ai, h->Ap);
    if (osqp_setup(&h->solver, h->P, h->q, h->A, h->l, h->u, m, n, h->settings)) { real_osqp_free(h); return NULL; }
#else
    h->data = calloc(1, sizeof(OSQPData));
    h->data->n = n; h->data->m = m;
    h->data->P = csc_matrix(n, n, h->Pp[n], h->Px, h->Pi, h->Pp);
    h->data->A = csc_matrix(m, n, h->Ap[n], h->Ax, h->Ai, h->Ap);
    h->data->q = h->q; h->data->l = h->l; h->data->u = h->u;
    if (osqp_setup(&h->work, h->data, h->settings)) { real_osqp_free(h); return NULL; }
#endif
    return h;
}

/* the per-call updates of MPC::mpcLoop: updateGradient, updateBounds (mpc_6dof.cpp:616-640) */
int real_osqp_update(real_osqp *h, const double *q, const double *l, const double *u)
{
    ro_float *qq = dup_vec(q, (int)h->n), *ll = dup_vec(l, (int)h->m), *uu = dup_vec(u, (int)h->m);
#if REAL_OSQP_V1
    int rc = (int)osqp_update_data_vec(h->solver, qq, ll, uu);
#else
    int rc = (int)osqp_update_lin_cost(h->work, qq);
    if (!rc) rc = (int)osqp_update_bounds(h->work, ll, uu);
#endif
    free(qq); free(ll); free(uu);
    return rc;
}

/* returns OSQP's status_val (1 solved, -3 primal infeasible, -2 max iterations reached, ...) */
int real_osqp_solve(real_osqp *h, double *x, double *y, int *iter, double *pri_res, double *dua_res)
{
#if REAL_OSQP_V1
    if (osqp_solve(h->solver)) return -100;
    const OSQPInfo *info = h->solver->info;
    const OSQPSolution *sol = h->solver->solution;
    for (int i = 0; i < h->n; i++) x[i] = sol->x[i];
    for (int i = 0; i < h->m; i++) y[i] = sol->y[i];
    *pri_res = info->prim_res; *dua_res = info->dual_res;
#else
    if (osqp_solve(h->work)) return -100;
    const OSQPInfo *info = h->work->info;
    const OSQPSolution *sol = h->work->solution;
    for (int i = 0; i < h->n; i++) x[i] = sol->x[i];
    for (int i = 0; i < h->m; i++) y[i] = sol->y[i];
    *pri_res = info->pri_res; *dua_res = info->dua_res;
#endif
    *iter = (int)info->iter;
    return (int)info->status_val;
}

void real_osqp_free(real_osqp *h)
{
    if (!h) return;
#if REAL_OSQP_V1
    if (h->solver) osqp_cleanup(h->solver);
    free(h->P); free(h->A);
#else
    if (h->work) osqp_cleanup(h->work);
    if (h->data) { free(h->data->P); free(h->data->A); free(h->data); }
#endif
    free(h->settings);
    free(h->Pp); free(h->Pi); free(h->Px); free(h->Ap); free(h->Ai); free(h->Ax);
    free(h->q); free(h->l); free(h->u);
    free(h);
}
