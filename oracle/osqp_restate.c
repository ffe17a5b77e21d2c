/*
 * CPU ORACLE -- TEST INFRASTRUCTURE ONLY (see mpc_oracle.h).  PARITY UNPINNED.
 *
 * Restatement of the OSQP algorithm that the reference reaches through osqp-eigen:
 *   setup    <- OsqpEigen::Solver::initSolver()      mpc_6dof.cpp:540, mpc_12state.cpp:1035,1055,1075
 *   update   <- updateBounds / updateGradient        mpc_6dof.cpp:588,599, mpc_12state.cpp:1202-1267
 *   solve    <- OsqpEigen::Solver::solve()           mpc_6dof.cpp:639, mpc_12state.cpp:1405,1422,1439
 * OSQP itself (commit 25b6b39) is not vendored in the reference; what follows is
 * written from the published algorithm (Stellato et al. 2020: Algorithm 1, the Ruiz
 * equilibration of section 5.1, the per-constraint step sizes and the rho adaptation of
 * section 5.2, the termination and infeasibility criteria of section 3.4) using the
 * solver's documented constants.  The linear system is solved in its reduced form
 *   (P + sigma I + A' diag(rho) A) xt = sigma x - q + A'(rho z - y),   zt = A xt
 * by a dense Cholesky factorisation; this is algebraically the quasi-definite KKT solve
 * of the paper with nu eliminated.  settings.linsys = 1 solves the quasi-definite system itself,
 *   [P + sigma I, A'; A, -diag(1/rho)] [xt; nu] = [sigma x - q; z - y/rho],  zt = z + (nu - y)/rho,
 * with a sparse LDL' factorisation (fill-reducing minimum-degree ordering, elimination tree,
 * up-looking numeric factorisation) as OSQP's default linear-system solver does on the sparse
 * matrices the reference hands it (sparseView(), mpc_6dof.cpp:532-536); it is the form the CPU
 * baseline is timed in.
 */
#include "mpc_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define RHO_MIN 1e-6
#define RHO_MAX 1e6
#define RHO_EQ_OVER_RHO_INEQ 1e3
#define RHO_TOL 1e-4
#define MIN_SCALING 1e-4
#define MAX_SCALING 1e4
#define DIVISION_TOL (1.0 / ORACLE_INFTY)

typedef struct {
    int rows, cols, nnz;
    int *ptr, *idx;
    double *val;
} csr;

/* sparse LDL' of the permuted quasi-definite KKT matrix */
typedef struct {
    int N;                 /* n + m */
    int *perm, *iperm;     /* perm[new] = old */
    int *Kp, *Ki;          /* upper triangle, column-compressed, permuted */
    double *Kx;
    int *rho_slot;         /* position in Kx of the -1/rho entry of constraint i */
    int *parent, *Lnz, *Lp, *Li, *flag, *pattern;
    double *Lx, *D, *Y, *b;
} kkt_ldl;

struct osqp_work {
    int n, m;
    kkt_ldl *kkt;
    osqp_settings s;
    csr P, A, At;          /* scaled data */
    double *q, *l, *u;     /* scaled */
    double *D, *E, *Dinv, *Einv, c, cinv;
    double *rho_vec, *rho_inv_vec;
    double *rho_scale;     /* optional per-row multiplier of rho on inequality rows (NULL: 1) */
    int *constr_type;
    double *L;             /* dense Cholesky factor of the reduced matrix, n x n lower */
    double *kkt_rhs;       /* linsys = 1: right-hand side / solution of the KKT system, n + m */
    double *x, *z, *y, *x_prev, *z_prev, *xt, *zt, *delta_x, *delta_y;
    double *Ax, *Px, *Aty, *wn, *wm;
    double *sol_x, *sol_y;
    int iter, status, rho_updates;
    double pri_res, dua_res;
    /* step-size schedule of the engine's fixed-rho mode (osqp_set_rho_phases): per phase a per-row
       multiplier of rho on the inequality rows and the iteration it starts at; factor of every
       phase kept (dense mode) so that moving between them costs a pointer swap */
    int nphase, phase, phase_start[3];
    double *phase_scale[3], *phase_L[3];
};

/* ---------------------------------------------------------------- small helpers */
static double *dalloc(int k) { return (double *)calloc((size_t)(k > 0 ? k : 1), sizeof(double)); }

static csr csr_from_dense(const double *M, int rows, int cols)
{
    csr S;
    S.rows = rows; S.cols = cols; S.nnz = 0;
    for (long i = 0; i < (long)rows * cols; i++) if (M[i] != 0.0) S.nnz++;
    S.ptr = (int *)calloc((size_t)rows + 1, sizeof(int));
    S.idx = (int *)calloc((size_t)(S.nnz > 0 ? S.nnz : 1), sizeof(int));
    S.val = dalloc(S.nnz);
    int k = 0;
    for (int i = 0; i < rows; i++) {
        S.ptr[i] = k;
        for (int j = 0; j < cols; j++) {
            double v = M[(long)i * cols + j];
            if (v != 0.0) { S.idx[k] = j; S.val[k] = v; k++; }
        }
    }
    S.ptr[rows] = k;
    return S;
}

static csr csr_transpose(const csr *S)
{
    csr T;
    T.rows = S->cols; T.cols = S->rows; T.nnz = S->nnz;
    T.ptr = (int *)calloc((size_t)T.rows + 1, sizeof(int));
    T.idx = (int *)calloc((size_t)(T.nnz > 0 ? T.nnz : 1), sizeof(int));
    T.val = dalloc(T.nnz);
    for (int k = 0; k < S->nnz; k++) T.ptr[S->idx[k] + 1]++;
    for (int i = 0; i < T.rows; i++) T.ptr[i + 1] += T.ptr[i];
    int *fill = (int *)calloc((size_t)T.rows + 1, sizeof(int));
    for (int i = 0; i < S->rows; i++)
        for (int k = S->ptr[i]; k < S->ptr[i + 1]; k++) {
            int j = S->idx[k];
            int d = T.ptr[j] + fill[j]++;
            T.idx[d] = i; T.val[d] = S->val[k];
        }
    free(fill);
    return T;
}

static void csr_free(csr *S) { free(S->ptr); free(S->idx); free(S->val); }

static void csr_mv(const csr *S, const double *x, double *y)
{
    for (int i = 0; i < S->rows; i++) {
        double a = 0.0;
        for (int k = S->ptr[i]; k < S->ptr[i + 1]; k++) a += S->val[k] * x[S->idx[k]];
        y[i] = a;
    }
}

static double norm_inf(const double *v, int k)
{
    double a = 0.0;
    for (int i = 0; i < k; i++) { double t = fabs(v[i]); if (t > a) a = t; }
    return a;
}
static double scaled_norm_inf(const double *s, const double *v, int k)
{
    double a = 0.0;
    for (int i = 0; i < k; i++) { double t = fabs(s[i] * v[i]); if (t > a) a = t; }
    return a;
}

static double limit_scaling(double v)
{
    if (v < MIN_SCALING) return 1.0;
    if (v > MAX_SCALING) return MAX_SCALING;
    return v;
}

/* ---------------------------------------------------------------- settings */
void osqp_default_settings(osqp_settings *s)
{
    s->rho = 0.1; s->sigma = 1e-6; s->alpha = 1.6;
    s->eps_abs = 1e-3; s->eps_rel = 1e-3;
    s->eps_prim_inf = 1e-4; s->eps_dual_inf = 1e-4;
    s->max_iter = 4000; s->check_termination = 25;
    s->scaling = 10; s->adaptive_rho = 1; s->adaptive_rho_interval = 0;
    s->adaptive_rho_tolerance = 5.0; s->warm_start = 1; s->scaled_termination = 0;
    s->check_fine = 0;
    s->linsys = 0;
}

void osqp_fixed_rho_settings(osqp_settings *s, double rho)
{
    osqp_default_settings(s);
    s->rho = rho; s->scaling = 0; s->adaptive_rho = 0;
    s->check_fine = 0;
}

/* ---------------------------------------------------------------- scaling (sec. 5.1) */
static void scale_data(osqp_work *w, double *Pd, double *Ad)
{
    int n = w->n, m = w->m;
    for (int i = 0; i < n; i++) w->D[i] = 1.0;
    for (int i = 0; i < m; i++) w->E[i] = 1.0;
    w->c = 1.0;
    double *Dt = dalloc(n), *Et = dalloc(m);
    for (int it = 0; it < w->s.scaling; it++) {
        /* inf-norms of the columns of the KKT matrix [P A'; A 0] */
        for (int j = 0; j < n; j++) {
            double a = 0.0;
            for (int i = 0; i < n; i++) { double t = fabs(Pd[(long)i * n + j]); if (t > a) a = t; }
            for (int i = 0; i < m; i++) { double t = fabs(Ad[(long)i * n + j]); if (t > a) a = t; }
            Dt[j] = a;
        }
        for (int i = 0; i < m; i++) {
            double a = 0.0;
            for (int j = 0; j < n; j++) { double t = fabs(Ad[(long)i * n + j]); if (t > a) a = t; }
            Et[i] = a;
        }
        for (int j = 0; j < n; j++) Dt[j] = 1.0 / sqrt(limit_scaling(Dt[j]));
        for (int i = 0; i < m; i++) Et[i] = 1.0 / sqrt(limit_scaling(Et[i]));
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) Pd[(long)i * n + j] *= Dt[i] * Dt[j];
        for (int i = 0; i < m; i++)
            for (int j = 0; j < n; j++) Ad[(long)i * n + j] *= Et[i] * Dt[j];
        for (int j = 0; j < n; j++) { w->q[j] *= Dt[j]; w->D[j] *= Dt[j]; }
        for (int i = 0; i < m; i++) w->E[i] *= Et[i];
        /* cost normalisation */
        double mean = 0.0;
        for (int j = 0; j < n; j++) {
            double a = 0.0;
            for (int i = 0; i < n; i++) { double t = fabs(Pd[(long)i * n + j]); if (t > a) a = t; }
            mean += a;
        }
        mean /= (double)n;
        double ct = limit_scaling(mean);
        double qn = limit_scaling(norm_inf(w->q, n));
        if (qn > ct) ct = qn;
        ct = 1.0 / ct;
        for (long i = 0; i < (long)n * n; i++) Pd[i] *= ct;
        for (int j = 0; j < n; j++) w->q[j] *= ct;
        w->c *= ct;
    }
    for (int j = 0; j < n; j++) w->Dinv[j] = 1.0 / w->D[j];
    for (int i = 0; i < m; i++) w->Einv[i] = 1.0 / w->E[i];
    w->cinv = 1.0 / w->c;
    for (int i = 0; i < m; i++) { w->l[i] *= w->E[i]; w->u[i] *= w->E[i]; }
    free(Dt); free(Et);
}

/* ---------------------------------------------------------------- rho and factor */
static int classify(const osqp_work *w, int i)
{
    if (w->l[i] < -ORACLE_INFTY * MIN_SCALING && w->u[i] > ORACLE_INFTY * MIN_SCALING) return -1;
    if (w->u[i] - w->l[i] < RHO_TOL) return 1;
    return 0;
}

static void apply_rho(osqp_work *w)
{
    for (int i = 0; i < w->m; i++) {
        double r = w->constr_type[i] == -1 ? RHO_MIN
                 : w->constr_type[i] == 1 ? RHO_EQ_OVER_RHO_INEQ * w->s.rho
                 : (w->rho_scale ? w->rho_scale[i] * w->s.rho : w->s.rho);
        w->rho_vec[i] = r; w->rho_inv_vec[i] = 1.0 / r;
    }
}

static int kkt_numeric(osqp_work *w);

static int factor(osqp_work *w)
{
    if (w->kkt) return kkt_numeric(w);
    int n = w->n;
    double *K = w->L;
    memset(K, 0, sizeof(double) * (size_t)n * n);
    for (int i = 0; i < n; i++) {
        for (int k = w->P.ptr[i]; k < w->P.ptr[i + 1]; k++) K[(long)i * n + w->P.idx[k]] += w->P.val[k];
        K[(long)i * n + i] += w->s.sigma;
    }
    for (int r = 0; r < w->m; r++) {
        double rho = w->rho_vec[r];
        for (int a = w->A.ptr[r]; a < w->A.ptr[r + 1]; a++)
            for (int b = w->A.ptr[r]; b < w->A.ptr[r + 1]; b++)
                K[(long)w->A.idx[a] * n + w->A.idx[b]] += rho * w->A.val[a] * w->A.val[b];
    }
    /* in-place lower Cholesky */
    for (int j = 0; j < n; j++) {
        double d = K[(long)j * n + j];
        for (int k = 0; k < j; k++) d -= K[(long)j * n + k] * K[(long)j * n + k];
        if (!(d > 0.0)) return 1;
        d = sqrt(d);
        K[(long)j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = K[(long)i * n + j];
            const double *ri = &K[(long)i * n], *rj = &K[(long)j * n];
            for (int k = 0; k < j; k++) s -= ri[k] * rj[k];
            K[(long)i * n + j] = s / d;
        }
    }
    return 0;
}

static void chol_solve(const osqp_work *w, double *b)
{
    int n = w->n;
    const double *L = w->L;
    for (int i = 0; i < n; i++) {
        double s = b[i];
        const double *ri = &L[(long)i * n];
        for (int k = 0; k < i; k++) s -= ri[k] * b[k];
        b[i] = s / ri[i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = b[i];
        for (int k = i + 1; k < n; k++) s -= L[(long)k * n + i] * b[k];
        b[i] = s / L[(long)i * n + i];
    }
}


/* ---------------------------------------------------------------- sparse KKT LDL' (linsys = 1) */
typedef struct { int i, j; double v; int rho_of; } trip;

static int trip_cmp(const void *a, const void *b)
{
    const trip *x = (const trip *)a, *y = (const trip *)b;
    if (x->j != y->j) return x->j < y->j ? -1 : 1;
    return x->i < y->i ? -1 : (x->i > y->i ? 1 : 0);
}

/* greedy minimum-degree ordering of a symmetric pattern given as bit rows (adj[v] has bit u set) */
static void min_degree_order(int N, unsigned long long *adj, int words, int *perm)
{
    char *gone = (char *)calloc((size_t)N, 1);
    int *deg = (int *)calloc((size_t)N, sizeof(int));
    for (int v = 0; v < N; v++) {
        int d = 0;
        for (int k = 0; k < words; k++) d += __builtin_popcountll(adj[(size_t)v * words + k]);
        deg[v] = d;
    }
    for (int step = 0; step < N; step++) {
        int best = -1;
        for (int v = 0; v < N; v++) if (!gone[v] && (best < 0 || deg[v] < deg[best])) best = v;
        perm[step] = best;
        gone[best] = 1;
        const unsigned long long *av = adj + (size_t)best * words;
        for (int k = 0; k < words; k++) {
            unsigned long long bits = av[k];
            while (bits) {
                int u = k * 64 + __builtin_ctzll(bits);
                bits &= bits - 1;
                unsigned long long *au = adj + (size_t)u * words;
                int d = 0;
                for (int q = 0; q < words; q++) au[q] |= av[q];
                au[best >> 6] &= ~(1ull << (best & 63));
                au[u >> 6] &= ~(1ull << (u & 63));
                for (int q = 0; q < words; q++) d += __builtin_popcountll(au[q]);
                deg[u] = d;
            }
        }
    }
    free(gone); free(deg);
}

static void kkt_free(kkt_ldl *k)
{
    if (!k) return;
    free(k->perm); free(k->iperm); free(k->Kp); free(k->Ki); free(k->Kx); free(k->rho_slot);
    free(k->parent); free(k->Lnz); free(k->Lp); free(k->Li); free(k->flag); free(k->pattern);
    free(k->Lx); free(k->D); free(k->Y); free(k->b);
    free(k);
}

static int kkt_setup(osqp_work *w)
{
    const int n = w->n, m = w->m, N = n + m;
    kkt_ldl *k = (kkt_ldl *)calloc(1, sizeof(kkt_ldl));
    k->N = N;
    /* entries of the upper triangle in the original ordering */
    int cap = w->P.nnz + n + w->A.nnz + m, cnt = 0;
    trip *t = (trip *)calloc((size_t)cap, sizeof(trip));
    char *has_diag = (char *)calloc((size_t)n, 1);
    for (int i = 0; i < n; i++)
        for (int p = w->P.ptr[i]; p < w->P.ptr[i + 1]; p++) {
            int j = w->P.idx[p];
            if (j < i) continue;
            double v = w->P.val[p];
            if (j == i) { v += w->s.sigma; has_diag[i] = 1; }
            t[cnt].i = i; t[cnt].j = j; t[cnt].v = v; t[cnt].rho_of = -1; cnt++;
        }
    for (int i = 0; i < n; i++)
        if (!has_diag[i]) { t[cnt].i = i; t[cnt].j = i; t[cnt].v = w->s.sigma; t[cnt].rho_of = -1; cnt++; }
    for (int r = 0; r < m; r++) {
        for (int p = w->A.ptr[r]; p < w->A.ptr[r + 1]; p++) {
            t[cnt].i = w->A.idx[p]; t[cnt].j = n + r; t[cnt].v = w->A.val[p]; t[cnt].rho_of = -1; cnt++;
        }
        t[cnt].i = n + r; t[cnt].j = n + r; t[cnt].v = -w->rho_inv_vec[r]; t[cnt].rho_of = r; cnt++;
    }
    free(has_diag);
    /* ordering */
    const int words = (N + 63) / 64;
    unsigned long long *adj = (unsigned long long *)calloc((size_t)N * words, sizeof(unsigned long long));
    for (int e = 0; e < cnt; e++)
        if (t[e].i != t[e].j) {
            adj[(size_t)t[e].i * words + (t[e].j >> 6)] |= 1ull << (t[e].j & 63);
            adj[(size_t)t[e].j * words + (t[e].i >> 6)] |= 1ull << (t[e].i & 63);
        }
    k->perm = (int *)calloc((size_t)N, sizeof(int));
    k->iperm = (int *)calloc((size_t)N, sizeof(int));
    min_degree_order(N, adj, words, k->perm);
    free(adj);
    for (int a = 0; a < N; a++) k->iperm[k->perm[a]] = a;
    /* permuted upper triangle, column-compressed */
    for (int e = 0; e < cnt; e++) {
        int a = k->iperm[t[e].i], b = k->iperm[t[e].j];
        if (a > b) { int c = a; a = b; b = c; }
        t[e].i = a; t[e].j = b;
    }
    qsort(t, (size_t)cnt, sizeof(trip), trip_cmp);
    k->Kp = (int *)calloc((size_t)N + 1, sizeof(int));
    k->Ki = (int *)calloc((size_t)cnt, sizeof(int));
    k->Kx = dalloc(cnt);
    k->rho_slot = (int *)calloc((size_t)m + 1, sizeof(int));
    for (int e = 0; e < cnt; e++) {
        k->Kp[t[e].j + 1]++;
        k->Ki[e] = t[e].i; k->Kx[e] = t[e].v;
        if (t[e].rho_of >= 0) k->rho_slot[t[e].rho_of] = e;
    }
    for (int j = 0; j < N; j++) k->Kp[j + 1] += k->Kp[j];
    free(t);
    /* elimination tree and column counts of L */
    k->parent = (int *)calloc((size_t)N, sizeof(int));
    k->Lnz = (int *)calloc((size_t)N, sizeof(int));
    k->flag = (int *)calloc((size_t)N, sizeof(int));
    k->pattern = (int *)calloc((size_t)N, sizeof(int));
    k->Lp = (int *)calloc((size_t)N + 1, sizeof(int));
    for (int c = 0; c < N; c++) {
        k->parent[c] = -1; k->flag[c] = c; k->Lnz[c] = 0;
        for (int p = k->Kp[c]; p < k->Kp[c + 1]; p++) {
            int i = k->Ki[p];
            for (; i < c && k->flag[i] != c; i = k->parent[i]) {
                if (k->parent[i] == -1) k->parent[i] = c;
                k->Lnz[i]++;
                k->flag[i] = c;
            }
        }
    }
    for (int c = 0; c < N; c++) k->Lp[c + 1] = k->Lp[c] + k->Lnz[c];
    k->Li = (int *)calloc((size_t)(k->Lp[N] > 0 ? k->Lp[N] : 1), sizeof(int));
    k->Lx = dalloc(k->Lp[N]);
    k->D = dalloc(N); k->Y = dalloc(N); k->b = dalloc(N);
    w->kkt = k;
    return kkt_numeric(w);
}

/* up-looking LDL': row c of L is the solution of L(0:c-1, 0:c-1) D y = K(0:c-1, c) on the
 * elimination-tree reach of column c */
static int kkt_numeric(osqp_work *w)
{
    kkt_ldl *k = w->kkt;
    const int N = k->N;
    for (int r = 0; r < w->m; r++) k->Kx[k->rho_slot[r]] = -w->rho_inv_vec[r];
    for (int c = 0; c < N; c++) {
        int top = N;
        k->Y[c] = 0.0; k->flag[c] = c; k->Lnz[c] = 0;
        for (int p = k->Kp[c]; p < k->Kp[c + 1]; p++) {
            int i = k->Ki[p], len = 0;
            k->Y[i] += k->Kx[p];
            for (; k->flag[i] != c; i = k->parent[i]) { k->pattern[len++] = i; k->flag[i] = c; }
            while (len > 0) k->pattern[--top] = k->pattern[--len];
        }
        double d = k->Y[c];
        k->Y[c] = 0.0;
        for (; top < N; top++) {
            int i = k->pattern[top];
            double yi = k->Y[i];
            k->Y[i] = 0.0;
            int p2 = k->Lp[i] + k->Lnz[i];
            for (int p = k->Lp[i]; p < p2; p++) k->Y[k->Li[p]] -= k->Lx[p] * yi;
            double lki = yi / k->D[i];
            d -= lki * yi;
            k->Li[p2] = c; k->Lx[p2] = lki; k->Lnz[i]++;
        }
        if (d == 0.0) return 1;
        k->D[c] = d;
    }
    return 0;
}

/* sol <- KKT^-1 rhs (both in the original ordering, length n + m) */
static void kkt_solve(const osqp_work *w, double *v)
{
    kkt_ldl *k = w->kkt;
    const int N = k->N;
    double *b = k->b;
    for (int a = 0; a < N; a++) b[a] = v[k->perm[a]];
    for (int c = 0; c < N; c++) {
        double bc = b[c];
        for (int p = k->Lp[c]; p < k->Lp[c + 1]; p++) b[k->Li[p]] -= k->Lx[p] * bc;
    }
    for (int c = 0; c < N; c++) b[c] /= k->D[c];
    for (int c = N - 1; c >= 0; c--) {
        double s = b[c];
        for (int p = k->Lp[c]; p < k->Lp[c + 1]; p++) s -= k->Lx[p] * b[k->Li[p]];
        b[c] = s;
    }
    for (int a = 0; a < N; a++) v[k->perm[a]] = b[a];
}

/* ---------------------------------------------------------------- setup / update */
osqp_work *osqp_setup(int n, int m, const double *P, const double *A, const double *q,
                      const double *l, const double *u, const osqp_settings *s)
{
    osqp_work *w = (osqp_work *)calloc(1, sizeof(osqp_work));
    w->n = n; w->m = m; w->s = *s;
    if (w->s.adaptive_rho && w->s.adaptive_rho_interval == 0) w->s.adaptive_rho_interval = 50;
    if (w->s.rho < RHO_MIN) w->s.rho = RHO_MIN;
    if (w->s.rho > RHO_MAX) w->s.rho = RHO_MAX;
    w->q = dalloc(n); w->l = dalloc(m); w->u = dalloc(m);
    memcpy(w->q, q, sizeof(double) * n);
    memcpy(w->l, l, sizeof(double) * m);
    memcpy(w->u, u, sizeof(double) * m);
    w->D = dalloc(n); w->Dinv = dalloc(n); w->E = dalloc(m); w->Einv = dalloc(m);
    double *Pd = dalloc(n * n), *Ad = dalloc(m * n);
    memcpy(Pd, P, sizeof(double) * (size_t)n * n);
    memcpy(Ad, A, sizeof(double) * (size_t)m * n);
    if (w->s.scaling > 0) scale_data(w, Pd, Ad);
    else {
        for (int j = 0; j < n; j++) w->D[j] = w->Dinv[j] = 1.0;
        for (int i = 0; i < m; i++) w->E[i] = w->Einv[i] = 1.0;
        w->c = w->cinv = 1.0;
    }
    w->P = csr_from_dense(Pd, n, n);
    w->A = csr_from_dense(Ad, m, n);
    w->At = csr_transpose(&w->A);
    free(Pd); free(Ad);
    w->rho_vec = dalloc(m); w->rho_inv_vec = dalloc(m);
    w->constr_type = (int *)calloc((size_t)m + 1, sizeof(int));
    for (int i = 0; i < m; i++) w->constr_type[i] = classify(w, i);
    apply_rho(w);
    if (w->s.linsys == 1) {
        if (kkt_setup(w)) { osqp_cleanup(w); return NULL; }
        w->kkt_rhs = dalloc(n + m);
    } else {
        w->L = dalloc(n * n);
        if (factor(w)) { osqp_cleanup(w); return NULL; }
    }
    w->x = dalloc(n); w->z = dalloc(m); w->y = dalloc(m);
    w->x_prev = dalloc(n); w->z_prev = dalloc(m);
    w->xt = dalloc(n); w->zt = dalloc(m);
    w->delta_x = dalloc(n); w->delta_y = dalloc(m);
    w->Ax = dalloc(m); w->Px = dalloc(n); w->Aty = dalloc(n);
    w->wn = dalloc(n); w->wm = dalloc(m);
    w->sol_x = dalloc(n); w->sol_y = dalloc(m);
    w->status = ORACLE_UNSOLVED;
    return w;
}

void osqp_cleanup(osqp_work *w)
{
    if (!w) return;
    csr_free(&w->P); csr_free(&w->A); csr_free(&w->At);
    free(w->q); free(w->l); free(w->u); free(w->D); free(w->E); free(w->Dinv); free(w->Einv);
    free(w->rho_vec); free(w->rho_inv_vec); free(w->rho_scale); free(w->constr_type); free(w->L);
    free(w->x); free(w->z); free(w->y); free(w->x_prev); free(w->z_prev); free(w->xt); free(w->zt);
    free(w->delta_x); free(w->delta_y); free(w->Ax); free(w->Px); free(w->Aty); free(w->wn);
    free(w->wm); free(w->sol_x); free(w->sol_y);
    kkt_free(w->kkt); free(w->kkt_rhs);
    for (int p = 0; p < 3; p++) { free(w->phase_scale[p]); free(w->phase_L[p]); }
    free(w);
}

int osqp_update_bounds(osqp_work *w, const double *l, const double *u)
{
    for (int i = 0; i < w->m; i++) if (l[i] > u[i]) return 1;
    for (int i = 0; i < w->m; i++) { w->l[i] = l[i] * w->E[i]; w->u[i] = u[i] * w->E[i]; }
    w->status = ORACLE_UNSOLVED;
    /* constraints whose kind (equality / inequality / unbounded) changed get a new rho */
    int changed = 0;
    for (int i = 0; i < w->m; i++) {
        int t = classify(w, i);
        if (t != w->constr_type[i]) { w->constr_type[i] = t; changed = 1; }
    }
    if (changed) {
        for (int p = 0; p < 3; p++) { free(w->phase_L[p]); w->phase_L[p] = NULL; }   /* factors kept per phase are stale */
        apply_rho(w);
        if (factor(w)) return 2;
    }
    return 0;
}

int osqp_update_lin_cost(osqp_work *w, const double *q)
{
    for (int j = 0; j < w->n; j++) w->q[j] = w->c * w->D[j] * q[j];
    w->status = ORACLE_UNSOLVED;
    return 0;
}

int osqp_set_rho_scale(osqp_work *w, const double *scale)
{
    if (!w->rho_scale) w->rho_scale = dalloc(w->m);
    memcpy(w->rho_scale, scale, sizeof(double) * w->m);
    apply_rho(w);
    return factor(w);
}

/* An instance that is still running at the first termination check at or after iteration
 * start[p] continues with rho_scale = scale + p * m on its inequality rows (p = 1 .. nphase - 1;
 * phase 0 starts every solve).  x, z and y carry over unchanged, as in OSQP's own update_rho.
 * Mirrors the engine's tg_solver_opts::phase_iters / phase_scale. */
static void apply_phase(osqp_work *w, int p);
int osqp_set_rho_phases(osqp_work *w, int nphase, const int *start, const double *scale)
{
    for (int p = 0; p < 3; p++) {
        free(w->phase_scale[p]); free(w->phase_L[p]);
        w->phase_scale[p] = NULL; w->phase_L[p] = NULL;
    }
    w->nphase = 0; w->phase = 0;
    if (nphase <= 1) return 0;
    if (nphase > 3) return 1;
    for (int p = 0; p < nphase; p++) {
        w->phase_start[p] = p ? start[p] : 0;
        w->phase_scale[p] = dalloc(w->m);
        memcpy(w->phase_scale[p], scale + (size_t)p * w->m, sizeof(double) * w->m);
    }
    w->nphase = nphase;
    w->phase = -1;
    apply_phase(w, 0);
    return 0;
}

static void apply_phase(osqp_work *w, int p)
{
    if (w->phase == p) return;
    if (!w->rho_scale) w->rho_scale = dalloc(w->m);
    /* keep the factor of the phase that is left (dense mode) */
    if (!w->kkt && w->phase >= 0 && !w->phase_L[w->phase]) {
        w->phase_L[w->phase] = dalloc(w->n * w->n);
        memcpy(w->phase_L[w->phase], w->L, sizeof(double) * (size_t)w->n * w->n);
    }
    memcpy(w->rho_scale, w->phase_scale[p], sizeof(double) * w->m);
    apply_rho(w);
    if (!w->kkt && w->phase_L[p]) memcpy(w->L, w->phase_L[p], sizeof(double) * (size_t)w->n * w->n);
    else factor(w);
    w->phase = p;
}

void osqp_warm_start(osqp_work *w, const double *x, const double *y)
{
    w->s.warm_start = 1;
    for (int j = 0; j < w->n; j++) w->x[j] = x[j] * w->Dinv[j];
    for (int i = 0; i < w->m; i++) w->y[i] = y[i] * w->Einv[i] * w->c;
    csr_mv(&w->A, w->x, w->z);
}

/* ---------------------------------------------------------------- residuals */
static void update_info(osqp_work *w)
{
    int n = w->n, m = w->m;
    /* primal: z_prev <- A x - z */
    csr_mv(&w->A, w->x, w->Ax);
    for (int i = 0; i < m; i++) w->z_prev[i] = w->Ax[i] - w->z[i];
    w->pri_res = (w->s.scaling && !w->s.scaled_termination) ? scaled_norm_inf(w->Einv, w->z_prev, m)
                                                            : norm_inf(w->z_prev, m);
    /* dual: x_prev <- P x + q + A' y */
    csr_mv(&w->P, w->x, w->Px);
    csr_mv(&w->At, w->y, w->Aty);
    for (int j = 0; j < n; j++) w->x_prev[j] = w->Px[j] + w->q[j] + w->Aty[j];
    w->dua_res = (w->s.scaling && !w->s.scaled_termination)
                     ? w->cinv * scaled_norm_inf(w->Dinv, w->x_prev, n)
                     : norm_inf(w->x_prev, n);
}

static double eps_prim(const osqp_work *w, double eps_abs, double eps_rel)
{
    double a, b;
    if (w->s.scaling && !w->s.scaled_termination) {
        a = scaled_norm_inf(w->Einv, w->z, w->m); b = scaled_norm_inf(w->Einv, w->Ax, w->m);
    } else { a = norm_inf(w->z, w->m); b = norm_inf(w->Ax, w->m); }
    return eps_abs + eps_rel * (a > b ? a : b);
}

static double eps_dual(const osqp_work *w, double eps_abs, double eps_rel)
{
    double a, b, c;
    if (w->s.scaling && !w->s.scaled_termination) {
        a = scaled_norm_inf(w->Dinv, w->q, w->n); b = scaled_norm_inf(w->Dinv, w->Aty, w->n);
        c = scaled_norm_inf(w->Dinv, w->Px, w->n);
        double mx = a > b ? a : b; if (c > mx) mx = c;
        return eps_abs + eps_rel * w->cinv * mx;
    }
    a = norm_inf(w->q, w->n); b = norm_inf(w->Aty, w->n); c = norm_inf(w->Px, w->n);
    double mx = a > b ? a : b; if (c > mx) mx = c;
    return eps_abs + eps_rel * mx;
}

static int is_primal_infeasible(osqp_work *w, double eps)
{
    int m = w->m;
    /* project delta_y onto the polar of the recession cone of [l, u] */
    for (int i = 0; i < m; i++) {
        if (w->u[i] > ORACLE_INFTY * MIN_SCALING) {
            if (w->l[i] < -ORACLE_INFTY * MIN_SCALING) w->delta_y[i] = 0.0;
            else if (w->delta_y[i] > 0.0) w->delta_y[i] = 0.0;
        } else if (w->l[i] < -ORACLE_INFTY * MIN_SCALING) {
            if (w->delta_y[i] < 0.0) w->delta_y[i] = 0.0;
        }
    }
    int unsc = w->s.scaling && !w->s.scaled_termination;
    double nrm = unsc ? scaled_norm_inf(w->E, w->delta_y, m) : norm_inf(w->delta_y, m);
    if (nrm > DIVISION_TOL) {
        double lhs = 0.0;
        for (int i = 0; i < m; i++) {
            double d = w->delta_y[i];
            lhs += w->u[i] * (d > 0.0 ? d : 0.0) + w->l[i] * (d < 0.0 ? d : 0.0);
        }
        if (lhs < -eps * nrm) {
            csr_mv(&w->At, w->delta_y, w->wn);
            double a = unsc ? scaled_norm_inf(w->Dinv, w->wn, w->n) : norm_inf(w->wn, w->n);
            return a < eps * nrm;
        }
    }
    return 0;
}

static int is_dual_infeasible(osqp_work *w, double eps)
{
    int n = w->n, m = w->m;
    int unsc = w->s.scaling && !w->s.scaled_termination;
    double nrm = unsc ? scaled_norm_inf(w->D, w->delta_x, n) : norm_inf(w->delta_x, n);
    double cs = unsc ? w->c : 1.0;
    if (nrm > DIVISION_TOL) {
        double qdx = 0.0;
        for (int j = 0; j < n; j++) qdx += w->q[j] * w->delta_x[j];
        if (qdx < -cs * eps * nrm) {
            csr_mv(&w->P, w->delta_x, w->wn);
            double a = unsc ? scaled_norm_inf(w->Dinv, w->wn, n) : norm_inf(w->wn, n);
            if (a < cs * eps * nrm) {
                csr_mv(&w->A, w->delta_x, w->wm);
                for (int i = 0; i < m; i++) {
                    double v = unsc ? w->Einv[i] * w->wm[i] : w->wm[i];
                    if ((w->u[i] < ORACLE_INFTY * MIN_SCALING && v > eps * nrm) ||
                        (w->l[i] > -ORACLE_INFTY * MIN_SCALING && v < -eps * nrm))
                        return 0;
                }
                return 1;
            }
        }
    }
    return 0;
}

static int check_termination_ex(osqp_work *w, int approximate, int test_infeasibility);
static int check_termination(osqp_work *w, int approximate) { return check_termination_ex(w, approximate, 1); }

static int check_termination_ex(osqp_work *w, int approximate, int test_infeasibility)
{
    double ea = w->s.eps_abs, er = w->s.eps_rel, epi = w->s.eps_prim_inf, edi = w->s.eps_dual_inf;
    if (approximate) { ea *= 10; er *= 10; epi *= 10; edi *= 10; }
    int prim_ok = 0, dual_ok = 0, prim_inf = 0, dual_inf = 0;
    if (w->m == 0) prim_ok = 1;
    else if (w->pri_res < eps_prim(w, ea, er)) prim_ok = 1;
    else if (test_infeasibility) prim_inf = is_primal_infeasible(w, epi);
    if (w->dua_res < eps_dual(w, ea, er)) dual_ok = 1;
    else if (test_infeasibility) dual_inf = is_dual_infeasible(w, edi);
    if (prim_ok && dual_ok) { w->status = approximate ? 2 : ORACLE_SOLVED; return 1; }
    if (prim_inf) { w->status = approximate ? 3 : ORACLE_PRIMAL_INFEASIBLE; return 1; }
    if (dual_inf) { w->status = approximate ? 4 : ORACLE_DUAL_INFEASIBLE; return 1; }
    return 0;
}

/* rho estimate of sec. 5.2; uses the scaled residual vectors left in z_prev / x_prev */
static double rho_estimate(const osqp_work *w)
{
    double pri = norm_inf(w->z_prev, w->m), dua = norm_inf(w->x_prev, w->n);
    double a = norm_inf(w->z, w->m), b = norm_inf(w->Ax, w->m);
    pri /= ((a > b ? a : b) + 1e-10);
    double c0 = norm_inf(w->q, w->n), c1 = norm_inf(w->Aty, w->n), c2 = norm_inf(w->Px, w->n);
    double mx = c0 > c1 ? c0 : c1; if (c2 > mx) mx = c2;
    dua /= (mx + 1e-10);
    double r = w->s.rho * sqrt(pri / (dua + 1e-10));
    if (r < RHO_MIN) r = RHO_MIN;
    if (r > RHO_MAX) r = RHO_MAX;
    return r;
}

/* ---------------------------------------------------------------- solve (Algorithm 1) */
int osqp_solve(osqp_work *w)
{
    int n = w->n, m = w->m;
    const double sigma = w->s.sigma, alpha = w->s.alpha;
    if (!w->s.warm_start) {
        memset(w->x, 0, sizeof(double) * n);
        memset(w->z, 0, sizeof(double) * m);
        memset(w->y, 0, sizeof(double) * m);
    }
    w->status = ORACLE_UNSOLVED;
    w->rho_updates = 0;
    if (w->nphase > 1) apply_phase(w, 0);
    int iter, checked = 0;
    const int trace = getenv("ORACLE_TRACE") != NULL;      /* debugging aid: residuals and thresholds at every check */
    for (iter = 1; iter <= w->s.max_iter; iter++) {
        memcpy(w->x_prev, w->x, sizeof(double) * n);
        memcpy(w->z_prev, w->z, sizeof(double) * m);
        /* xt, zt */
        if (w->kkt) {
            double *r = w->kkt_rhs;
            for (int j = 0; j < n; j++) r[j] = sigma * w->x_prev[j] - w->q[j];
            for (int i = 0; i < m; i++) r[n + i] = w->z_prev[i] - w->rho_inv_vec[i] * w->y[i];
            kkt_solve(w, r);
            memcpy(w->xt, r, sizeof(double) * n);
            for (int i = 0; i < m; i++) w->zt[i] = w->z_prev[i] + w->rho_inv_vec[i] * (r[n + i] - w->y[i]);
        } else {
            for (int i = 0; i < m; i++) w->wm[i] = w->rho_vec[i] * w->z_prev[i] - w->y[i];
            csr_mv(&w->At, w->wm, w->xt);
            for (int j = 0; j < n; j++) w->xt[j] += sigma * w->x_prev[j] - w->q[j];
            chol_solve(w, w->xt);
            csr_mv(&w->A, w->xt, w->zt);
        }
        /* x */
        for (int j = 0; j < n; j++) {
            w->x[j] = alpha * w->xt[j] + (1.0 - alpha) * w->x_prev[j];
            w->delta_x[j] = w->x[j] - w->x_prev[j];
        }
        /* z, y */
        for (int i = 0; i < m; i++) {
            double zr = alpha * w->zt[i] + (1.0 - alpha) * w->z_prev[i];
            double zi = zr + w->rho_inv_vec[i] * w->y[i];
            if (zi < w->l[i]) zi = w->l[i];
            if (zi > w->u[i]) zi = w->u[i];
            w->z[i] = zi;
            w->delta_y[i] = w->rho_vec[i] * (zr - zi);
            w->y[i] += w->delta_y[i];
        }
        checked = w->s.check_termination && (iter % w->s.check_termination == 0);
        if (w->s.check_termination && w->s.check_fine > 0)
            checked = iter >= w->s.check_termination &&
                      (iter - w->s.check_termination) % w->s.check_fine == 0;
        if (checked) {
            update_info(w);
            /* with the finer schedule the infeasibility certificates are tested only where OSQP's
               own schedule has a check */
            int full = w->s.check_fine <= 0 || iter % w->s.check_termination == 0 || iter == w->s.max_iter;
            if (trace)
                fprintf(stderr, "oracle check at %d (phase %d): pri %.17g of %.17g, dua %.17g of %.17g\n", iter, w->phase,
                        w->pri_res, eps_prim(w, w->s.eps_abs, w->s.eps_rel), w->dua_res, eps_dual(w, w->s.eps_abs, w->s.eps_rel));
            if (check_termination_ex(w, 0, full)) break;
            /* step-size schedule: still running at the first check at or after a phase's start */
            if (w->nphase > 1 && iter < w->s.max_iter) {
                int p = w->phase;
                while (p + 1 < w->nphase && iter >= w->phase_start[p + 1]) p++;
                if (p != w->phase) { apply_phase(w, p); w->rho_updates++; }
            }
        }
        if (w->s.adaptive_rho && w->s.adaptive_rho_interval &&
            (iter % w->s.adaptive_rho_interval == 0)) {
            if (!checked) update_info(w);
            double rn = rho_estimate(w);
            if (rn > w->s.rho * w->s.adaptive_rho_tolerance ||
                rn < w->s.rho / w->s.adaptive_rho_tolerance) {
                w->s.rho = rn;
                apply_rho(w);
                factor(w);
                w->rho_updates++;
            }
        }
    }
    if (iter > w->s.max_iter) iter = w->s.max_iter;
    w->iter = iter;
    if (w->status == ORACLE_UNSOLVED) {
        if (!checked) update_info(w);
        if (!check_termination(w, 1)) w->status = ORACLE_MAX_ITER;
    }
    /* store the unscaled solution */
    if (w->status == ORACLE_PRIMAL_INFEASIBLE || w->status == 3) {
        for (int j = 0; j < n; j++) w->sol_x[j] = NAN;
        for (int i = 0; i < m; i++) w->sol_y[i] = w->E[i] * w->delta_y[i];
    } else if (w->status == ORACLE_DUAL_INFEASIBLE || w->status == 4) {
        for (int j = 0; j < n; j++) w->sol_x[j] = w->D[j] * w->delta_x[j];
        for (int i = 0; i < m; i++) w->sol_y[i] = NAN;
    } else {
        for (int j = 0; j < n; j++) w->sol_x[j] = w->D[j] * w->x[j];
        for (int i = 0; i < m; i++) w->sol_y[i] = w->cinv * w->E[i] * w->y[i];
    }
    return w->status;
}

void osqp_get_solution(const osqp_work *w, double *x, double *y)
{
    if (x) memcpy(x, w->sol_x, sizeof(double) * w->n);
    if (y) memcpy(y, w->sol_y, sizeof(double) * w->m);
}
int osqp_get_iter(const osqp_work *w) { return w->iter; }
int osqp_get_rho_updates(const osqp_work *w) { return w->rho_updates; }
double osqp_get_rho(const osqp_work *w) { return w->s.rho; }
void osqp_get_residuals(const osqp_work *w, double *pri, double *dua)
{
    if (pri) *pri = w->pri_res;
    if (dua) *dua = w->dua_res;
}
void osqp_get_scaling(const osqp_work *w, double *D, double *E, double *c)
{
    if (D) memcpy(D, w->D, sizeof(double) * w->n);
    if (E) memcpy(E, w->E, sizeof(double) * w->m);
    if (c) *c = w->c;
}
void osqp_get_rho_vec(const osqp_work *w, double *rho_vec)
{
    memcpy(rho_vec, w->rho_vec, sizeof(double) * w->m);
}
