// Host-side construction of one horizon QP "stage" (6-DOF, or the XY / Z / Yaw stage of
// the 12-state model) and of everything the device kernels need that is shared by all
// instances of a batch: the reduced-KKT inverse, the stencil tables of A, A' and P in the
// kernel's internal ordering, bound / rho templates and the I/O index maps.
//
// Reference for the QP itself: src/mpc_6dof.cpp:243-519 (class MPC) and
// src/mpc_12state.cpp:74-1014 (class MPC12STATE).  Variable order of the reference:
//   [x(0); ...; x(N); u(0); ...; u(N-1)]
// row order: [dynamics nx(N+1) | state box nx(N+1) | input box nu*N | (XY) mixed 6N].
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/tg_mpc.h"
#include "device_types.h"

namespace tg {

constexpr double kRhoMin = 1e-6;          // OSQP: step size of rows with both bounds infinite
constexpr double kRhoEqFactor = 1e3;      // OSQP: equality rows get 1e3 * rho
constexpr double kRhoTol = 1e-4;          // OSQP: u - l below this => equality row
constexpr double kInfThresh = 1e30 * 1e-4; // OSQP: |bound| beyond INFTY*MIN_SCALING is infinite
constexpr int kTile = 16;                 // instances per workgroup = MFMA column count

struct StageModel {
    int family = TG_FAM_6DOF;
    int N = 0, nx = 0, nu = 0, n_mixed = 0;
    double Ad[36] = {0}, Bd[18] = {0};     // row-major nx x nx, nx x nu
    double qdiag[6] = {0};                 // diagonal of Q (the reference only sets diagonal entries)
    double r = 0;                          // R = r * I
    bool smoothing = false;
    double smooth_w = 0;
    double xmax[6] = {0}, xmin[6] = {0};   // state box (all steps, step 0 included)
    double umax[3] = {0}, umin[3] = {0};
    double mixed_max[6] = {0};             // XY mixed rows, symmetric
    // where the stage reads x0 / ref and writes X / U in the model-level arrays
    int x0_row[6] = {0};                   // row of component a in the x0 array
    int ref_stride = 0, ref_off = 0;       // ref row of (step i, comp a) = i*ref_stride + ref_off + a
    bool ref_pos_only = false;             // yaw: ref array is [N+1] holding only the angle
    int X_stride = 0, X_off = 0;           // X row of (step i, comp a) = i*X_stride + X_off + a
    int U_stride = 0, U_off = 0;
};

struct StageProblem {
    StageModel mdl;
    int n = 0, m = 0, Nx = 0, Nu = 0;
    double rho = 0, sigma = 0, rho_scale[4] = {1, 1, 1, 1};
    // reference-ordered data (getters, assembly kernel)
    std::vector<double> P, A;              // dense row-major n x n, m x n
    std::vector<double> l_t, u_t, rho_vec; // m
    std::vector<double> Kinv;              // dense n x n, reference order
    // internal (kernel) ordering: axis-major, step-major cells of D states + 1 input
    int D = 0, nax = 0, nblk = 0, nb = 0, nops = 0, n_int = 0;
    int axis_base[kMaxAxes] = {0, 0, 0};
    std::vector<int> op_of_block;
    std::vector<int> var_ref;              // n_int -> reference variable or -1
    std::vector<double> Mfrag;             // [nops][ceil(nb/16)][ceil(nb/4) + kRingDepth][64]
    // Block-row low-rank form of the same operators (long horizons).  K is block tridiagonal, so a
    // tile row of K^-1 outside its 16 x 16 diagonal tile D_I has rank <= 2 D (D integrator states
    // couple a step to either side): K^-1[I, :] = D_I (+) U_I C_I with U_I 16 x q orthonormal.
    //   x_I = D_I r_I + U_I (C_I r)
    // costs ceil(nrb q / 16) nb/4 + nrb (4 + q/4) MFMAs per operator instead of nrb nb/4.
    int lr_q = 0, lr_crb = 0, lr_uk = 0;   // q = 2 D; row blocks of the stacked C; k-steps of U
    std::vector<double> Cfrag;             // [nops][lr_crb][ceil(nb/4) + kRingDepth][64]
    std::vector<double> DUfrag;            // [nops][ceil(nb/16)][4 + lr_uk][64]: D_I then U_I
    double rho_eq = 0;
    double rho_box[kMaxAxes][kMaxSlots] = {}, lb[kMaxAxes][kMaxSlots] = {}, ub[kMaxAxes][kMaxSlots] = {};
    double rho_mx[6] = {0}, ub_mx[6] = {0};
    // reference-order tables for the assembly kernel
    std::vector<double> q_w;               // n
    std::vector<int32_t> q_ref_row;        // n
    std::vector<int32_t> asm_x0_row, asm_ov_box, asm_ov_mixed; // m
};

inline double default_rho(const StageModel &mdl)
{
    // Fixed step size used in place of OSQP's adaptive rho on the equilibrated problem: half the
    // geometric mean of the largest state weight and the input weight, rounded to 1-2-5
    // (7000/500 -> 1000, 7000/200 and 5000/200 -> 500, 1000/100 -> 200).  The factor and the
    // per-class multipliers below come from the same sweeps over seeded instances.
    double w = 0;
    for (int a = 0; a < mdl.nx; a++) w = std::max(w, mdl.qdiag[a]);
    double g = 0.5 * std::sqrt(std::max(w, 1e-12) * std::max(mdl.r, 1e-12));
    if (!(g > 0)) return 1.0;
    double e = std::pow(10.0, std::floor(std::log10(g)));
    double f = g / e;
    double s = f < 1.5 ? 1 : f < 3.5 ? 2 : f < 7.5 ? 5 : 10;
    return std::min(std::max(s * e, 1e-3), 1e6 / kRhoEqFactor);
}

// dense symmetric positive definite inverse via Cholesky, long double accumulation
inline bool spd_inverse(const std::vector<double> &K, int n, std::vector<double> &Kinv)
{
    std::vector<long double> L((size_t)n * n, 0.0L);
    for (int j = 0; j < n; j++) {
        long double d = K[(size_t)j * n + j];
        for (int k = 0; k < j; k++) d -= L[(size_t)j * n + k] * L[(size_t)j * n + k];
        if (!(d > 0)) return false;
        d = sqrtl(d);
        L[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            long double s = K[(size_t)i * n + j];
            for (int k = 0; k < j; k++) s -= L[(size_t)i * n + k] * L[(size_t)j * n + k];
            L[(size_t)i * n + j] = s / d;
        }
    }
    // Linv (lower), then Kinv = Linv' * Linv
    std::vector<long double> Li((size_t)n * n, 0.0L);
    for (int c = 0; c < n; c++) {
        Li[(size_t)c * n + c] = 1.0L / L[(size_t)c * n + c];
        for (int i = c + 1; i < n; i++) {
            long double s = 0;
            for (int k = c; k < i; k++) s -= L[(size_t)i * n + k] * Li[(size_t)k * n + c];
            Li[(size_t)i * n + c] = s / L[(size_t)i * n + i];
        }
    }
    Kinv.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; i++)
        for (int j = 0; j <= i; j++) {
            long double s = 0;
            for (int k = i; k < n; k++) s += Li[(size_t)k * n + i] * Li[(size_t)k * n + j];
            Kinv[(size_t)i * n + j] = Kinv[(size_t)j * n + i] = (double)s;
        }
    return true;
}

// Orthonormal basis U (rows x q, row-major) of the column space of R (rows x cols, row-major) by
// Gram-Schmidt with column pivoting and one re-orthogonalisation, and C = U' R (q x cols).
// Directions below tol * (largest column norm) are left as zero columns of U / rows of C.
inline void column_space(const std::vector<long double> &R, int rows, int cols, int q, long double tol,
                         std::vector<long double> &U, std::vector<long double> &C)
{
    std::vector<long double> W = R;
    U.assign((size_t)rows * q, 0.0L);
    C.assign((size_t)q * cols, 0.0L);
    long double first = 0.0L;
    for (int k = 0; k < q; k++) {
        int best = -1; long double bn = 0.0L;
        for (int j = 0; j < cols; j++) {
            long double nn = 0.0L;
            for (int i = 0; i < rows; i++) nn += W[(size_t)i * cols + j] * W[(size_t)i * cols + j];
            if (nn > bn) { bn = nn; best = j; }
        }
        bn = sqrtl(bn);
        if (k == 0) first = bn;
        if (best < 0 || !(bn > tol * first) || bn == 0.0L) break;
        std::vector<long double> u(rows);
        for (int i = 0; i < rows; i++) u[i] = W[(size_t)i * cols + best];
        for (int pass = 0; pass < 2; pass++)
            for (int p2 = 0; p2 < k; p2++) {
                long double d = 0.0L;
                for (int i = 0; i < rows; i++) d += U[(size_t)i * q + p2] * u[i];
                for (int i = 0; i < rows; i++) u[i] -= d * U[(size_t)i * q + p2];
            }
        long double nn = 0.0L;
        for (int i = 0; i < rows; i++) nn += u[i] * u[i];
        nn = sqrtl(nn);
        if (!(nn > 0.0L)) break;
        for (int i = 0; i < rows; i++) { u[i] /= nn; U[(size_t)i * q + k] = u[i]; }
        for (int j = 0; j < cols; j++) {
            long double d = 0.0L;
            for (int i = 0; i < rows; i++) d += u[i] * W[(size_t)i * cols + j];
            for (int i = 0; i < rows; i++) W[(size_t)i * cols + j] -= d * u[i];
        }
    }
    for (int k = 0; k < q; k++)
        for (int j = 0; j < cols; j++) {
            long double d = 0.0L;
            for (int i = 0; i < rows; i++) d += U[(size_t)i * q + k] * R[(size_t)i * cols + j];
            C[(size_t)k * cols + j] = d;
        }
}

// Built-in multipliers of rho per class of box row (position, velocity, acceleration, input).
// A uniform rho leaves a long iteration tail on instances whose position or input rows are
// active (the 6-DOF altitude floor, saturated inputs): the step those rows need is set by the
// state / input weights, not by their geometric mean.  Values picked from sweeps over seeded
// random instances at N = 20 and N = 50 (held-out seeds; tools/rho_policy_probe.py runs such sweeps):
// 6-DOF N = 20: slowest instance 375 -> 30 iterations at eps 1e-3, 99.9 % done at the first check
// (iteration 25).  N = 50: (12, 3, 1, 4) passes the first check with 93.5 % of the instances, mean
// 26.8 iterations, slowest of 8192 at 125-225 depending on the seed; a larger step on the altitude
// floor rows, (20, 3, 1, 5), shortens the tail to 100 but only 73 % pass the first check (mean 31.8).
// With the batch's instance slots refilled from a queue the mean sets the throughput.
inline void default_rho_scale(const StageModel &mdl, double out[4])
{
    if (mdl.family == TG_FAM_6DOF) {
        if (mdl.N <= 30) { out[0] = 15.0; out[1] = 1.0; out[2] = 1.0; out[3] = 3.0; }
        else { out[0] = 12.0; out[1] = 3.0; out[2] = 1.0; out[3] = 4.0; }
    }
    else { out[0] = 1.0; out[1] = 10.0; out[2] = 3.0; out[3] = 3.0; }   // slowest XY / yaw instance 2466 / 297 -> 109 / 94
}

// Built-in step-size schedule: iteration counts at which an instance that is still running moves
// to the next operator, and that operator's class multipliers relative to the first one's.
// iters[p] <= 0: no such phase.
// The class multipliers above are tuned for the bulk of the instances, which pass OSQP's default
// tolerance at the first or second check.  The instances that take long, and everything run to
// a tight tolerance, are held up by the altitude floor rows of the 6-DOF model: their iteration
// count falls in proportion to a larger step there, which in turn would slow the bulk down.  So
// the step on those rows grows only for instances still running after 100 and 200 iterations
// (x 8, x 64).  N = 50, 16384 seeded instances: at eps 3e-9 mean 132 / slowest 1200 iterations
// against 324 / 4775 with the best single operator; at eps 1e-3 the mean stays 26.8 and the
// slowest instance needs 125 instead of 300 (tools/rho_policy_probe.py).
inline void default_phases(const StageModel &mdl, int iters[2], double scale[2][4])
{
    for (int p = 0; p < 2; p++) {
        iters[p] = -1;
        for (int k = 0; k < 4; k++) scale[p][k] = 1.0;
    }
    if (mdl.family == TG_FAM_6DOF) {
        if (mdl.N <= 30) {
            // Short horizons have no such tail in a single solve (N = 20 needs 75 iterations at eps 1e-9,
            // 125 at most), but a closed loop has: a vehicle that the plant carries a few centimetres
            // below the altitude floor poses a problem that is infeasible by that much, and at a fixed
            // step 4000 iterations neither solve nor certify it (8192 vehicles x 500 steps: 0.002 % of the
            // solves, which made a step 1.14 ms instead of 0.21).  One later operator, far out, ends those
            // after 175-275 iterations; an earlier boundary (100) would slow tight single solves down
            // (slowest of 2048 at eps 1e-9: 350 against 125), and every boundary costs a batch of 4096 an
            // empty continuation launch (6 us of 117).
            iters[0] = 150; scale[0][0] = 64.0;
            scale[1][0] = 64.0;
            return;
        }
        iters[0] = 100; scale[0][0] = 8.0;
        iters[1] = 200; scale[1][0] = 64.0;
    } else if (mdl.family == TG_FAM_XY) {
        // the acceleration rows (box and octagon): N = 20, eps 1e-7: mean 203 against 285 iterations,
        // 99th percentile 300 against 775; a larger step on the speed rows slows this stage down
        iters[0] = 100; scale[0][2] = 3.0;
        iters[1] = 200; scale[1][2] = 10.0;
    } else {
        // Z and yaw stages: the speed rows.  N = 20, 4096 seeded instances, eps 1e-7: yaw mean 180 /
        // slowest 700 iterations against 1180 / more than 8000 with the first operator alone, Z 91 /
        // 400 against 160 / 7800; eps 1e-5: yaw 122 / 475 against 455 / 5825.  At eps 1e-3 nothing
        // changes in the mean (the bulk is done by iteration 75).
        iters[0] = 100; scale[0][1] = 10.0;
        iters[1] = 200; scale[1][1] = 100.0;
    }
}

inline bool build_stage(const StageModel &mdl, double rho, double sigma, const double rho_scale_in[4],
                        StageProblem &sp, std::string &err)
{
    sp = StageProblem();
    sp.mdl = mdl;
    const int N = mdl.N, nx = mdl.nx, nu = mdl.nu;
    const int Nx = nx * (N + 1), Nu = nu * N, n = Nx + Nu, m = 2 * Nx + Nu + mdl.n_mixed * N;
    sp.n = n; sp.m = m; sp.Nx = Nx; sp.Nu = Nu;
    sp.rho = rho > 0 ? rho : default_rho(mdl);
    sp.rho = std::min(std::max(sp.rho, kRhoMin), 1e6);
    sp.sigma = sigma;
    // ---- P (castMPCToQPHessian) and A (castMPCToQPConstraintMatrix), dense like the reference
    sp.P.assign((size_t)n * n, 0.0);
    sp.A.assign((size_t)m * n, 0.0);
    auto Pm = [&](int i, int j) -> double & { return sp.P[(size_t)i * n + j]; };
    auto Am = [&](int i, int j) -> double & { return sp.A[(size_t)i * n + j]; };
    for (int i = 0; i <= N; i++)
        for (int a = 0; a < nx; a++) Pm(nx * i + a, nx * i + a) = mdl.qdiag[a];
    for (int i = 0; i < N; i++)
        for (int a = 0; a < nu; a++) Pm(Nx + nu * i + a, Nx + nu * i + a) = mdl.r;
    if (mdl.smoothing)
        for (int i = 0; i + 1 < N; i++)
            for (int a = 0; a < nu; a++) {
                int r0 = Nx + nu * i + a, r1 = r0 + nu;
                Pm(r0, r0) += mdl.smooth_w; Pm(r1, r1) += mdl.smooth_w;
                Pm(r0, r1) -= mdl.smooth_w; Pm(r1, r0) -= mdl.smooth_w;
            }
    for (int r = 0; r < Nx; r++) Am(r, r) = -1.0;
    for (int i = 1; i <= N; i++)
        for (int a = 0; a < nx; a++) {
            for (int b = 0; b < nx; b++) Am(nx * i + a, nx * (i - 1) + b) = mdl.Ad[a * nx + b];
            for (int b = 0; b < nu; b++) Am(nx * i + a, Nx + nu * (i - 1) + b) = mdl.Bd[a * nu + b];
        }
    for (int r = 0; r < n; r++) Am(Nx + r, r) = 1.0;
    if (mdl.n_mixed) {
        const double s3 = std::sqrt(3.0) / 2.0, s2 = std::sqrt(2.0) / 2.0;
        for (int i = 0; i < N; i++) {
            int r = 2 * Nx + Nu + 6 * i, c = nx * (i + 1);
            Am(r + 0, c + 1) = s3;   Am(r + 0, c + 4) = 0.5;
            Am(r + 1, c + 1) = s3;   Am(r + 1, c + 4) = -0.5;
            Am(r + 2, c + 1) = 0.5;  Am(r + 2, c + 4) = s3;
            Am(r + 3, c + 1) = -0.5; Am(r + 3, c + 4) = s3;
            Am(r + 4, c + 2) = s2;   Am(r + 4, c + 5) = s2;
            Am(r + 5, c + 2) = s2;   Am(r + 5, c + 5) = -s2;
        }
    }
    // ---- bound templates (x0 = 0) and per-row rho
    sp.l_t.assign(m, 0.0); sp.u_t.assign(m, 0.0); sp.rho_vec.assign(m, 0.0);
    for (int i = 0; i <= N; i++)
        for (int a = 0; a < nx; a++) { sp.l_t[Nx + nx * i + a] = mdl.xmin[a]; sp.u_t[Nx + nx * i + a] = mdl.xmax[a]; }
    for (int i = 0; i < N; i++)
        for (int a = 0; a < nu; a++) { sp.l_t[2 * Nx + nu * i + a] = mdl.umin[a]; sp.u_t[2 * Nx + nu * i + a] = mdl.umax[a]; }
    for (int i = 0; i < N; i++)
        for (int k = 0; k < mdl.n_mixed; k++) {
            sp.l_t[2 * Nx + Nu + 6 * i + k] = -mdl.mixed_max[k]; sp.u_t[2 * Nx + Nu + 6 * i + k] = mdl.mixed_max[k];
        }
    for (int r = 0; r < m; r++) {
        double l = sp.l_t[r], u = sp.u_t[r];
        if (l > u) { err = "lower bound above upper bound in the constraint template"; return false; }
        if (l < -kInfThresh && u > kInfThresh) sp.rho_vec[r] = kRhoMin;
        else if (u - l < kRhoTol) sp.rho_vec[r] = kRhoEqFactor * sp.rho;
        else sp.rho_vec[r] = sp.rho;
    }
    // per-class scaling of the inequality rows' rho
    {
        double cls[4];
        default_rho_scale(mdl, cls);
        for (int k = 0; k < 4; k++) if (rho_scale_in && rho_scale_in[k] > 0) cls[k] = rho_scale_in[k];
        for (int k = 0; k < 4; k++) sp.rho_scale[k] = cls[k];
        auto scale_row = [&](int r, double f) { if (sp.rho_vec[r] == sp.rho) sp.rho_vec[r] = f * sp.rho; };
        for (int i = 0; i <= N; i++)
            for (int a = 0; a < nx; a++) {
                int c = mdl.family == TG_FAM_6DOF ? (a < 3 ? 0 : 1) : a % 3;
                scale_row(Nx + nx * i + a, cls[c]);
            }
        for (int r = 0; r < Nu; r++) scale_row(2 * Nx + r, cls[3]);
        for (int r = 2 * Nx + Nu; r < m; r++) scale_row(r, cls[(r - 2 * Nx - Nu) % 6 < 4 ? 1 : 2]);
    }
    // ---- reduced KKT matrix K = P + sigma I + A' diag(rho) A  and its inverse
    std::vector<double> K(sp.P);
    for (int j = 0; j < n; j++) K[(size_t)j * n + j] += sigma;
    {
        std::vector<int> cols; std::vector<double> vals;
        for (int r = 0; r < m; r++) {
            cols.clear(); vals.clear();
            for (int j = 0; j < n; j++) if (Am(r, j) != 0.0) { cols.push_back(j); vals.push_back(Am(r, j)); }
            for (size_t a = 0; a < cols.size(); a++)
                for (size_t b = 0; b < cols.size(); b++)
                    K[(size_t)cols[a] * n + cols[b]] += sp.rho_vec[r] * vals[a] * vals[b];
        }
    }
    if (!spd_inverse(K, n, sp.Kinv)) { err = "reduced KKT matrix is not positive definite"; return false; }
    // ---- kernel ordering: (axis a, step i, slot k) -> axis_base[a] + i*(D+1) + k
    const bool six = mdl.family == TG_FAM_6DOF, xy = mdl.family == TG_FAM_XY;
    const int D = six ? 2 : 3, S = D + 1, nax = six ? 3 : (xy ? 2 : 1);
    sp.D = D; sp.nax = nax;
    sp.nblk = nax;
    const int per_axis = S * (N + 1);
    sp.nb = per_axis;                           // rows of one block, also the stride between blocks
    sp.n_int = sp.nblk * sp.nb;
    for (int a = 0; a < nax; a++) sp.axis_base[a] = a * sp.nb;
    sp.var_ref.assign(sp.n_int, -1);
    std::vector<int> blk_of(n, -1);
    for (int a = 0; a < nax; a++)
        for (int i = 0; i <= N; i++)
            for (int k = 0; k < S; k++) {
                int ref;
                if (k < D) ref = nx * i + (six ? a + 3 * k : (xy ? 3 * a + k : k));
                else if (i < N) ref = Nx + nu * i + a;
                else continue;
                int ji = sp.axis_base[a] + i * S + k;
                sp.var_ref[ji] = ref;
                blk_of[ref] = ji / sp.nb;
            }
    for (int j = 0; j < n; j++) if (blk_of[j] < 0) { err = "internal: variable not mapped"; return false; }
    // the blocks must be decoupled in K (6-DOF: the three axes never interact); the XY stage's
    // mixed rows couple its two axes, see below
    for (int i = 0; i < n && !xy; i++)
        for (int j = 0; j < n; j++)
            if (blk_of[i] != blk_of[j] && K[(size_t)i * n + j] != 0.0) { err = "internal: axes are coupled"; return false; }
    // operators: blocks whose K sub-matrices coincide share one inverse
    // reference variable of local row li of block b; rows past the block (MFMA padding) have none
    auto vref = [&](int b, int li) -> int { return li < sp.nb ? sp.var_ref[b * sp.nb + li] : -1; };
    auto Kint = [&](int b, int i, int j) -> double {
        int ri = sp.var_ref[b * sp.nb + i], rj = sp.var_ref[b * sp.nb + j];
        return (ri >= 0 && rj >= 0) ? K[(size_t)ri * n + rj] : 0.0;
    };
    sp.op_of_block.assign(sp.nblk, -1);
    std::vector<int> op_rep;
    for (int b = 0; b < sp.nblk; b++) {
        for (size_t o = 0; o < op_rep.size() && sp.op_of_block[b] < 0; o++) {
            bool same = true;
            for (int i = 0; i < sp.nb && same; i++)
                for (int j = 0; j < sp.nb; j++) if (Kint(b, i, j) != Kint(op_rep[o], i, j)) { same = false; break; }
            if (same) sp.op_of_block[b] = (int)o;
        }
        if (sp.op_of_block[b] < 0) { sp.op_of_block[b] = (int)op_rep.size(); op_rep.push_back(b); }
    }
    sp.nops = (int)op_rep.size();
    // XY: x and y are interchangeable in K (same weights and step sizes, and a swap of the axes
    // maps the set of mixed rows onto itself), so K^-1 = [[E, F], [F, E]] and the kernel applies
    // (E + F) / 2 to the sum and (E - F) / 2 to the difference of the two axes' right-hand sides:
    // two products of half the size instead of one.
    if (xy) {
        double worst = 0.0, scale = 0.0;
        for (int i = 0; i < sp.nb; i++)
            for (int j = 0; j < sp.nb; j++) {
                int xi = vref(0, i), xj = vref(0, j), yi = vref(1, i), yj = vref(1, j);
                if ((xi < 0) != (yi < 0) || (xj < 0) != (yj < 0)) { err = "internal: XY axes are laid out differently"; return false; }
                if (xi < 0 || xj < 0) continue;
                double e = sp.Kinv[(size_t)xi * n + xj], f = sp.Kinv[(size_t)xi * n + yj];
                worst = std::max(worst, std::fabs(e - sp.Kinv[(size_t)yi * n + yj]));
                worst = std::max(worst, std::fabs(f - sp.Kinv[(size_t)yi * n + xj]));
                scale = std::max(scale, std::fabs(e));
            }
        if (!(worst <= 1e-9 * scale)) { err = "internal: the XY operator is not symmetric in its two axes"; return false; }
        sp.nops = 2; sp.op_of_block[0] = 0; sp.op_of_block[1] = 1;
    }
    // entry (i, j) of operator o (rows / columns past the block are MFMA padding)
    auto opval = [&](int o, int i, int j) -> double {
        const int b = xy ? 0 : op_rep[o];
        const int ri = vref(b, i), rj = vref(b, j);
        if (ri < 0 || rj < 0) return 0.0;
        double v = sp.Kinv[(size_t)ri * n + rj];
        if (xy) {
            const double f = sp.Kinv[(size_t)ri * n + vref(1, j)];
            v = 0.5 * (o == 0 ? v + f : v - f);
        }
        return v;
    };
    // operator fragments: Mfrag[op][rb][kk][lane] = op[rb*16 + (lane&15)][4*kk + (lane>>4)]
    const int nrb = (sp.nb + 15) / 16, ks = (sp.nb + 3) / 4, ksa = ks + kRingDepth;   // zero steps: prefetch overrun
    sp.Mfrag.assign((size_t)sp.nops * nrb * ksa * 64, 0.0);
    for (int o = 0; o < sp.nops; o++)
        for (int rb = 0; rb < nrb; rb++)
            for (int kk = 0; kk < ks; kk++)
                for (int lane = 0; lane < 64; lane++)
                    sp.Mfrag[(((size_t)o * nrb + rb) * ksa + kk) * 64 + lane] = opval(o, rb * 16 + (lane & 15), 4 * kk + (lane >> 4));
    // block-row low-rank form; left empty (lr_q = 0) when the computed inverse does not have it to
    // working precision, and the dense fragments are used
    {
        // smoothing ties an input to its neighbours as well: one more coupling variable per side
        const int q = 2 * (D + (mdl.smoothing ? 1 : 0)), crb = (nrb * q + 15) / 16, uk = (q + 3) / 4;
        std::vector<double> Cf((size_t)sp.nops * crb * ksa * 64, 0.0), DUf((size_t)sp.nops * nrb * (4 + uk) * 64, 0.0);
        bool ok = true;
        for (int o = 0; o < sp.nops && ok; o++) {
            long double scale = 0.0L;
            for (int i = 0; i < sp.nb; i++) scale = std::max(scale, (long double)std::fabs(opval(o, i, i)));
            std::vector<long double> Cst((size_t)nrb * q * sp.nb, 0.0L);   // stacked C, row I*q + k
            for (int I = 0; I < nrb && ok; I++) {
                const int r0 = I * 16, rows = std::min(16, sp.nb - r0);
                std::vector<long double> R((size_t)rows * sp.nb, 0.0L), U, C;
                for (int i = 0; i < rows; i++)
                    for (int j = 0; j < sp.nb; j++)
                        if (j < r0 || j >= r0 + 16) R[(size_t)i * sp.nb + j] = opval(o, r0 + i, j);
                column_space(R, rows, sp.nb, q, 1e-13L, U, C);
                long double worst = 0.0L;
                for (int i = 0; i < rows; i++)
                    for (int j = 0; j < sp.nb; j++) {
                        long double v = 0.0L;
                        for (int k = 0; k < q; k++) v += U[(size_t)i * q + k] * C[(size_t)k * sp.nb + j];
                        worst = std::max(worst, fabsl(v - R[(size_t)i * sp.nb + j]));
                    }
                if (!(worst <= 1e-13L * scale)) { ok = false; break; }
                for (int k = 0; k < q; k++)
                    for (int j = 0; j < sp.nb; j++) Cst[((size_t)I * q + k) * sp.nb + j] = C[(size_t)k * sp.nb + j];
                // D_I (4 k-steps over the tile's own columns) followed by U_I (uk k-steps over its q coefficients)
                for (int kk = 0; kk < 4 + uk; kk++)
                    for (int lane = 0; lane < 64; lane++) {
                        const int i = lane & 15, c = 4 * (kk < 4 ? kk : kk - 4) + (lane >> 4);
                        double v = 0.0;
                        if (kk < 4) v = opval(o, r0 + i, r0 + c);
                        else if (i < rows && c < q) v = (double)U[(size_t)i * q + c];
                        DUf[(((size_t)o * nrb + I) * (4 + uk) + kk) * 64 + lane] = v;
                    }
            }
            for (int rb = 0; rb < crb && ok; rb++)
                for (int kk = 0; kk < ks; kk++)
                    for (int lane = 0; lane < 64; lane++) {
                        const int i = rb * 16 + (lane & 15), j = 4 * kk + (lane >> 4);
                        Cf[(((size_t)o * crb + rb) * ksa + kk) * 64 + lane] =
                            (i < nrb * q && j < sp.nb) ? (double)Cst[(size_t)i * sp.nb + j] : 0.0;
                    }
        }
        if (ok) { sp.lr_q = q; sp.lr_crb = crb; sp.lr_uk = uk; sp.Cfrag.swap(Cf); sp.DUfrag.swap(DUf); }
    }
    // per-axis, per-slot step sizes and box templates (identical at every step)
    sp.rho_eq = sp.rho_vec[0];
    for (int r = 0; r < Nx; r++) if (sp.rho_vec[r] != sp.rho_eq) { err = "internal: dynamics rows must share one rho"; return false; }
    for (int a = 0; a < nax; a++)
        for (int k = 0; k < S; k++) {
            int ref = sp.var_ref[sp.axis_base[a] + k];   // step 0
            int brow = Nx + ref;
            sp.rho_box[a][k] = sp.rho_vec[brow]; sp.lb[a][k] = sp.l_t[brow]; sp.ub[a][k] = sp.u_t[brow];
        }
    for (int k = 0; k < mdl.n_mixed; k++) { sp.rho_mx[k] = sp.rho_vec[2 * Nx + Nu + k]; sp.ub_mx[k] = sp.u_t[2 * Nx + Nu + k]; }
    // ---- reference-order tables for the assembly kernel
    sp.q_w.assign(n, 0.0); sp.q_ref_row.assign(n, -1);
    sp.asm_x0_row.assign(m, -1); sp.asm_ov_box.assign(m, -1); sp.asm_ov_mixed.assign(m, -1);
    for (int j = 0; j < Nx; j++) {
        int i = j / nx, a = j % nx;
        if (mdl.qdiag[a] != 0.0 && (!mdl.ref_pos_only || a == 0)) {
            sp.q_w[j] = -1.0 * mdl.qdiag[a];
            sp.q_ref_row[j] = mdl.ref_pos_only ? i : i * mdl.ref_stride + mdl.ref_off + a;
        }
        if (xy && i >= 1) sp.asm_ov_box[Nx + j] = nx * (i - 1) + a;
    }
    for (int r = 0; r < nx; r++) sp.asm_x0_row[r] = mdl.x0_row[r];
    for (int r = 2 * Nx + Nu; r < m; r++) sp.asm_ov_mixed[r] = r - (2 * Nx + Nu);
    return true;
}

// ---------------------------------------------------------------- model factories
inline StageModel model_6dof(const tg_mpc6_params &p)
{
    // setTransitionMatrix mpc_6dof.cpp:274, setInputMatrix :287, setQ :244, setR :260,
    // setStateBounds :299, setControlBounds :320
    StageModel s;
    s.family = TG_FAM_6DOF; s.N = p.N; s.nx = 6; s.nu = 3; s.n_mixed = 0;
    for (int i = 0; i < 6; i++) s.Ad[i * 6 + i] = 1.0;
    for (int i = 0; i < 3; i++) {
        s.Ad[i * 6 + 3 + i] = p.dt;
        s.Bd[(3 + i) * 3 + i] = p.dt;
        s.qdiag[i] = p.state_weight;
        s.xmin[3 + i] = -1.0 * p.max_vel[i]; s.xmax[3 + i] = p.max_vel[i];
        s.umin[i] = -1.0 * p.max_accel[i];   s.umax[i] = p.max_accel[i];
    }
    s.r = p.input_weight;
    s.smoothing = p.enable_control_smoothing != 0; s.smooth_w = p.smooth_input_weight;
    s.xmin[0] = s.xmin[1] = -TG_INFTY; s.xmax[0] = s.xmax[1] = s.xmax[2] = TG_INFTY;
    s.xmin[2] = p.min_altitude;
    for (int a = 0; a < 6; a++) s.x0_row[a] = a;
    s.ref_stride = 6; s.ref_off = 0; s.X_stride = 6; s.X_off = 0; s.U_stride = 3; s.U_off = 0;
    return s;
}

inline StageModel model_12(const tg_mpc12_params &p, int family)
{
    // mpc_12state.cpp:75-310 (Q, R, A, B, bounds per stage), :600-630 (mixed rows)
    StageModel s;
    s.family = family; s.N = p.N;
    const double dt = p.dt;
    const double A3[9] = {1, dt, dt * dt / 2.0, 0, 1, dt, 0, 0, 1};
    s.smoothing = p.enable_control_smoothing != 0;
    if (family == TG_FAM_XY) {
        s.nx = 6; s.nu = 2; s.n_mixed = 6;
        for (int b = 0; b < 2; b++)
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) s.Ad[(3 * b + i) * 6 + 3 * b + j] = A3[i * 3 + j];
        s.Bd[2 * 2 + 0] = dt; s.Bd[5 * 2 + 1] = dt;
        s.qdiag[0] = s.qdiag[3] = p.xy_state_weight;
        s.r = p.xy_input_weight; s.smooth_w = p.xy_smooth_input_weight;
        const double b0[6] = {TG_INFTY, p.xy_max_vel, p.xy_max_accel, TG_INFTY, p.xy_max_vel, p.xy_max_accel};
        for (int a = 0; a < 6; a++) { s.xmax[a] = b0[a]; s.xmin[a] = -b0[a]; }
        for (int a = 0; a < 2; a++) { s.umax[a] = p.xy_max_jerk; s.umin[a] = -1.0 * p.xy_max_jerk; }
        for (int k = 0; k < 4; k++) s.mixed_max[k] = p.xy_max_vel;
        s.mixed_max[4] = s.mixed_max[5] = p.xy_max_accel;
        for (int a = 0; a < 6; a++) s.x0_row[a] = a;
        s.ref_stride = 12; s.ref_off = 0; s.X_stride = 12; s.X_off = 0; s.U_stride = 4; s.U_off = 0;
    } else {
        const bool z = family == TG_FAM_Z;
        s.nx = 3; s.nu = 1; s.n_mixed = 0;
        std::memcpy(s.Ad, A3, sizeof(A3));
        s.Bd[2] = dt;
        s.qdiag[0] = z ? p.z_state_weight : p.yaw_state_weight;
        s.r = z ? p.z_input_weight : p.yaw_input_weight;
        s.smooth_w = z ? p.z_smooth_input_weight : p.yaw_smooth_input_weight;
        const double pmax = z ? TG_INFTY : 2.0 * M_PI;
        const double vmax = z ? p.z_max_vel : p.yaw_max_vel;
        const double amax = z ? p.z_max_accel : p.yaw_max_accel;
        const double jmax = z ? p.z_max_jerk : p.yaw_max_jerk;
        s.xmax[0] = pmax; s.xmax[1] = vmax; s.xmax[2] = amax;
        for (int a = 0; a < 3; a++) s.xmin[a] = -s.xmax[a];
        s.umax[0] = jmax; s.umin[0] = -1.0 * jmax;
        const int off = z ? 6 : 9;
        for (int a = 0; a < 3; a++) s.x0_row[a] = off + a;
        s.ref_stride = 12; s.ref_off = off; s.X_stride = 12; s.X_off = off;
        s.U_stride = 4; s.U_off = z ? 2 : 3;
        s.ref_pos_only = !z;   // yaw reference comes from computeYawRefTrajectory as [N+1]
        if (!z) { s.ref_stride = 1; s.ref_off = 0; }
    }
    return s;
}

} // namespace tg
