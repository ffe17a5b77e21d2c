// gfx950 kernels of the batched MPC engine.
//
//   admm_kernel      one workgroup = 16 instances of one QP stage, resident on chip for
//                    all ADMM iterations; the shared reduced-KKT inverse is applied with
//                    v_mfma_f64_16x16x4_f64 (instances are the 16 MFMA columns)
//   assemble_kernel  per-instance q, l, u of the reference's QP (castMPCToQPGradient /
//                    castMPCToQPConstraintBounds, mpc_6dof.cpp:371,450 and the 12-state twins)
//   xy_bounds_kernel MPC12STATE::computeXYBounds        (mpc_12state.cpp:724-791)
//   yaw_ref_kernel   MPC12STATE::computeYawRefTrajectory (mpc_12state.cpp:525-552)
//
// ADMM iteration (OSQP Algorithm 1 with scaling and rho adaptation off), per instance:
//   xt = Kinv (sigma x - q + A'(rho z - y));  zt = A xt
//   x  = alpha xt + (1-alpha) x
//   zr = alpha zt + (1-alpha) z;  z+ = clamp(zr + y/rho, l, u);  y+ = y + rho (zr - z+)
// The kernel keeps yr = y/rho.  The rows of A (castMPCToQPConstraintMatrix,
// mpc_6dof.cpp:409, mpc_12state.cpp:571/641/683) are never stored: a thread owns whole
// "cells" (one horizon step of one axis: D integrator states + the input) and applies the
// dynamics / box / mixed rows of its cells as fixed stencils on neighbouring cells.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <type_traits>

#include "device_types.h"

#ifndef TG_ABLATE
#define TG_ABLATE 0   // development, bit mask: 1 = no MFMA, 2 = no rhs phase, 4 = no projection phase
#endif
#ifndef TG_VARIANTS
#define TG_VARIANTS 0xFFFFFFFFu   // development: which kernel variants launch_admm compiles (see TG_HAS there)
#endif
#define TG_HAS(b) ((TG_VARIANTS >> (b)) & 1u)
// development: -DTG_STAMP accumulates the shader-clock time of every phase of an iteration per wave
// (workgroup 0) and leaves the sums in row 0 of U, [wave][8] (tools/phase_stamps.py)
#ifdef TG_STAMP
#define STAMP(i) do { __builtin_amdgcn_s_waitcnt(0); const long long now__ = (long long)__builtin_amdgcn_s_memtime(); \
                      if ((i) > 0) stamp_acc[(i) - 1] += now__ - stamp_last; stamp_last = now__; } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

namespace tg {

typedef double d4 __attribute__((ext_vector_type(4)));

// v_max_f64 / v_min_f64 issued as such: fmin / fmax first quieten every operand that comes from
// memory (one more instruction per bound, 12 per cell and iteration where the box rows are kept
// as s), which the hardware's IEEE maxNum / minNum does not need.
__device__ __forceinline__ double clampd(double v, double lo, double hi)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(v), "v"(lo));
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(r), "v"(hi));
    return r;
}

// max over the four lanes that share an instance inside one wave (lane bits 4 and 5)
__device__ __forceinline__ double wave_gmax(double v)
{
    v = fmax(v, __shfl_xor(v, 16));
    v = fmax(v, __shfl_xor(v, 32));
    return v;
}

__device__ __forceinline__ void lds_max(unsigned long long *p, double v)
{
    atomicMax(p, (unsigned long long)__double_as_longlong(v));
}

// Coefficient of state l of step i in dynamics row k of step i+1 (l >= k): the upper
// triangular Toeplitz block [1 dt dt^2/2; 0 1 dt; 0 0 1] (D = 3) or [1 dt; 0 1] (D = 2).
template <int D> __device__ __forceinline__ double chain_coef(int k, int l, double dt, double c2)
{
    const int d = l - k;
    return d == 0 ? 1.0 : (d == 1 ? dt : c2);
}

// A workgroup is NT threads (8 waves): thread (g = tid / 16, t = tid % 16) owns up to MAXC cells of
// instance slot t -- cells g, g + NT/16, ... of the stage, or, in the two-axis stage (MIXED), both
// axes' cells of steps g, g + NT/16, ...  The MFMA work is cut into items dealt to the waves by the
// host (StageDev::wi_*, wj_*), at most MAXIT per wave and stage of the product.
// PAIR:  two adjacent blocks may share an operator (6-DOF x / y): one fragment load, two accumulators.
// KSREG > 0: the operator fragments of the wave's single item stay in registers for the whole
//        solve (needs MAXIT == 1 and ks <= KSREG); KSREG == 0: they are streamed from L2.
// SEP:   rhs and xt live in separate LDS vectors, results are stored as soon as an item is done.
// LR:    the operators are applied as diagonal tiles + block-row low-rank rest (two stages, one
//        vector for rhs and xt; needs KSREG == 0 and !SEP).
//
// LDS regions, all [rows][16 instances] of doubles, addressed by absolute double index:
//   V0  rhs of the linear solve            n_int + S rows (the last S rows take the writes of
//   VX  xt, and x during a check           n_int + S rows  a thread's unused cell slots)
//   VX2 KPART == 2: second partial xt      n_int + S rows
//   WEQ w = rho z - y of the dynamics rows ncell*D + D rows (their y itself during a check)
//   WMX the same for the XY mixed rows     6 (N+1) + 6 rows
//   SB  LR: C r of every block             blocks * sb_rows rows
//   ZERO rows that stay zero: what a neighbour read returns past either end of the horizon
//   BGT  l = u of the dynamics rows of step 0 (-x0); those of the later steps are zero
//   CTAB per axis and slot: step size, lower and upper bound; then the reduction scratch
// SM:    the box rows' z and y are kept as one number s = alpha zt + (1 - alpha) z + y/rho, the
//        argument of the last projection: z = clamp(s, l, u), y/rho = s - z bit for bit, so the pair is
//        recomputed where it is used and a cell needs a register less per slot.  Valid when every
//        box contains zero (z = y = 0 at a cold start is then s = 0) and the boxes do not change
//        between warm-started calls; the host checks both (not MIXED).
// QL:    q of the cells' position slots lives in LDS (region QV) instead of registers.
// SMJ:   step-major cells: a thread owns all NAX cells of its steps g, g + NT/16, ... (as the two-axis
//        stage always does), so that the axis of a cell slot is known at compile time and the
//        per-axis step sizes and bounds are scalar operands instead of LDS reads.
// UK:    k-steps of U_I per tile in the streamed low-rank form (P.lr_ukp; fragments past the
//        operator's own are zero).
// PDT:   depth of the fragment ring of the streamed variants (an item is a whole number of rounds).
// KPART: 2 = register-resident dense variant in which some row blocks are split in two items of half
//        the k range each (fewer row blocks than waves); the second halves' products go to VX2.
// UNI:   how the step sizes and box bounds of a cell reach the vector phases when the cell's axis is
//        not known at compile time: 0 = per-axis LDS table; 1 = the slots past the position are
//        scalars (all axes share them), the position slot through the table; 2 = the position slot
//        too, as a select between two scalars (only the last axis differs: the altitude floor).
//        The host checks the conditions at launch.
template <int NT, int D, int NAX, bool MIXED, int MAXC, int MAXIT, bool PAIR, int KSREG, bool SEP, bool LR,
          bool SM = false, bool QL = false, int UK = 2, bool SMJ = false, int PDT = kRingDepth, int KPART = 1, int UNI = 0>
__global__ __launch_bounds__(NT) void admm_kernel(const StageDev P, const BatchArgs a)
{
    constexpr int S = D + 1;
    constexpr int kThreads = NT, kWaves = NT / 64, kStride = NT / kTileDev;
    extern __shared__ double lds[];
    const int tid = threadIdx.x;
    const int t = tid & 15, g = tid >> 4;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // The 16 columns of the tile are slots: a slot starts on instance blockIdx.x * 16 + t and, when
    // the grid is smaller than the batch, takes the next unassigned instance from a.queue each
    // time its instance terminates.  Every instance runs its own iteration count and its own
    // schedule of checks, so its result does not depend on what shares the tile with it.
    // Slot bookkeeping that only the checks need lives in LDS (slot_inst, slot_start, and the
    // slot's ||q||_inf in red[6]) to keep the iteration loop's registers for the iterates.
    bool exhausted = a.queue == nullptr;    // no further instance to take
    bool done = false;
    // instances of this launch: the batch, or (later phases of the step-size schedule) as many
    // entries of a.list as the previous phase handed over; read where it is needed (start and
    // refills) rather than kept in a register across the iterations
    auto batch_size = [&]() __attribute__((always_inline)) -> int { return a.count ? __builtin_amdgcn_readfirstlane(*a.count) : a.B; };
    // A continuation launch is sized for the whole batch but usually takes a few instances or none:
    // a workgroup whose start tickets lie past the list and that could take nothing from the queue
    // leaves before any set-up.
    // (the first launch of a solve clears the hand-over counters of the solve after it; it may itself
    // be a launch over a list that turns out empty, so this comes first)
    if (a.zero_counts != nullptr && blockIdx.x == 0 && tid < 2) a.zero_counts[tid] = 0;
    if (a.count != nullptr) {
        const int nb0 = batch_size();
        if ((int)blockIdx.x * kTileDev >= nb0 && (a.queue == nullptr || a.queue_base >= nb0)) return;
    }
    const int N = P.N, N1 = P.N + 1, ncell = NAX * N1;
    const int n_int = P.n_int, nb = P.nb, ks = P.ks, nrb = P.nrb;
    // Cell cc of thread group g: linear cell index ax * (N+1) + st, or ncell for an unused slot.
    // The two-axis stage (XY) keeps both cells of a step in one thread (cc even: axis 0, odd: axis
    // 1): the linear solve works on their sum and difference (see the rhs phase), and the step's
    // mixed rows need both.
    // axis of cell c: comparisons instead of a division by the run-time N + 1
    auto axis_of = [&](int c) __attribute__((always_inline)) -> int {
        int ax = 0;
#pragma unroll
        for (int k = 1; k < NAX; k++) ax += c >= k * N1 ? 1 : 0;
        return ax;
    };
    constexpr bool CAX = SMJ || MIXED;   // the axis of cell slot cc is cc % NAX
    auto cell_of = [&](int cc) __attribute__((always_inline)) -> int {
        if (CAX) { const int st = g + kStride * (cc / NAX); return st <= N ? (cc % NAX) * N1 + st : ncell; }
        const int c = g + kStride * cc;
        return c < ncell ? c : ncell;
    };
    // The same for the code that runs once per check (residuals, exit, refill): gq equals g but is
    // made opaque at every check, so that what this code derives from it (cell numbers, steps,
    // offsets into the batch arrays) is computed where it is used instead of being kept across the
    // iterations, where there is no register for it and it would live in scratch memory.
    int gq = g;
    auto cell_q = [&](int cc) __attribute__((always_inline)) -> int {
        if (CAX) { const int st = gq + kStride * (cc / NAX); return st <= N ? (cc % NAX) * N1 + st : ncell; }
        const int c = gq + kStride * cc;
        return c < ncell ? c : ncell;
    };

    // register-resident operator fragments: requested first, so that their L2 latency passes
    // behind the set-up below
    const int nit = P.wi_cnt[wave];
    double mreg[KSREG > 0 ? KSREG : 1];
    // the single MFMA item of a wave whose fragments are register resident, looked up once
    const int grp_r = P.wi_grp[wave][0], rb_r = P.wi_rb[wave][0];
    const bool two_r = PAIR && P.grp_cnt[grp_r] == 2;
    // KPART == 2 (register-resident dense fragments, fewer row blocks than waves): some row blocks
    // are split in two items, each half of the k range, so that every wave has work; the second
    // half's product goes to a vector of its own (VX2) and the projection adds the two.
    static_assert(KPART == 1 || (KSREG > 0 && !LR && SEP && MAXIT == 1), "k-range parts: register-resident dense variants only");
    const int k0_r = KPART == 2 ? P.wi_k0[wave][0] : 0;          // first k-step of the wave's item
    const int kl_r = KPART == 2 ? P.wi_nr[wave][0] : KSREG;      // its k-steps
    const int part_r = KPART == 2 ? P.wi_part[wave][0] : 0;
    const int boff_r = (P.grp_first[grp_r] * nb + 4 * k0_r + (lane >> 4)) * 16 + (lane & 15);
    // LR with register-resident fragments (short horizons): C r is formed as four partial sums,
    // one quarter of the k range per first-stage item, so that eight items fill the eight waves;
    // the second stage adds the partials and applies U_I: xt_I = D_I r_I + U_I sum_p s_I^(p).  A wave
    // holds the KQ fragments of its (group, part) and the 5 of its (group, tile).
    constexpr bool LRK = LR && KSREG > 0;
    constexpr int KQ = LRK ? KSREG / 4 : 1;
    double creg[KQ], dreg[LRK ? 5 : 1];
    const int grp_2 = P.wj_grp[wave][0], tile_2 = P.wj_rb[wave][0];
    if constexpr (LRK) {
        const bool has1 = nit > 0, has2 = P.wj_cnt[wave] > 0;
#pragma unroll
        for (int kk = 0; kk < KQ; kk++) {
            const int k = rb_r * KQ + kk;              // rb_r: the part of the k range
            creg[kk] = (has1 && k < ks) ? P.Mfrag[((size_t)P.grp_op[grp_r] * P.ks_alloc + k) * 64 + lane] : 0.0;
        }
#pragma unroll
        for (int kk = 0; kk < 5; kk++) dreg[kk] = has2 ? P.DUfrag[(((size_t)P.grp_op[grp_2] * nrb + tile_2) * 5 + kk) * 64 + lane] : 0.0;
    } else if constexpr (KSREG > 0) {
        const int grp = grp_r, rb = rb_r;
#pragma unroll
        for (int kk = 0; kk < KSREG; kk++) {
            const bool in = nit > 0 && kk < kl_r && k0_r + kk < ks;
            mreg[kk] = in ? P.Mfrag[(((size_t)P.grp_op[grp] * nrb + rb) * P.ks_alloc + k0_r + kk) * 64 + lane] : 0.0;
        }
    }

    const int V0 = 0;
    const int VX = SEP ? (n_int + S) * 16 : 0;
    const int X2 = KPART == 2 ? (n_int + S) * 16 : 0;                        // VX2 = VX + X2: products of the second k-range parts
    const int WEQ = (SEP ? 2 + (KPART == 2 ? 1 : 0) : 1) * (n_int + S) * 16;
    const int WMX = WEQ + (ncell * D + D) * 16;
    const int SB = WMX + (MIXED ? (N1 * 6 + 6) * 16 : 0);                   // LR: C r of every block, [blocks][sb_rows][16]
    const int QV = SB + (LR ? P.lr_np * (n_int / nb) * P.sb_rows * 16 : 0);   // QL: q of every cell, [ncell + 1][16]
    const int ZERO = QV + (QL ? (ncell + 1) * 16 : 0);
    const int BGT = ZERO + 8 * 16;                                           // [NAX][D][16] -x0: l = u of the step-0 rows
    const int CTAB = BGT + NAX * D * 16;                                     // [NAX][S][3] rho, lb, ub
    unsigned long long *red = (unsigned long long *)(lds + CTAB + NAX * S * 3 + 1); // [9][16] maxima
    double *rsum = (double *)(red + 9 * 16);                                        // [waves][16] partial sums
    int *slot_inst = (int *)(rsum + kWaves * 16);                                   // [16] instance of each slot
    int *slot_start = slot_inst + 16;                                               // [16] tile iteration it began at
    double *v0 = lds + V0, *vx = lds + VX;

    const double sigma = P.sigma, alpha = P.alpha, oma = 1.0 - P.alpha;
    const double dt = P.dt, c2 = P.c2, rho_eq = P.rho_eq;
    constexpr double s3 = 0.86602540378443864676, s2 = 0.70710678118654752440; // sqrt(3)/2, sqrt(2)/2

    for (int j = tid; j < CTAB; j += kThreads) lds[j] = 0.0;
    if (tid < NAX * S) {
        const int ax = tid / S, k = tid % S;
        lds[CTAB + tid * 3 + 0] = P.rho_box[ax][k];
        lds[CTAB + tid * 3 + 1] = P.lb[ax][k];
        lds[CTAB + tid * 3 + 2] = P.ub[ax][k];
    }
    __syncthreads();

    // ------------------------------------------------------------ per-thread cell state
    static_assert(!SM || !MIXED, "the two-axis stage's boxes change from call to call");
    // variables and their identity box rows (SM: zb holds s, yb is not kept)
    double x[MAXC][S], zb[MAXC][S], yb[SM ? 1 : MAXC][SM ? 1 : S];
    // dynamics rows of the cell: only y is kept.  Their z equals the bound b (l = u) after every
    // projection, and b is -x0 at step 0 and zero elsewhere, so both are read through i_bg.
    double yg[MAXC][D];
    double q[QL ? 1 : MAXC];                       // q of the position slot
    double hiv[MIXED ? MAXC : 1][2];               // XY: |v|, |a| box of the cell (per instance)
    // XY mixed rows (mpc_12state.cpp:600-630): the 6 rows of step m = g, g + NT/16, ... belong to
    // thread group g as a unit of their own, whichever thread owns the two cells of that step.
    // Their bounds are +-vb on the 4 velocity rows and +-am on the 2 acceleration rows
    // (computeXYBounds, mpc_12state.cpp:758-779).
    constexpr int MAXM = MIXED ? (MAXC + 1) / 2 : 1;
    double zm[MAXM][6], ym[MAXM][6], mvb[MAXM], mam[MAXM];
    int i_mw[MAXM];      // absolute index of the step's mixed-row w in WMX (spare rows when inactive)
    int i_mv[MAXM][2];   // index of the step's cell of axis 0 / 1, relative to VX
    bool mact[MAXM];
    int i_var[MAXC];     // index of the cell's first variable, relative to V0 / VX
    int i_prev[MAXC];    // absolute index of the previous cell's variables in VX (or ZERO)
    int i_prev2[KPART == 2 ? MAXC : 1];   // the same in VX2
    int i_wnext[MAXC];   // absolute index of the next cell's dynamics-row w (or ZERO)
    int i_ct[MAXC];      // absolute index of the cell's axis in the constant table
    bool lastax[MAXC];   // the cell is on the last axis
    int i_bg[MAXC];      // absolute index of the cell's dynamics-row bound (BGT at step 0, else ZERO)
    int i_mx[MIXED ? MAXC : 1];     // absolute index of the step's mixed-row w (or ZERO)
    bool axis0[MIXED ? MAXC : 1];

#pragma unroll
    for (int cc = 0; cc < MAXC; cc++) {
        const int c = cell_of(cc);
        const bool act = c < ncell;
        const int ax = act ? axis_of(c) : 0;
        const int st = act ? c - ax * N1 : 0;
        int abase = 0;
#pragma unroll
        for (int k = 0; k < NAX; k++) if (ax == k) abase = P.axis_base[k];
        i_var[cc] = act ? (abase + st * S) * 16 + t : n_int * 16 + t;
        i_prev[cc] = (act && st > 0) ? VX + i_var[cc] - S * 16 : ZERO + t;
        if constexpr (KPART == 2) i_prev2[cc] = (act && st > 0) ? i_prev[cc] + X2 : ZERO + t;
        i_wnext[cc] = (act && st < N) ? WEQ + ((c + 1) * D) * 16 + t : ZERO + t;
        i_ct[cc] = CTAB + ax * S * 3;
        lastax[cc] = ax == NAX - 1;
        i_bg[cc] = (act && st == 0) ? BGT + ax * D * 16 + t : ZERO + t;
        if (MIXED) {
            i_mx[cc] = (act && st >= 1) ? WMX + st * 6 * 16 + t : ZERO + t;
            axis0[cc] = ax == 0;
        }
    }
    if (MIXED) {
#pragma unroll
        for (int mm = 0; mm < MAXM; mm++) {
            const int m = g + kStride * mm;
            mact[mm] = m >= 1 && m <= N;
            const int ms = mact[mm] ? m : 0;
            i_mw[mm] = mact[mm] ? WMX + m * 6 * 16 + t : WMX + N1 * 6 * 16 + t;
            i_mv[mm][0] = (P.axis_base[0] + ms * S) * 16 + t;
            i_mv[mm][1] = (P.axis_base[1] + ms * S) * 16 + t;
        }
    }

    // per-axis constants of cell slot cc: scalar kernel arguments when the slot's axis is known at
    // compile time, else the LDS table
    // UNI >= 1: every axis has the same step size and box on the slots past the position (the host
    // checks), so those constants are scalars too and only the position slot goes through the table.
    // UNI == 2: on the position slot the axes differ only in the last one's step size and lower
    // bound (the altitude floor): a select between two scalars on whether the cell is on that axis.
    auto crho = [&](int cc, int k) __attribute__((always_inline)) -> double {
        return CAX ? P.rho_box[cc % NAX][k] : (UNI >= 1 && k >= 1) ? P.rho_box[0][k]
             : UNI == 2 ? (lastax[cc] ? P.rho_box[NAX - 1][0] : P.rho_box[0][0]) : lds[i_ct[cc] + k * 3]; };
    auto clo = [&](int cc, int k) __attribute__((always_inline)) -> double {
        return CAX ? P.lb[cc % NAX][k] : (UNI >= 1 && k >= 1) ? P.lb[0][k]
             : UNI == 2 ? (lastax[cc] ? P.lb[NAX - 1][0] : P.lb[0][0]) : lds[i_ct[cc] + k * 3 + 1]; };
    auto chi = [&](int cc, int k) __attribute__((always_inline)) -> double {
        return CAX ? P.ub[cc % NAX][k] : (UNI >= 1 && k >= 1) ? P.ub[0][k] : UNI == 2 ? P.ub[0][0] : lds[i_ct[cc] + k * 3 + 2]; };
    // the termination check reads the table (it runs once per 25 iterations and has no registers to spare)
    auto crho_t = [&](int cc, int k) __attribute__((always_inline)) -> double { return CAX ? P.rho_box[cc % NAX][k] : lds[i_ct[cc] + k * 3]; };
    // Step sizes and box bounds of the thread's cells, in registers when few cells per thread leave
    // room: the LDS pipe is the busiest unit of the rhs and projection phases.  Slots 1.. have
    // symmetric bounds in every stage (the host checks); slot 0 keeps both ends (altitude floor).
    constexpr bool REGC = MAXC <= 2 && !MIXED;
    double c_rho[REGC ? MAXC : 1][S], c_hi[REGC ? MAXC : 1][S], c_lo0[REGC ? MAXC : 1];
    double c_bg[REGC ? MAXC : 1][D];   // the cell's dynamics-row bound (-x0 at step 0, else zero), set by load()
    if constexpr (REGC) {
#pragma unroll
        for (int cc = 0; cc < MAXC; cc++) {
#pragma unroll
            for (int k = 0; k < S; k++) { c_rho[cc][k] = crho(cc, k); c_hi[cc][k] = chi(cc, k); }
            c_lo0[cc] = clo(cc, 0);
        }
    }

    // box of slot k of the thread's cell cc
    auto box_of = [&](int cc, int k, double &lo, double &hi) __attribute__((always_inline)) {
        if (MIXED && (k == 1 || k == 2)) { hi = hiv[cc][k - 1]; lo = -hi; }
        else if (REGC) { hi = c_hi[cc][k]; lo = k == 0 ? c_lo0[cc] : -hi; }
        else { lo = clo(cc, k); hi = chi(cc, k); }
    };
    // SM: z and y/rho of a box row from s
    auto zy_of = [&](int cc, int k, double lo, double hi, double &zv, double &yv) __attribute__((always_inline)) {
        if constexpr (SM) { const double sv = zb[cc][k]; zv = clampd(sv, lo, hi); yv = sv - zv; }
        else { zv = zb[cc][k]; yv = yb[cc][k]; }
    };

    // Start the slot's instance `inst` (fresh slots only; the others keep their state): iterates
    // zero or the previous solve's, q, the step-0 bound, the per-instance boxes, and the w of the
    // first rhs.  Returns the slot's part of ||q||_inf.
    auto load = [&](const bool fresh, const int inst) __attribute__((always_inline)) -> double {
        // an instance whose upstream stage did not reach "solved" is not attempted
        // (MPC12STATE::mpcLoop returns at the first failing stage, mpc_12state.cpp:1374-1453)
        const bool gated = fresh && inst < a.B && a.gate && a.gate[inst] != TG_DEV_SOLVED;
        const bool valid = fresh && inst < a.B && !gated;
        double qmax = 0.0;
        if (!fresh) return qmax;
        // the reference entries of all the thread's cells are requested before the first is used
        // (a slot's instance sits anywhere in the batch: every one of these reads is a cache miss)
        double refv[MAXC];
#pragma unroll
        for (int cc = 0; cc < MAXC; cc++) {
            const int c = cell_q(cc);
            const bool act = c < ncell;
            const int ax = act ? axis_of(c) : 0;
            const int st = act ? c - ax * N1 : 0;
            int refoff = 0;
#pragma unroll
            for (int k = 0; k < NAX; k++) if (ax == k) refoff = P.ref_off[k];
            refv[cc] = a.ref[(act && valid) ? (size_t)(st * P.ref_stride + refoff) * a.ref_ld + inst : (size_t)0];
        }
#pragma unroll
        for (int cc = 0; cc < MAXC; cc++) {
            const int c = cell_q(cc);
            const bool act = c < ncell;
            const int ax = act ? axis_of(c) : 0;
            const int st = act ? c - ax * N1 : 0;
            int abase = 0, x0r = 0, refoff = 0;
            double qw = 0.0;
#pragma unroll
            for (int k = 0; k < NAX; k++)
                if (ax == k) { abase = P.axis_base[k]; x0r = P.x0_row[k]; refoff = P.ref_off[k]; qw = P.qw[k]; }
            double zg0[D], b0[D];
            double qc = 0.0;
#pragma unroll
            for (int k = 0; k < S; k++) { x[cc][k] = 0.0; zb[cc][k] = 0.0; if constexpr (!SM) yb[cc][k] = 0.0; }
#pragma unroll
            for (int k = 0; k < D; k++) { zg0[k] = 0.0; yg[cc][k] = 0.0; b0[k] = 0.0; }
            if (MIXED) {
                hiv[cc][0] = chi(cc, 1);
                hiv[cc][1] = chi(cc, 2);
            }
            if (act && valid) {
                qc = qw * refv[cc];
                qmax = fmax(qmax, fabs(qc));
                if (st == 0) {
#pragma unroll
                    for (int k = 0; k < D; k++) {
                        b0[k] = -a.x0[(size_t)(x0r + k * P.x0_slot) * a.ld + inst];
                        lds[i_bg[cc] + k * 16] = b0[k];
                    }
                }
                if (MIXED && st >= 1) {
                    const size_t r0 = (size_t)(6 * (st - 1) + 3 * ax) * a.ov_ld + inst;
                    hiv[cc][0] = a.ov_box[r0 + (size_t)a.ov_ld];
                    hiv[cc][1] = a.ov_box[r0 + (size_t)2 * a.ov_ld];
                }
                if (a.warm_in) {
                    const size_t w0 = (size_t)(abase + st * S) * a.ldw + inst;
#pragma unroll
                    for (int k = 0; k < S; k++) {
                        // the rows hold x, z and y itself; the kernel keeps y / rho under its own operator
                        x[cc][k] = a.warm[w0 + (size_t)k * a.ldw];
                        const double zv = a.warm[w0 + (size_t)(n_int + k) * a.ldw];
                        const double yv = a.warm[w0 + (size_t)(2 * n_int + k) * a.ldw] / crho_t(cc, k);
                        if constexpr (SM) zb[cc][k] = zv + yv;
                        else { zb[cc][k] = zv; yb[cc][k] = yv; }
                    }
                    const size_t e0 = (size_t)(3 * n_int + c * D) * a.ldw + inst;
#pragma unroll
                    for (int k = 0; k < D; k++) {
                        zg0[k] = a.warm[e0 + (size_t)k * a.ldw];
                        yg[cc][k] = a.warm[e0 + (size_t)(ncell * D + k) * a.ldw];
                    }
                }
            }
            // w of the cell's rows for the first rhs (unused cells write the spare rows at the end)
            const int cw = act ? c : ncell;
            if constexpr (QL) lds[QV + cw * 16 + t] = qc; else q[cc] = qc;
#pragma unroll
            for (int k = 0; k < D; k++) {
                lds[WEQ + (cw * D + k) * 16 + t] = rho_eq * (zg0[k] - yg[cc][k]);
                // the first projection relaxes with the starting z (zero, or the previous solve's):
                // alpha zt + (1-alpha) z0 + y = alpha zt + (1-alpha) b + [y + (1-alpha)(z0 - b)]
                yg[cc][k] += oma * (zg0[k] - b0[k]);
                if constexpr (REGC) c_bg[cc][k] = b0[k];
            }
        }
        if (MIXED) {
#pragma unroll
            for (int mm = 0; mm < MAXM; mm++) {
                const int m = g + kStride * mm;
                mvb[mm] = 0.0; mam[mm] = 0.0;
#pragma unroll
                for (int k = 0; k < 6; k++) { zm[mm][k] = 0.0; ym[mm][k] = 0.0; }
                if (mact[mm] && valid) {
                    mvb[mm] = a.ov_mixed[(size_t)(6 * (m - 1) + 0) * a.ov_ld + inst];
                    mam[mm] = a.ov_mixed[(size_t)(6 * (m - 1) + 4) * a.ov_ld + inst];
                    if (a.warm_in) {
                        const size_t m0 = (size_t)(3 * n_int + 2 * ncell * D + m * 6) * a.ldw + inst;
#pragma unroll
                        for (int k = 0; k < 6; k++) {
                            zm[mm][k] = a.warm[m0 + (size_t)k * a.ldw];
                            ym[mm][k] = a.warm[m0 + (size_t)(N1 * 6 + k) * a.ldw] / P.rho_mx[k];
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < 6; k++) lds[i_mw[mm] + k * 16] = P.rho_mx[k] * (zm[mm][k] - ym[mm][k]);
            }
        }
        done = !valid;
        if (gated && g == 0) {
            if (a.status) a.status[inst] = TG_DEV_UNSOLVED;
            if (a.iters) a.iters[inst] = 0;
        }
        return qmax;
    };
    {
        const int tk0 = blockIdx.x * kTileDev + t;
        const int nb_ = batch_size();
        // nothing beyond the slots' first instances (a later operator's launch finds few, or none):
        // no refill to wait for -- a workgroup without instances leaves at once
        if (a.queue != nullptr && a.queue_base >= nb_) exhausted = true;
        const int inst0 = tk0 < nb_ ? (a.list ? a.list[tk0] : tk0) : a.B;   // a.B: no instance
        double qf = load(true, inst0);
        if (g == 0) { red[6 * 16 + t] = 0ull; slot_inst[t] = inst0; slot_start[t] = 0; }
        __syncthreads();
        qf = wave_gmax(qf);
        if ((lane >> 4) == 0) lds_max(&red[6 * 16 + t], qf);
    }

    // ---- streamed operators (KSREG == 0): a wave's fragments lie in P.Wfrag in the order it uses
    // them -- first-stage items (each a whole number of ring rounds), then, in the low-rank form,
    // the D_I / U_I fragments of its tiles -- so one ring of PD loads in flight serves the whole
    // MFMA phase.  The items' LDS offsets are wave-uniform and looked up once, here.
    constexpr bool STREAM = KSREG == 0;
    constexpr int PD = PDT;
    constexpr int KD = 4 + UK;                      // fragments of a second-stage item
    const int laneoff = (lane >> 4) * 16 + (lane & 15);
    // wave-uniform base of the stream; a load is base + lane, so that the base stays in scalar registers
    const double *ws = STREAM ? P.Wfrag + (size_t)P.wave_off[wave] * 64 : nullptr;
    const unsigned ulane = (unsigned)lane;
    // Keeps a wave-uniform LDS offset in a scalar register and its per-lane address out of the
    // loop-invariant code the compiler would hoist (and then spill: one address per item).
    auto uniform_off = [](int v) __attribute__((always_inline)) -> int {
        v = __builtin_amdgcn_readfirstlane(v);
        asm volatile("" : "+s"(v));
        return v;
    };
    const int pstride = LR ? (n_int / nb) * P.sb_rows * 16 : 0;   // doubles between the partial sums of C r
    const int njt = (STREAM && LR) ? P.wj_cnt[wave] : 0;
    int s1_b[MAXIT], s1_o[MAXIT], s1_nr[MAXIT], s1_rows[MAXIT], s2_r[MAXIT], s2_s[MAXIT], s2_rows[MAXIT];
    // a wave's items of a stage are all pairs of blocks sharing an operator, or all single blocks
    const bool s1_two_w = STREAM && PAIR && nit > 0 && P.grp_cnt[P.wi_grp[wave][0]] == 2;
    const bool s2_two_w = STREAM && PAIR && njt > 0 && P.grp_cnt[P.wj_grp[wave][0]] == 2;
    if constexpr (STREAM) {
#pragma unroll
        for (int it = 0; it < MAXIT; it++) {
            const bool in = it < nit;
            const int grp = in ? P.wi_grp[wave][it] : 0, rb = in ? P.wi_rb[wave][it] : 0;
            const int k0 = in ? P.wi_k0[wave][it] : 0, part = in ? P.wi_part[wave][it] : 0;
            const int blk = P.grp_first[grp];
            s1_nr[it] = in ? P.wi_nr[wave][it] : 0;
            s1_b[it] = V0 + (blk * nb + 4 * k0) * 16;
            s1_o[it] = LR ? SB + part * pstride + (blk * P.sb_rows + rb * 16) * 16 : VX + (blk * nb + rb * 16) * 16;
            s1_rows[it] = nb - rb * 16;
        }
#pragma unroll
        for (int jt = 0; jt < MAXIT; jt++) {
            const bool in = jt < njt;
            const int grp = in ? P.wj_grp[wave][jt] : 0, I = in ? P.wj_rb[wave][jt] : 0;
            const int blk = P.grp_first[grp];
            s2_r[jt] = V0 + (blk * nb + I * 16) * 16;
            s2_s[jt] = SB + (blk * P.sb_rows + I * P.lr_q) * 16;
            s2_rows[jt] = nb - I * 16;
        }
    }
    double af[STREAM ? PD : 1];

    const int max_iter = P.max_iter, check = P.check, fine = P.check_fine;
    const double eps_pinf = P.eps_prim_inf;
    if (__syncthreads_and((done && exhausted) ? 1 : 0)) return;   // nothing to solve in this tile

    // One ADMM iteration.  CHK marks an iteration that is followed by a termination check: it
    // also needs the step delta_y = y+ - y of every row for OSQP's primal infeasibility test:
    // ||delta_y||_inf, u'(dy)+ + l'(dy)- and ||A' delta_y||_inf are accumulated as the rows are
    // projected (pin_n, pin_l, m_atdy).  A' delta_y of a cell's variables takes delta_y of the
    // next step's dynamics rows, which belong to another thread: the cell's owner forms them
    // itself, with the operations their owner uses, from the next cell's xt and its own, so that
    // nothing has to be kept or exchanged until the check.
    double pin_n = 0.0, pin_l = 0.0, m_atdy = 0.0;
#ifdef TG_STAMP
    long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last = 0;
    long long chk_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, chk_last = 0;   // sections of a termination check; [7] counts them
#define CSTAMP(i) do { __builtin_amdgcn_s_waitcnt(0); const long long now__ = (long long)__builtin_amdgcn_s_memtime(); \
                       if ((i) > 0) chk_acc[(i) - 1] += now__ - chk_last; chk_last = now__; } while (0)
#else
#define CSTAMP(i) do { } while (0)
#endif
    auto step = [&](auto chk_c) __attribute__((always_inline)) {
        // The iterates must not depend on whether this is a checking iteration (a slot meets them
        // at different points of its run depending on its neighbours): every rounding is written
        // out, and the compiler may not fuse differently in the two instantiations.
#pragma clang fp contract(off)
        constexpr bool CHK = decltype(chk_c)::value;
#ifdef TG_STAMP
        { __builtin_amdgcn_s_waitcnt(0); const long long now__ = (long long)__builtin_amdgcn_s_memtime();
          if (stamp_last) stamp_acc[7] += now__ - stamp_last; stamp_last = now__; }
#else
        STAMP(0);
#endif
        // -------------------------------------------------------- rhs = sigma x - q + A'(rho z - y)
        // Cells are handled in groups of G: every LDS read of the group is issued before its first
        // LDS write.  The compiler cannot tell that the regions do not overlap, so a write between
        // two cells' reads would serialise their latencies.  The dense variants with many cells
        // per thread have no registers to spare for it; the low-rank ones take pairs.
        constexpr int G = MIXED ? 2 : (MAXC <= 2 ? MAXC : (LR ? 2 : 1));
#pragma unroll
        for (int c0 = 0; c0 < ((TG_ABLATE & 2) ? 0 : MAXC); c0 += G) {
            double wn[G][D], wo[G][D], rr[G][S], mixc[G][3], bx[SM ? G : 1][S][2], qv[G];
#pragma unroll
            for (int j = 0; j < G; j++) {
                const int cc = c0 + j;
                if (cc >= MAXC) continue;
                const int cw = cell_of(cc);
#pragma unroll
                for (int k = 0; k < D; k++) {
                    wo[j][k] = lds[WEQ + (cw * D + k) * 16 + t];
                    wn[j][k] = lds[i_wnext[cc] + k * 16];
                }
                if constexpr (QL) qv[j] = lds[QV + cw * 16 + t]; else qv[j] = q[cc];
                if constexpr (G > 1 || SM) {
#pragma unroll
                    for (int k = 0; k < S; k++) {
                        rr[j][k] = REGC ? c_rho[cc][k] : crho(cc, k);
                        if constexpr (SM) box_of(cc, k, bx[j][k][0], bx[j][k][1]);
                    }
                }
                mixc[j][0] = mixc[j][1] = mixc[j][2] = 0.0;
                if (MIXED) {
                    const double *wm = lds + i_mx[cc];
                    const double w0 = wm[0], w1 = wm[16], w2 = wm[32], w3 = wm[48], w4 = wm[64], w5 = wm[80];
                    // axis 0: s3 (w0 + w1) + (w2 - w3) / 2, axis 1: (w0 - w1) / 2 + s3 (w2 + w3); selecting
                    // the sums keeps the coefficients compile-time constants instead of per-lane registers
                    const bool a0 = axis0[cc];
                    mixc[j][1] = __builtin_fma(s3, a0 ? w0 + w1 : w2 + w3, 0.5 * (a0 ? w2 - w3 : w0 - w1));
                    mixc[j][2] = s2 * (a0 ? w4 + w5 : w4 - w5);
                }
            }
            double racc[G][S];
#pragma unroll
            for (int j = 0; j < G; j++) {
                const int cc = c0 + j;
                if (cc >= MAXC) continue;
#pragma unroll
                for (int l = 0; l < D; l++) {
                    double zv, yv;
                    zy_of(cc, l, SM ? bx[j][l][0] : 0.0, SM ? bx[j][l][1] : 0.0, zv, yv);
                    double acc = __builtin_fma((G > 1 || SM) ? rr[j][l] : crho(cc, l), zv - yv, sigma * x[cc][l]) - wo[j][l];
                    if (l == 0) acc -= qv[j];
#pragma unroll
                    for (int k = 0; k <= l; k++) acc = __builtin_fma(chain_coef<D>(k, l, dt, c2), wn[j][k], acc);
                    if (MIXED && l >= 1) acc += mixc[j][l];
                    racc[j][l] = acc;
                }
                // input slot; at step N it is a padding variable whose x, z, y and neighbours stay zero
                double zvD, yvD;
                zy_of(cc, D, SM ? bx[j][D][0] : 0.0, SM ? bx[j][D][1] : 0.0, zvD, yvD);
                racc[j][D] = __builtin_fma((G > 1 || SM) ? rr[j][D] : crho(cc, D), zvD - yvD, __builtin_fma(dt, wn[j][D - 1], sigma * x[cc][D]));
            }
            if constexpr (MIXED) {
                // The two axes are interchangeable in K = P + sigma I + A' rho A (same weights, same
                // step sizes, and the mixed rows map onto themselves when x and y swap), so
                // K^-1 = [[E, F], [F, E]] and  xt_x + xt_y = (E + F)(r_x + r_y),
                // xt_x - xt_y = (E - F)(r_x - r_y): two products of half the size.  The axis-0 rows
                // carry the sum, the axis-1 rows the difference; the operators hold (E +- F) / 2.
                double *o0 = v0 + i_var[c0], *o1 = v0 + i_var[c0 + 1];
#pragma unroll
                for (int l = 0; l < S; l++) { o0[l * 16] = racc[0][l] + racc[1][l]; o1[l * 16] = racc[0][l] - racc[1][l]; }
            } else {
#pragma unroll
                for (int j = 0; j < G; j++) {
                    const int cc = c0 + j;
                    if (cc >= MAXC) continue;
                    double *out = v0 + i_var[cc];
#pragma unroll
                    for (int l = 0; l < S; l++) out[l * 16] = racc[j][l];
                }
            }
        }
        STAMP(1);
        if constexpr (STREAM) {   // the ring's first fragments travel while the waves gather at the barrier
#pragma unroll
            for (int p = 0; p < PD; p++) af[p] = ws[ulane + p * 64];
        }
        __syncthreads();
        STAMP(2);
        // -------------------------------------------------------- xt = Kinv * rhs  (MFMA)
        // D[16 rows x 16 instances] += A[16 x 4] * B[4 x 16]: A = operator fragment (registers
        // when it fits, else streamed from L2 through a prefetch ring), B = rhs rows from LDS
        if constexpr (LRK) {
            const int nblk = n_int / nb;
            const size_t pstride = (size_t)nblk * P.sb_rows * 16;      // doubles between partial sums
            {   // stage 1: this wave's quarter of C r
                const int blk = P.grp_first[grp_r];
                const double *bp0 = v0 + ((size_t)blk * nb + 4 * rb_r * KQ + (lane >> 4)) * 16 + (lane & 15);
                auto run = [&](auto two_c) __attribute__((always_inline)) {
                    constexpr bool TWO = decltype(two_c)::value;
                    double bf0[KQ], bf1[TWO ? KQ : 1];
#pragma unroll
                    for (int kk = 0; kk < KQ; kk++) {
                        bf0[kk] = bp0[kk * 64];
                        if constexpr (TWO) bf1[kk] = bp0[(size_t)nb * 16 + kk * 64];
                    }
                    d4 c0 = (d4){0, 0, 0, 0}, c1 = (d4){0, 0, 0, 0};
#pragma unroll
                    for (int kk = 0; kk < KQ; kk++) {
                        if (TWO || !(kk & 1)) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(creg[kk], bf0[kk], c0, 0, 0, 0);
                        else c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(creg[kk], bf0[kk], c1, 0, 0, 0);
                        if constexpr (TWO) c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(creg[kk], bf1[kk], c1, 0, 0, 0);
                    }
                    if constexpr (!TWO) {
#pragma unroll
                        for (int r = 0; r < 4; r++) c0[r] += c1[r];
                    }
                    double *o0 = lds + SB + rb_r * pstride + ((size_t)blk * P.sb_rows + (lane >> 4)) * 16 + (lane & 15);
#pragma unroll
                    for (int r = 0; r < 4; r++) o0[r * 64] = c0[r];
                    if constexpr (TWO) {
#pragma unroll
                        for (int r = 0; r < 4; r++) o0[(size_t)P.sb_rows * 16 + r * 64] = c1[r];
                    }
                };
                if (nit > 0) { if (PAIR && two_r) run(std::true_type{}); else run(std::false_type{}); }
            }
            // The tile's own product D_I r_I does not need C r: it is issued before the barrier,
            // where it fills the matrix pipe while other waves still wait for their operands, and
            // its accumulators are kept across the barrier.
            const int blk2 = P.grp_first[grp_2];
            const bool two2 = PAIR && P.grp_cnt[grp_2] == 2;
            const bool has2 = P.wj_cnt[wave] > 0;
            d4 h0 = (d4){0, 0, 0, 0}, h1 = (d4){0, 0, 0, 0};
            {
                const double *r0p = v0 + ((size_t)blk2 * nb + tile_2 * 16 + (lane >> 4)) * 16 + (lane & 15);
                auto runD = [&](auto two_c) __attribute__((always_inline)) {
                    constexpr bool TWO = decltype(two_c)::value;
                    double bf0[4], bf1[TWO ? 4 : 1];
#pragma unroll
                    for (int kk = 0; kk < 4; kk++) {
                        bf0[kk] = r0p[kk * 64];
                        if constexpr (TWO) bf1[kk] = r0p[(size_t)nb * 16 + kk * 64];
                    }
#pragma unroll
                    for (int kk = 0; kk < 4; kk++) {
                        if (TWO || !(kk & 1)) h0 = __builtin_amdgcn_mfma_f64_16x16x4f64(dreg[kk], bf0[kk], h0, 0, 0, 0);
                        else h1 = __builtin_amdgcn_mfma_f64_16x16x4f64(dreg[kk], bf0[kk], h1, 0, 0, 0);
                        if constexpr (TWO) h1 = __builtin_amdgcn_mfma_f64_16x16x4f64(dreg[kk], bf1[kk], h1, 0, 0, 0);
                    }
                };
                if (has2) { if (PAIR && two2) runD(std::true_type{}); else runD(std::false_type{}); }
            }
            STAMP(3);
            __syncthreads();
            STAMP(4);
            {   // stage 2: + U_I (s^(0) + ... + s^(3)), the four partial sums added on the way in
                const double *s0p = lds + SB + ((size_t)blk2 * P.sb_rows + tile_2 * P.lr_q + (lane >> 4)) * 16 + (lane & 15);
                auto run = [&](auto two_c) __attribute__((always_inline)) {
                    constexpr bool TWO = decltype(two_c)::value;
                    double sp0[4], sp1[TWO ? 4 : 1];
#pragma unroll
                    for (int kk = 0; kk < 4; kk++) {
                        sp0[kk] = s0p[kk * pstride];
                        if constexpr (TWO) sp1[kk] = s0p[(size_t)P.sb_rows * 16 + kk * pstride];
                    }
                    h0 = __builtin_amdgcn_mfma_f64_16x16x4f64(dreg[4], (sp0[0] + sp0[1]) + (sp0[2] + sp0[3]), h0, 0, 0, 0);
                    if constexpr (TWO) h1 = __builtin_amdgcn_mfma_f64_16x16x4f64(dreg[4], (sp1[0] + sp1[1]) + (sp1[2] + sp1[3]), h1, 0, 0, 0);
                    if constexpr (!TWO) {
#pragma unroll
                        for (int r = 0; r < 4; r++) h0[r] += h1[r];
                    }
                    double *o0 = vx + ((size_t)blk2 * nb + tile_2 * 16 + (lane >> 4)) * 16 + (lane & 15);
                    const int row0 = tile_2 * 16 + (lane >> 4);   // rows past the block are MFMA padding
#pragma unroll
                    for (int r = 0; r < 4; r++) if (row0 + 4 * r < nb) o0[r * 64] = h0[r];
                    if constexpr (TWO) {
#pragma unroll
                        for (int r = 0; r < 4; r++) if (row0 + 4 * r < nb) o0[(size_t)nb * 16 + r * 64] = h1[r];
                    }
                };
                if (has2) { if (PAIR && two2) run(std::true_type{}); else run(std::false_type{}); }
            }
        } else if constexpr (LR) {
            // Block-row low-rank form: stage 1, s = C r for every block as two partial sums (the
            // k range of a row block of the stacked C is cut in two items, so that the eight waves
            // share the work evenly); stage 2, per 16-row tile I, xt_I = D_I r_I + U_I s_I written
            // over r_I: only the tile's own wave reads or writes those rows after the barrier, so
            // rhs and xt share one vector without another barrier.  The wave's fragments arrive
            // through the ring in the order of use; the first PD were requested at the top of the
            // iteration, and the second stage's first ones are requested during the first stage.
            const double *wp = ws + PD * 64;
            const int nb16 = nb * 16, sb16 = P.sb_rows * 16;
            const bool np2 = P.lr_np == 2;
            // a wave's items are all pairs of blocks or all single blocks (the host deals them so)
            auto stage1 = [&](auto two_c) __attribute__((always_inline)) {
                constexpr bool TWO = decltype(two_c)::value;
#pragma unroll
                for (int it = 0; it < MAXIT; it++) {
                    if (it < nit) {
                        const double *b = lds + uniform_off(s1_b[it]) + laneoff;
                        // two independent accumulation chains either way: a dependent MFMA issues
                        // every 128 clocks, two alternating chains every 64
                        d4 c0 = (d4){0, 0, 0, 0}, c1 = (d4){0, 0, 0, 0};
                        for (int r = 0; r < s1_nr[it]; r++) {
#pragma unroll
                            for (int p = 0; p < PD; p++) {
                                const double bf0 = b[p * 64];
                                if (TWO || !(p & 1)) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(af[p], bf0, c0, 0, 0, 0);
                                else c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(af[p], bf0, c1, 0, 0, 0);
                                if constexpr (TWO) {
                                    const double bf1 = b[nb16 + p * 64];
                                    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(af[p], bf1, c1, 0, 0, 0);
                                }
                                af[p] = wp[ulane + p * 64];
                            }
                            wp += PD * 64; b += PD * 64;
                        }
                        if constexpr (!TWO) {
#pragma unroll
                            for (int r = 0; r < 4; r++) c0[r] += c1[r];
                        }
                        // rows past the stacked C are zero rows of the operator: stored like the others
                        double *o0 = lds + uniform_off(s1_o[it]) + laneoff;
#pragma unroll
                        for (int r = 0; r < 4; r++) o0[r * 64] = c0[r];
                        if constexpr (TWO) {
#pragma unroll
                            for (int r = 0; r < 4; r++) o0[sb16 + r * 64] = c1[r];
                        }
                    }
                }
            };
            if (PAIR && s1_two_w) stage1(std::true_type{}); else stage1(std::false_type{});
            STAMP(3);
            __syncthreads();
            STAMP(4);
            // wp now points at fragment PD of the second stage: fragment j + PD is wp[j * 64]
            auto stage2 = [&](auto two_c) __attribute__((always_inline)) {
                constexpr bool TWO = decltype(two_c)::value;
#pragma unroll
                for (int jt = 0; jt < MAXIT; jt++) {
                    if (jt < njt) {
                        double *r0p = lds + uniform_off(s2_r[jt]) + laneoff;
                        const double *s0p = lds + uniform_off(s2_s[jt]) + laneoff;
                        double bf0[KD], bf1[TWO ? KD : 1];
#pragma unroll
                        for (int kk = 0; kk < 4; kk++) {
                            bf0[kk] = r0p[kk * 64];
                            if constexpr (TWO) bf1[kk] = r0p[nb16 + kk * 64];
                        }
#pragma unroll
                        for (int u = 0; u < UK; u++) {      // the two partial sums are added on the way in
                            bf0[4 + u] = s0p[u * 64];
                            if constexpr (TWO) bf1[4 + u] = s0p[sb16 + u * 64];
                            if (np2) {
                                bf0[4 + u] += s0p[pstride + u * 64];
                                if constexpr (TWO) bf1[4 + u] += s0p[sb16 + pstride + u * 64];
                            }
                        }
                        d4 c0 = (d4){0, 0, 0, 0}, c1 = (d4){0, 0, 0, 0};
#pragma unroll
                        for (int kk = 0; kk < KD; kk++) {
                            const int j = jt * KD + kk;          // compile-time after unrolling
                            if (TWO || !(kk & 1)) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(af[j % PD], bf0[kk], c0, 0, 0, 0);
                            else c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(af[j % PD], bf0[kk], c1, 0, 0, 0);
                            if constexpr (TWO) c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(af[j % PD], bf1[kk], c1, 0, 0, 0);
                            if (j + PD < MAXIT * KD) af[j % PD] = wp[ulane + j * 64];
                        }
                        if constexpr (!TWO) {
#pragma unroll
                            for (int r = 0; r < 4; r++) c0[r] += c1[r];
                        }
                        if (s2_rows[jt] >= 16) {                      // a whole tile (all but the block's last)
#pragma unroll
                            for (int r = 0; r < 4; r++) r0p[r * 64] = c0[r];
                            if constexpr (TWO) {
#pragma unroll
                                for (int r = 0; r < 4; r++) r0p[nb16 + r * 64] = c1[r];
                            }
                        } else {
                            const int rows = s2_rows[jt] - (lane >> 4);   // rows past the block are MFMA padding
#pragma unroll
                            for (int r = 0; r < 4; r++) if (4 * r < rows) r0p[r * 64] = c0[r];
                            if constexpr (TWO) {
#pragma unroll
                                for (int r = 0; r < 4; r++) if (4 * r < rows) r0p[nb16 + r * 64] = c1[r];
                            }
                        }
                    }
                }
            };
            if (PAIR && s2_two_w) stage2(std::true_type{}); else stage2(std::false_type{});
        } else
        {
            // with a separate xt vector (SEP) an item's rows are stored as soon as they are done;
            // otherwise xt overwrites the rhs and the stores wait for every wave's last read
            auto store = [&](int ooff, int rows_left, bool two, const d4 &c0, const d4 &c1) __attribute__((always_inline)) {
                double *o0 = lds + uniform_off(ooff) + laneoff;
                double *o1 = o0 + (size_t)nb * 16;
                if (rows_left >= 16) {                        // a whole row block (all but the block's last)
#pragma unroll
                    for (int r = 0; r < 4; r++) o0[r * 64] = c0[r];
                    if (PAIR && two) {
#pragma unroll
                        for (int r = 0; r < 4; r++) o1[r * 64] = c1[r];
                    }
                } else {
                    const int rows = rows_left - (lane >> 4);   // rows past the block are MFMA padding
#pragma unroll
                    for (int r = 0; r < 4; r++) if (4 * r < rows) o0[r * 64] = c0[r];
                    if (PAIR && two) {
#pragma unroll
                        for (int r = 0; r < 4; r++) if (4 * r < rows) o1[r * 64] = c1[r];
                    }
                }
            };
            d4 acc0[SEP ? 1 : MAXIT], acc1[(PAIR && !SEP) ? MAXIT : 1];
            if constexpr (KSREG > 0) {
                // ks <= KSREG (host dispatch): all B operands are fetched before the first
                // MFMA and the chain has no branches; fragments past ks are zero and their B
                // rows are whatever finite values follow the block in LDS
                if (nit > 0) {
                    const double *bp0 = v0 + boff_r;
                    const double *bp1 = bp0 + (size_t)nb * 16;
                    auto run = [&](auto two_c, auto cnt_c) __attribute__((always_inline)) {
                        constexpr bool TWO = decltype(two_c)::value;
                        constexpr int CNT = decltype(cnt_c)::value;      // k-steps of the item (KSREG, or half of it)
                        double bf0[CNT], bf1[TWO ? CNT : 1];
#pragma unroll
                        for (int kk = 0; kk < CNT; kk++) {
                            bf0[kk] = bp0[kk * 64];
                            if constexpr (TWO) bf1[kk] = bp1[kk * 64];
                        }
                        __builtin_amdgcn_sched_barrier(0);   // keep the loads ahead of the MFMA chain
                        // A dependent MFMA issues every 128 clocks, an independent one every 64:
                        // a single product is split over two accumulators (even / odd k steps),
                        // in case the wave has its SIMD to itself during this phase.
                        d4 c0 = (d4){0, 0, 0, 0}, c1 = (d4){0, 0, 0, 0};
#pragma unroll
                        for (int kk = 0; kk < CNT; kk++) {
#if !(TG_ABLATE & 1)
                            if (TWO || !(kk & 1)) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(mreg[kk], bf0[kk], c0, 0, 0, 0);
                            else c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(mreg[kk], bf0[kk], c1, 0, 0, 0);
                            if constexpr (TWO) c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(mreg[kk], bf1[kk], c1, 0, 0, 0);
#else
                            c0[0] += mreg[kk] * bf0[kk];
                            if constexpr (TWO) c1[0] += mreg[kk] * bf1[kk];
#endif
                        }
                        if constexpr (!TWO) {
#pragma unroll
                            for (int r = 0; r < 4; r++) c0[r] += c1[r];
                        }
                        if constexpr (SEP) store(VX + part_r * X2 + P.grp_first[grp_r] * nb * 16 + rb_r * 256, nb - rb_r * 16, TWO, c0, c1);
                        else { acc0[0] = c0; if constexpr (PAIR) acc1[0] = c1; }
                    };
                    constexpr int KH = (KSREG + 1) / 2;
                    if (KPART == 2 && kl_r <= KH) run(std::false_type{}, std::integral_constant<int, KH>{});
                    else if (PAIR && two_r) run(std::true_type{}, std::integral_constant<int, KSREG>{});
                    else run(std::false_type{}, std::integral_constant<int, KSREG>{});
                }
            } else {
                // the wave's items follow one another in its fragment stream (each a whole number
                // of ring rounds, padded with zero fragments): the ring never drains between items,
                // and its first PD fragments were requested at the top of the iteration
                const double *wp = ws + PD * 64;
                const int nb16 = nb * 16;
                auto stage1 = [&](auto two_c) __attribute__((always_inline)) {
                    constexpr bool TWO = decltype(two_c)::value;
#pragma unroll
                    for (int it = 0; it < MAXIT; it++) {
                        if (it < nit) {
                            const double *b = lds + uniform_off(s1_b[it]) + laneoff;
                            d4 c0 = (d4){0, 0, 0, 0}, c1 = (d4){0, 0, 0, 0};
                            for (int r = 0; r < s1_nr[it]; r++) {
#pragma unroll
                                for (int p = 0; p < PD; p++) {
                                    // the fragment's last use comes before its slot is reloaded, so
                                    // the load can land in the same registers
                                    const double bf0 = b[p * 64];
                                    if (TWO || !(p & 1)) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(af[p], bf0, c0, 0, 0, 0);
                                    else c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(af[p], bf0, c1, 0, 0, 0);
                                    if constexpr (TWO) {
                                        const double bf1 = b[nb16 + p * 64];
                                        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(af[p], bf1, c1, 0, 0, 0);
                                    }
                                    af[p] = wp[ulane + p * 64];
                                }
                                wp += PD * 64; b += PD * 64;
                            }
                            if constexpr (!TWO) {
#pragma unroll
                                for (int r = 0; r < 4; r++) c0[r] += c1[r];
                            }
                            if constexpr (SEP) store(s1_o[it], s1_rows[it], TWO, c0, c1);
                            else { acc0[it] = c0; if constexpr (PAIR) acc1[it] = c1; }
                        }
                    }
                };
                if (PAIR && s1_two_w) stage1(std::true_type{}); else stage1(std::false_type{});
            }
            if constexpr (!SEP) {
                __syncthreads();
#pragma unroll
                for (int it = 0; it < MAXIT; it++) {
                    if (it < nit) {
                        if constexpr (KSREG > 0) store(VX + P.grp_first[grp_r] * nb * 16 + rb_r * 256, nb - rb_r * 16, two_r, acc0[it], acc1[PAIR ? it : 0]);
                        else store(s1_o[it], s1_rows[it], s1_two_w, acc0[it], acc1[PAIR ? it : 0]);
                    }
                }
            }
        }
        STAMP(5);
        __syncthreads();
        STAMP(6);
        // -------------------------------------------------------- relaxation, projection, dual
        if constexpr (CHK) { pin_n = 0.0; pin_l = 0.0; m_atdy = 0.0; }
        {
            // The checking iteration of the variants with many cells per thread takes the cells one at a
            // time: it carries the parts of A' delta_y beside everything else, and what a second cell's
            // operands displace comes back from scratch memory, which at one check per 25 iterations is
            // a cache miss every time.
            constexpr int GP = (CHK && !MIXED && MAXC > 2) ? 1 : G;
#pragma unroll
            for (int c0 = 0; c0 < ((TG_ABLATE & 4) ? 0 : MAXC); c0 += GP) {
                double xt[GP][S], xp[GP][S], bnd[GP][S][2], rr[CHK ? GP : 1][S], bg[GP][D];
                double xn[CHK ? GP : 1][D], pa[CHK ? GP : 1][S], dnx[CHK ? GP : 1][D], dm[(CHK && MIXED) ? 6 : 1];   // CHK: next cell's xt, A' delta_y parts
#pragma unroll
                for (int j = 0; j < GP; j++) {
                    const int cc = c0 + j;
                    if (cc >= MAXC) continue;
                    const double *xin = vx + i_var[cc];
                    const double *xpr = lds + i_prev[cc];
                    if constexpr (CHK) {
                        const bool hn = i_wnext[cc] != ZERO + t;          // the cell has a next step
                        const double *xnx = hn ? xin + S * 16 : lds + ZERO + t;
#pragma unroll
                        for (int k = 0; k < D; k++) {
                            xn[j][k] = xnx[k * 16];
                            if constexpr (KPART == 2) xn[j][k] += hn ? xin[X2 + (S + k) * 16] : 0.0;
                        }
                    }
#pragma unroll
                    for (int k = 0; k < S; k++) {
                        xt[j][k] = xin[k * 16]; xp[j][k] = xpr[k * 16];
                        if constexpr (KPART == 2) {   // + the second k-range part (zero where a row block is not split)
                            xt[j][k] += xin[X2 + k * 16];
                            xp[j][k] += lds[i_prev2[cc] + k * 16];
                        }
                        if constexpr (GP > 1) {
                            if (MIXED && (k == 1 || k == 2)) { bnd[j][k][1] = hiv[cc][k - 1]; bnd[j][k][0] = -bnd[j][k][1]; }
                            else if (REGC) { bnd[j][k][1] = c_hi[cc][k]; bnd[j][k][0] = k == 0 ? c_lo0[cc] : -c_hi[cc][k]; }
                            else { bnd[j][k][0] = clo(cc, k); bnd[j][k][1] = chi(cc, k); }
                            if constexpr (CHK) rr[j][k] = REGC ? c_rho[cc][k] : crho(cc, k);
                        }
                    }
                    if constexpr (GP > 1) {
#pragma unroll
                        for (int k = 0; k < D; k++) bg[j][k] = REGC ? c_bg[cc][k] : lds[i_bg[cc] + k * 16];
                    }
                }
                if constexpr (MIXED) {
                    // the rows hold (xt_x + xt_y) / 2 and (xt_x - xt_y) / 2 of the step (and of the
                    // previous one): back to the two axes
#pragma unroll
                    for (int k = 0; k < S; k++) {
                        const double u = xt[0][k], v = xt[1][k], up = xp[0][k], vp = xp[1][k];
                        xt[0][k] = u + v; xt[1][k] = u - v; xp[0][k] = up + vp; xp[1][k] = up - vp;
                    }
                    if constexpr (CHK) {
#pragma unroll
                        for (int k = 0; k < D; k++) { const double u = xn[0][k], v = xn[1][k]; xn[0][k] = u + v; xn[1][k] = u - v; }
                    }
                }
#pragma unroll
                for (int j = 0; j < GP; j++) {
                    const int cc = c0 + j;
                    if (cc >= MAXC) continue;
                    const int cw = cell_of(cc);
                    // variables and their box rows
#pragma unroll
                    for (int k = 0; k < S; k++) {
                        x[cc][k] = __builtin_fma(alpha, xt[j][k], oma * x[cc][k]);
                        double lo, hi;   // a single cell at a time reads its constants where they are used
                        if constexpr (GP > 1) { lo = bnd[j][k][0]; hi = bnd[j][k][1]; }
                        else if (MIXED && (k == 1 || k == 2)) { hi = hiv[cc][k - 1]; lo = -hi; }
                        else { lo = clo(cc, k); hi = chi(cc, k); }
                        double zo, yo;
                        zy_of(cc, k, lo, hi, zo, yo);
                        const double zr = __builtin_fma(alpha, xt[j][k], oma * zo);
                        const double s = zr + yo;
                        const double zn = clampd(s, lo, hi);
                        if constexpr (CHK) {
                            double d = (GP > 1 ? rr[j][k] : crho(cc, k)) * (zr - zn);
                            if (hi > kInfBound) d = lo < -kInfBound ? 0.0 : fmin(d, 0.0);
                            else if (lo < -kInfBound) d = fmax(d, 0.0);
                            pin_n = fmax(pin_n, fabs(d));
                            pin_l += hi * fmax(d, 0.0) + lo * fmin(d, 0.0);
                            pa[j][k] = d;
                        }
                        if constexpr (SM) zb[cc][k] = s;
                        else { yb[cc][k] = s - zn; zb[cc][k] = zn; }
                    }
                    // dynamics rows of this step: -x_k(i) + sum_l c_kl x_l(i-1) + [k = D-1] dt u(i-1)
#pragma unroll
                    for (int k = 0; k < D; k++) {
                        double zt = -xt[j][k];
#pragma unroll
                        for (int l = k; l < D; l++) zt = __builtin_fma(chain_coef<D>(k, l, dt, c2), xp[j][l], zt);
                        if (k == D - 1) zt = __builtin_fma(dt, xp[j][D], zt);
                        const double b = GP > 1 ? bg[j][k] : lds[i_bg[cc] + k * 16];
                        const double zr = __builtin_fma(alpha, zt, oma * b);
                        const double s = zr + yg[cc][k];
                        yg[cc][k] = s - b;
                        if constexpr (CHK) {
                            const double d = rho_eq * (zr - b);
                            pin_n = fmax(pin_n, fabs(d));
                            pin_l += b * d;
                            pa[j][k] -= d;
                        }
                        lds[WEQ + (cw * D + k) * 16 + t] = rho_eq * (b - yg[cc][k]);
                    }
                    if constexpr (CHK) {
                        // delta_y of the next step's dynamics rows (their bound is zero: only step 0 has one),
                        // operation for operation what the owner of that cell computes above
                        const bool hn = i_wnext[cc] != ZERO + t;
#pragma unroll
                        for (int k = 0; k < D; k++) {
                            double zt = -xn[j][k];
#pragma unroll
                            for (int l = k; l < D; l++) zt = __builtin_fma(chain_coef<D>(k, l, dt, c2), xt[j][l], zt);
                            if (k == D - 1) zt = __builtin_fma(dt, xt[j][D], zt);
                            const double b = 0.0;
                            const double zr = __builtin_fma(alpha, zt, oma * b);
                            dnx[j][k] = hn ? rho_eq * (zr - b) : 0.0;
                        }
                    }
                }
                if constexpr (CHK && MIXED) {
#pragma unroll
                    for (int k = 0; k < 6; k++) dm[k] = 0.0;
                }
                if constexpr (MIXED) {
                    // mixed rows of the step (mpc_12state.cpp:600-630) on the xt of both axes
                    const int mm = c0 >> 1;
                    if (mact[mm]) {
                        const double vxx = xt[0][1], axx = xt[0][2], vyy = xt[1][1], ayy = xt[1][2];
                        const double hx = 0.5 * vxx, hy = 0.5 * vyy, sa = s2 * axx, sb = s2 * ayy;   // halves are exact
                        const double zt6[6] = {__builtin_fma(s3, vxx, hy), __builtin_fma(s3, vxx, -hy), __builtin_fma(s3, vyy, hx),
                                               __builtin_fma(s3, vyy, -hx), sa + sb, sa - sb};
#pragma unroll
                        for (int k = 0; k < 6; k++) {
                            const double bd = k < 4 ? mvb[mm] : mam[mm];
                            const double zr = __builtin_fma(alpha, zt6[k], oma * zm[mm][k]);
                            const double s = zr + ym[mm][k];
                            const double zn = clampd(s, -bd, bd);
                            ym[mm][k] = s - zn;
                            zm[mm][k] = zn;
                            if constexpr (CHK) {
                                double d = P.rho_mx[k] * (zr - zn);
                                if (bd > kInfBound) d = 0.0;
                                pin_n = fmax(pin_n, fabs(d));
                                pin_l += bd * fabs(d);
                                dm[k] = d;
                            }
                            lds[i_mw[mm] + k * 16] = P.rho_mx[k] * (zn - ym[mm][k]);
                        }
                    }
                }
                if constexpr (CHK) {
                    // ||A' delta_y||_inf over the cells' variables: own rows + the next step's dynamics
                    // rows + the step's mixed rows
#pragma unroll
                    for (int j = 0; j < GP; j++) {
                        const int cc = c0 + j;
                        if (cc >= MAXC) continue;
                        double mixc[3] = {0.0, 0.0, 0.0};
                        if constexpr (MIXED) {
                            // axis 0: s3 (d0 + d1) + (d2 - d3) / 2, axis 1: (d0 - d1) / 2 + s3 (d2 + d3)
                            const bool a0 = axis0[cc];
                            mixc[1] = s3 * (a0 ? dm[0] + dm[1] : dm[2] + dm[3]) + 0.5 * (a0 ? dm[2] - dm[3] : dm[0] - dm[1]);
                            mixc[2] = s2 * (a0 ? dm[4] + dm[5] : dm[4] - dm[5]);
                        }
#pragma unroll
                        for (int l = 0; l < D; l++) {
                            double v = pa[j][l];
#pragma unroll
                            for (int k = 0; k <= l; k++) v += chain_coef<D>(k, l, dt, c2) * dnx[j][k];
                            if (MIXED && l >= 1) v += mixc[l];
                            m_atdy = fmax(m_atdy, fabs(v));
                        }
                        m_atdy = fmax(m_atdy, fabs(pa[j][D] + dt * dnx[j][D - 1]));
                    }
                }
            }
        }
        STAMP(7);
    };

    // Checks come at `check`, then every `check_fine` (or every `check` when check_fine is 0), and
    // at max_iter, counted from the start of each slot's instance.  Slots are only refilled at
    // checks, and the host enables refilling only when max_iter is a multiple of check and check
    // a multiple of check_fine, so the tile's checks coincide with every slot's own.
    // The infeasibility certificate is tested where OSQP's own schedule has a check (multiples of
    // `check`, and the last iteration); the finer checks in between only test convergence.
    // phase_mask: which residues of (tile iteration / check_fine) mod (check / check_fine) are a
    // multiple of `check` for some live slot.
    const bool refill = a.queue != nullptr;
    const int nphase = fine > 0 ? max(check / fine, 1) : 1;
    unsigned phase_mask = 1u;
    // iterations count from the start of an instance's run: a later phase of the step-size schedule
    // takes its instances at iteration phase_start (a multiple of check)
    for (int iter = P.phase_start;;) {
        const int gap = (fine > 0 && iter >= check) ? fine : check;
        const int burst = refill ? gap : min(gap, max_iter - iter);
        for (int i = 1; i < burst; i++) {
            step(std::false_type{});
            __syncthreads();
        }
        iter += burst;
        const bool full = fine <= 0 || (refill ? ((phase_mask >> ((iter / fine) % nphase)) & 1u) != 0
                                               : (iter % check == 0 || iter == max_iter));
        if (full) step(std::true_type{}); else step(std::false_type{});
        const int local = iter - slot_start[t];    // iterations of the slot's own instance
        const bool last = local == max_iter;
        const bool eligible = local >= check || last;
        const bool sfull = full && (fine <= 0 || local % check == 0 || last);
        // -------------------------------------------------------- termination check
        {
            CSTAMP(0);
            asm volatile("" : "+v"(gq));
            __syncthreads();
            if (g == 0) {
#pragma unroll
                for (int k = 0; k < 9; k++) if (k != 6) red[k * 16 + t] = 0ull;
            }
            // x -> VX, y of the dynamics / mixed rows -> WEQ / WMX (restored below)
#pragma unroll
            for (int cc = 0; cc < MAXC; cc++) {
                const int cw = cell_q(cc);
                double *xo = vx + i_var[cc];
#pragma unroll
                for (int k = 0; k < S; k++) xo[k * 16] = x[cc][k];
#pragma unroll
                for (int k = 0; k < D; k++) lds[WEQ + (cw * D + k) * 16 + t] = yg[cc][k];   // as it is: read back below
            }
            if (MIXED) {
#pragma unroll
                for (int mm = 0; mm < MAXM; mm++) {
                    if (!mact[mm]) continue;
#pragma unroll
                    for (int k = 0; k < 6; k++) lds[i_mw[mm] + k * 16] = P.rho_mx[k] * ym[mm][k];
                }
            }
            __syncthreads();
            CSTAMP(1);
            double m_rp = 0, m_ax = 0, m_z = 0, m_rd = 0, m_px = 0, m_aty = 0;
#pragma unroll
            for (int cc = 0; cc < MAXC; cc++) {
                const int c = cell_q(cc);
                if (c < ncell) {
                    const int ax = axis_of(c), st = c - ax * N1;
                    const double *xin = vx + i_var[cc];
                    const double *xpr = lds + i_prev[cc];
                    // x of the cell read back from VX: the registers that held it are free from the
                    // copy above to the end of the exit code (the check is the one place short of them)
                    double xc[S];
#pragma unroll
                    for (int k = 0; k < S; k++) xc[k] = xin[k * 16];
                    double pdiag = 0.0;
#pragma unroll
                    for (int k = 0; k < NAX; k++) if (ax == k) pdiag = P.pdiag[k];
                    double zc[S], yc[S];   // z and y/rho of the cell's box rows
#pragma unroll
                    for (int k = 0; k < S; k++) {
                        double lo = 0.0, hi = 0.0;
                        if constexpr (SM) {
                            if (CAX || REGC) box_of(cc, k, lo, hi);
                            else { lo = lds[i_ct[cc] + k * 3 + 1]; hi = lds[i_ct[cc] + k * 3 + 2]; }
                        }
                        zy_of(cc, k, lo, hi, zc[k], yc[k]);
                    }
                    double yn[D], mixc[3] = {0.0, 0.0, 0.0};
#pragma unroll
                    for (int k = 0; k < D; k++) yn[k] = rho_eq * lds[i_wnext[cc] + k * 16];
                    if (MIXED) {
                        const double *wm = lds + i_mx[cc];
                        const double w0 = wm[0], w1 = wm[16], w2 = wm[32], w3 = wm[48], w4 = wm[64], w5 = wm[80];
                        const bool a0 = axis0[cc];
                        mixc[1] = s3 * (a0 ? w0 + w1 : w2 + w3) + 0.5 * (a0 ? w2 - w3 : w0 - w1);
                        mixc[2] = s2 * (a0 ? w4 + w5 : w4 - w5);
                    }
                    // dual residual P x + q + A' y on the state slots
#pragma unroll
                    for (int l = 0; l < D; l++) {
                        const double px = l == 0 ? pdiag * xc[0] : 0.0;
                        double aty = crho_t(cc, l) * yc[l] - rho_eq * lds[WEQ + (c * D + l) * 16 + t];
#pragma unroll
                        for (int k = 0; k <= l; k++) aty += chain_coef<D>(k, l, dt, c2) * yn[k];
                        if (MIXED && l >= 1) aty += mixc[l];
                        const double qq = l == 0 ? (QL ? lds[QV + c * 16 + t] : q[QL ? 0 : cc]) : 0.0;
                        m_rd = fmax(m_rd, fabs(px + qq + aty));
                        m_px = fmax(m_px, fabs(px));
                        m_aty = fmax(m_aty, fabs(aty));
                    }
                    if (st < N) { // input slot: P = r I + smooth * D'D (first difference of inputs)
                        double px = P.r * xc[D];
                        if (P.smooth != 0.0) {
                            const double um = st > 0 ? xpr[D * 16] : xc[D];
                            const double up = st < N - 1 ? xin[(D + S) * 16] : xc[D];
                            px += P.smooth * ((xc[D] - um) + (xc[D] - up));
                        }
                        const double aty = crho_t(cc, D) * yc[D] + dt * yn[D - 1];
                        m_rd = fmax(m_rd, fabs(px + aty));
                        m_px = fmax(m_px, fabs(px));
                        m_aty = fmax(m_aty, fabs(aty));
                    }
                    // primal residual: box rows (A x = x) ...
#pragma unroll
                    for (int k = 0; k < S; k++) {
                        if (k < D || st < N) {
                            m_rp = fmax(m_rp, fabs(xc[k] - zc[k]));
                            m_ax = fmax(m_ax, fabs(xc[k]));
                            m_z = fmax(m_z, fabs(zc[k]));
                        }
                    }
                    // ... dynamics rows ...
#pragma unroll
                    for (int k = 0; k < D; k++) {
                        double axv = -xc[k];
#pragma unroll
                        for (int l = k; l < D; l++) axv += chain_coef<D>(k, l, dt, c2) * xpr[l * 16];
                        if (k == D - 1) axv += dt * xpr[D * 16];
                        const double b = lds[i_bg[cc] + k * 16];
                        m_rp = fmax(m_rp, fabs(axv - b));
                        m_ax = fmax(m_ax, fabs(axv));
                        m_z = fmax(m_z, fabs(b));
                    }
                }
            }
            // ... mixed rows
            if (MIXED) {
#pragma unroll
                for (int mm = 0; mm < MAXM; mm++) {
                    if (!mact[mm]) continue;
                    const double *x0p = vx + i_mv[mm][0], *x1p = vx + i_mv[mm][1];
                    const double vxx = x0p[16], axx = x0p[32], vyy = x1p[16], ayy = x1p[32];
                    const double ax6[6] = {s3 * vxx + 0.5 * vyy, s3 * vxx - 0.5 * vyy, 0.5 * vxx + s3 * vyy,
                                           -0.5 * vxx + s3 * vyy, s2 * axx + s2 * ayy, s2 * axx - s2 * ayy};
#pragma unroll
                    for (int k = 0; k < 6; k++) {
                        m_rp = fmax(m_rp, fabs(ax6[k] - zm[mm][k]));
                        m_ax = fmax(m_ax, fabs(ax6[k]));
                        m_z = fmax(m_z, fabs(zm[mm][k]));
                    }
                }
            }
            CSTAMP(2);
            m_rp = wave_gmax(m_rp); m_ax = wave_gmax(m_ax); m_z = wave_gmax(m_z);
            m_rd = wave_gmax(m_rd); m_px = wave_gmax(m_px); m_aty = wave_gmax(m_aty);
            pin_n = wave_gmax(pin_n); m_atdy = wave_gmax(m_atdy);
            pin_l += __shfl_xor(pin_l, 16);
            pin_l += __shfl_xor(pin_l, 32);
            if ((lane >> 4) == 0) {
                lds_max(&red[0 * 16 + t], m_rp); lds_max(&red[1 * 16 + t], m_ax);
                lds_max(&red[2 * 16 + t], m_z);  lds_max(&red[3 * 16 + t], m_rd);
                lds_max(&red[4 * 16 + t], m_px); lds_max(&red[5 * 16 + t], m_aty);
                lds_max(&red[7 * 16 + t], pin_n); lds_max(&red[8 * 16 + t], m_atdy);
                rsum[wave * 16 + t] = pin_l;
            }
            __syncthreads();
            pin_n = __longlong_as_double((long long)red[7 * 16 + t]);
            m_atdy = __longlong_as_double((long long)red[8 * 16 + t]);
            pin_l = 0.0;
#pragma unroll
            for (int w = 0; w < kWaves; w++) pin_l += rsum[w * 16 + t];
            m_rp = __longlong_as_double((long long)red[0 * 16 + t]);
            m_ax = __longlong_as_double((long long)red[1 * 16 + t]);
            m_z = __longlong_as_double((long long)red[2 * 16 + t]);
            m_rd = __longlong_as_double((long long)red[3 * 16 + t]);
            m_px = __longlong_as_double((long long)red[4 * 16 + t]);
            m_aty = __longlong_as_double((long long)red[5 * 16 + t]);
            const double eps_p = P.eps_abs + P.eps_rel * fmax(m_ax, m_z);
            const double qn = __longlong_as_double((long long)red[6 * 16 + t]);
            const double eps_d = P.eps_abs + P.eps_rel * fmax(fmax(m_px, m_aty), qn);
            const bool conv = eligible && (m_rp < eps_p) && (m_rd < eps_d);
            // OSQP's certificate of primal infeasibility (is_primal_infeasible, only tried while
            // the primal residual is above its tolerance): delta_y, projected on the polar of the
            // recession cone of [l, u], separates the constraint set from the range of A
            const bool pinf = sfull && eligible && !(m_rp < eps_p) && eps_pinf > 0.0 && pin_n > kDivisionTol &&
                              pin_l < -eps_pinf * pin_n && m_atdy < eps_pinf * pin_n;
            // an instance still running at the end of this launch's phase moves on to the next
            // operator: its iterates go to the warm rows and its number to the next launch's list
            const bool fin = conv || pinf || last;
            const bool pexit = !fin && a.next_list != nullptr && local >= P.phase_end;
            CSTAMP(3);
            if (!done && (fin || pexit)) {
                done = true;
                const int inst = slot_inst[t];
                int verdict = conv ? TG_DEV_SOLVED : (pinf ? TG_DEV_PRIMAL_INFEASIBLE : TG_DEV_MAX_ITER);
                if (fin && !conv && !pinf) {
                    // max_iter reached: OSQP tests once more with every tolerance times ten and reports
                    // "solved inaccurate" / "primal infeasible inaccurate" if that passes (osqp_solve's
                    // last step; osqp-eigen's solve() returns false for these as for max_iter)
                    const double ea = P.eps_abs * 10.0, er = P.eps_rel * 10.0, ei = eps_pinf * 10.0;
                    const bool prim_ok = m_rp < ea + er * fmax(m_ax, m_z);
                    const bool dual_ok = m_rd < ea + er * fmax(fmax(m_px, m_aty), qn);
                    if (prim_ok && dual_ok) verdict = TG_DEV_SOLVED_INACCURATE;
                    else if (sfull && !prim_ok && eps_pinf > 0.0 && pin_n > kDivisionTol && pin_l < -ei * pin_n && m_atdy < ei * pin_n)
                        verdict = TG_DEV_PRIMAL_INFEASIBLE_INACCURATE;
                }
                // solution out (extractSolution, mpc_6dof.cpp:653; 12-state :1456-1551): for solved
                // instances only -- mpcLoop returns before extractSolution when solve() is false, so a
                // failed instance keeps the trajectories of its last successful solve
                const bool wr = fin && verdict == TG_DEV_SOLVED;
#pragma unroll
                for (int cc = 0; cc < MAXC; cc++) {
                    const int c = cell_q(cc);
                    if (wr && c < ncell) {
                        const int ax = axis_of(c), st = c - ax * N1;
                        int xoff = 0, uoff = 0;
#pragma unroll
                        for (int k = 0; k < NAX; k++) if (ax == k) { xoff = P.X_off[k]; uoff = P.U_off[k]; }
#pragma unroll
                        for (int k = 0; k < D; k++)
                            a.X[(size_t)(st * P.X_stride + xoff + k * P.X_slot) * a.ld + inst] = vx[i_var[cc] + k * 16];
                        if (st < N) a.U[(size_t)(st * P.U_stride + uoff) * a.ld + inst] = vx[i_var[cc] + D * 16];
                    }
                }
                if (g == 0 && fin) {
                    if (a.status) a.status[inst] = verdict;
                    if (a.iters) a.iters[inst] = local;
                    if (a.res) { a.res[inst] = m_rp; a.res[(size_t)a.res_ld + inst] = m_rd; }
                }
                if (g == 0 && pexit) a.next_list[atomicAdd(a.next_count, 1)] = inst;
                if constexpr (MIXED) {
                    // the two-axis stage of the 12-state chain, whose next stage may start for the
                    // vehicles that are done while the others continue: a handed-over vehicle's status
                    // says "not decided yet" until a later launch of this solve decides, and launches
                    // that are asked to list what they solve
                    if (g == 0 && pexit && a.status) a.status[inst] = TG_DEV_UNSOLVED;
                    if (g == 0 && fin && verdict == TG_DEV_SOLVED && a.done_list != nullptr)
                        a.done_list[atomicAdd(a.done_count, 1)] = inst;
                }
                if (a.warm_out || pexit) {
#pragma unroll
                    for (int cc = 0; cc < MAXC; cc++) {
                        const int c = cell_q(cc);
                        if (c < ncell) {
                            const int ax = axis_of(c), st = c - ax * N1;
                            int abase = 0;
#pragma unroll
                            for (int k = 0; k < NAX; k++) if (ax == k) abase = P.axis_base[k];
                            const size_t w0 = (size_t)(abase + st * S) * a.ldw + inst;
#pragma unroll
                            for (int k = 0; k < S; k++) {
                                double lo = 0.0, hi = 0.0, zv, yv;
                                if constexpr (SM) {
                                    if (CAX || REGC) box_of(cc, k, lo, hi);
                                    else { lo = lds[i_ct[cc] + k * 3 + 1]; hi = lds[i_ct[cc] + k * 3 + 2]; }
                                }
                                zy_of(cc, k, lo, hi, zv, yv);
                                a.warm[w0 + (size_t)k * a.ldw] = vx[i_var[cc] + k * 16];
                                a.warm[w0 + (size_t)(n_int + k) * a.ldw] = zv;
                                a.warm[w0 + (size_t)(2 * n_int + k) * a.ldw] = crho_t(cc, k) * yv;   // y itself: the same under every operator
                            }
                            const size_t e0 = (size_t)(3 * n_int + c * D) * a.ldw + inst;
#pragma unroll
                            for (int k = 0; k < D; k++) {
                                a.warm[e0 + (size_t)k * a.ldw] = lds[i_bg[cc] + k * 16];
                                a.warm[e0 + (size_t)(ncell * D + k) * a.ldw] = lds[WEQ + (c * D + k) * 16 + t];
                            }
                        }
                    }
                    if (MIXED) {
#pragma unroll
                        for (int mm = 0; mm < MAXM; mm++) {
                            if (!mact[mm]) continue;
                            const int m = g + kStride * mm;
                            const size_t m0 = (size_t)(3 * n_int + 2 * ncell * D + m * 6) * a.ldw + inst;
#pragma unroll
                            for (int k = 0; k < 6; k++) {
                                a.warm[m0 + (size_t)k * a.ldw] = zm[mm][k];
                                a.warm[m0 + (size_t)(N1 * 6 + k) * a.ldw] = P.rho_mx[k] * ym[mm][k];
                            }
                        }
                    }
                }
            }
            CSTAMP(4);
            // x back into its registers (VX has held it since the start of the check)
#pragma unroll
            for (int cc = 0; cc < MAXC; cc++) {
#pragma unroll
                for (int k = 0; k < S; k++) x[cc][k] = vx[i_var[cc] + k * 16];
            }
            // and y of the dynamics rows, from WEQ
#pragma unroll
            for (int cc = 0; cc < MAXC; cc++) {
                const int cw = cell_q(cc);
#pragma unroll
                for (int k = 0; k < D; k++) yg[cc][k] = lds[WEQ + (cw * D + k) * 16 + t];
            }
            // WEQ / WMX held y during the check: restore w = rho (z - y/rho)
#pragma unroll
            for (int cc = 0; cc < MAXC; cc++) {
                const int cw = cell_q(cc);
#pragma unroll
                for (int k = 0; k < D; k++)
                    lds[WEQ + (cw * D + k) * 16 + t] = rho_eq * (lds[i_bg[cc] + k * 16] - yg[cc][k]);
            }
            if (MIXED) {
#pragma unroll
                for (int mm = 0; mm < MAXM; mm++) {
                    if (!mact[mm]) continue;
#pragma unroll
                    for (int k = 0; k < 6; k++) lds[i_mw[mm] + k * 16] = P.rho_mx[k] * (zm[mm][k] - ym[mm][k]);
                }
            }
            CSTAMP(5);
            // slots whose instance has terminated take the next unassigned one
            const bool need = done && !exhausted;
            if (__syncthreads_or(need ? 1 : 0)) {
                // one ticket draw per workgroup: the slots that need an instance take consecutive
                // ones in slot order, so that their reads and writes of the batch arrays (instance
                // minor) fall into the same cache lines
                int *next = (int *)rsum;
                if (wave == 0) {
                    const unsigned long long mask = __ballot(need && g == 0);
                    int base = 0;
                    if (lane == 0) next[16] = 0;     // the word the slots' check phases are gathered in below
                    if (lane == 0) base = atomicAdd(a.queue, (int)__popcll(mask));
                    base = __shfl(base, 0);
                    if (need && g == 0) {
                        const int ni = a.queue_base + base + (int)__popcll(mask & ((1ull << t) - 1ull));
                        next[t] = ni < batch_size() ? (a.list ? a.list[ni] : ni) : -1;
                        red[6 * 16 + t] = 0ull;
                    }
                }
                __syncthreads();
                bool fresh = false;
                const int ni = need ? next[t] : -1;
                if (need) {
                    if (ni < 0) exhausted = true; else fresh = true;
                }
                double qf = load(fresh, ni);
                qf = wave_gmax(qf);
                if (fresh && (lane >> 4) == 0) lds_max(&red[6 * 16 + t], qf);
                if (fresh && g == 0) { slot_inst[t] = ni; slot_start[t] = iter - P.phase_start; }
                if (fine > 0) {
                    // bitwise OR over the live slots (__syncthreads_or only tells whether any bit is set)
                    const int st0 = fresh ? iter - P.phase_start : slot_start[t];
                    if (g == 0 && !done) atomicOr((unsigned *)&next[16], 1u << ((st0 / fine) % nphase));
                    __syncthreads();
                    phase_mask = (unsigned)next[16];
                } else {
                    __syncthreads();
                }
            }
#ifdef TG_STAMP
            CSTAMP(6);
            chk_acc[7] += 1;
            stamp_last = chk_last;     // the check is not part of the next iteration's first section
#endif
            if (__syncthreads_and((done && exhausted) ? 1 : 0)) break;
        }
    }
#ifdef TG_STAMP
    // waves 0 and 4 (the two waves of SIMD 0) of workgroup 0: columns 0..15 of row 0 belong to its own instances
    if (blockIdx.x == 0 && lane == 0 && (wave & 3) == 0)
        for (int i = 0; i < 8; i++) { a.U[(wave >> 2) * 8 + i] = (double)stamp_acc[i]; a.X[(wave >> 2) * 8 + i] = (double)chk_acc[i]; }
#endif
}

// ---------------------------------------------------------------- full q, l, u per instance
// one thread per (row block, instance); rows of q, l, u in the reference's ordering
__global__ __launch_bounds__(256) void assemble_kernel(const AsmDev P, const BatchArgs a,
                                                        double *__restrict__ q,
                                                        double *__restrict__ l,
                                                        double *__restrict__ u)
{
#pragma clang fp contract(off)
    const int inst = blockIdx.x * blockDim.x + threadIdx.x;
    if (inst >= a.B) return;
    const int r0 = blockIdx.y * P.rows_per_block;
    const int r1 = min(r0 + P.rows_per_block, max(P.n, P.m));
    for (int r = r0; r < r1; r++) {
        if (r < P.n && q) {
            double v = 0.0;
            const int rr = P.q_ref_row[r];
            if (rr >= 0) v = P.q_w[r] * a.ref[(size_t)rr * a.ref_ld + inst];
            q[(size_t)r * a.ld + inst] = v;
        }
        if (r < P.m && l) {
            double lo = P.l_t[r], hi = P.u_t[r];
            const int xr = P.x0_row[r];
            if (xr >= 0) { lo = -a.x0[(size_t)xr * a.ld + inst]; hi = lo; }
            const int ob = P.ov_box[r];
            if (ob >= 0) { hi = a.ov_box[(size_t)ob * a.ov_ld + inst]; lo = -1.0 * hi; }
            const int om = P.ov_mixed[r];
            if (om >= 0) { hi = a.ov_mixed[(size_t)om * a.ov_ld + inst]; lo = -1.0 * hi; }
            l[(size_t)r * a.ld + inst] = lo;
            u[(size_t)r * a.ld + inst] = hi;
        }
    }
}

// ---------------------------------------------------------------- computeXYBounds
// X holds the Z-stage solution in rows 12 i + 6..8.  Output rows: step i-1 (i = 1..N), 6 each.
__global__ __launch_bounds__(256) void xy_bounds_kernel(int B, int ld, int N, double xy_max_vel,
                                                         double xy_max_accel, double z_max_accel,
                                                         const double *__restrict__ X,
                                                         double *__restrict__ xy_max,
                                                         double *__restrict__ mixed_max)
{
#pragma clang fp contract(off)   // same roundings as the reference's separate multiply and subtract
    const int inst = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y + 1;
    if (inst >= B || i > N) return;
    const double v_zt = X[(size_t)(12 * i + 7) * ld + inst];
    double vb;
    if (v_zt < 0) {
        vb = xy_max_vel;
    } else {
        const double a_zt = X[(size_t)(12 * i + 8) * ld + inst];
        const double d = a_zt / z_max_accel;
        vb = xy_max_vel * 0.9;
        if ((1 - d * d) >= 0) vb = xy_max_vel * sqrt(1 - d * d);
    }
    double *b = xy_max + (size_t)6 * (i - 1) * ld + inst;
    double *mx = mixed_max + (size_t)6 * (i - 1) * ld + inst;
    b[0] = TG_DEV_INFTY; b[(size_t)ld] = vb; b[(size_t)2 * ld] = xy_max_accel;
    b[(size_t)3 * ld] = TG_DEV_INFTY; b[(size_t)4 * ld] = vb; b[(size_t)5 * ld] = xy_max_accel;
    mx[0] = vb; mx[(size_t)ld] = vb; mx[(size_t)2 * ld] = vb; mx[(size_t)3 * ld] = vb;
    mx[(size_t)4 * ld] = xy_max_accel; mx[(size_t)5 * ld] = xy_max_accel;
}

// ---------------------------------------------------------------- computeYawRefTrajectory
__global__ __launch_bounds__(256) void yaw_ref_kernel(int B, int ld, int N,
                                                       const double *__restrict__ ref,
                                                       const double *__restrict__ X,
                                                       double *__restrict__ yaw_ref)
{
#pragma clang fp contract(off)
    const int inst = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (inst >= B || i > N) return;
    const double xt = ref[(size_t)(12 * i + 0) * ld + inst], yt = ref[(size_t)(12 * i + 3) * ld + inst];
    const double xi = X[(size_t)(12 * i + 0) * ld + inst], yi = X[(size_t)(12 * i + 3) * ld + inst];
    yaw_ref[(size_t)i * ld + inst] = atan2(yt - yi, xt - xi);
}

// ---------------------------------------------------------------- merge of the 12-state chain
// MPC12STATE::extractSolution (mpc_12state.cpp:1528-1551) runs after the third stage has solved, so
// _x_opt / _u_opt change only when all three did (mpcLoop returns at the first failing stage,
// :1405-1449).  The stages leave their solutions in Xs / Us; rows [r0, r0 + rows_per_block) of the
// stacked [X; U] of every instance with three "solved" statuses go to the caller's arrays, the
// columns of the other instances are left as they are.
__global__ __launch_bounds__(256) void merge12_kernel(int B, int ld, int rowsX, int rowsU, int rows_per_block,
                                                       const double *__restrict__ Xs, const double *__restrict__ Us,
                                                       const int *__restrict__ status,
                                                       double *__restrict__ X, double *__restrict__ U, int *clear)
{
    // (clear: a counter of the chain that has been consumed by now and starts the next chain at zero)
    if (clear != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *clear = 0;
    const int inst = blockIdx.x * blockDim.x + threadIdx.x;
    if (inst >= B) return;
    const bool ok = status[inst] == TG_DEV_SOLVED && status[(size_t)ld + inst] == TG_DEV_SOLVED &&
                    status[(size_t)2 * ld + inst] == TG_DEV_SOLVED;
    if (!ok) return;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(r0 + rows_per_block, rowsX + rowsU);
    for (int r = r0; r < r1; r++) {
        if (r < rowsX) X[(size_t)r * ld + inst] = Xs[(size_t)r * ld + inst];
        else U[(size_t)(r - rowsX) * ld + inst] = Us[(size_t)(r - rowsX) * ld + inst];
    }
}

// ---------------------------------------------------------------- closed-loop roll-out helpers
// Plant update with the first control of the horizon, x0 <- A x0 + B u(0): the rule the reference's
// test mains roll the solution out with (test_mpc_6dof.cpp:115-122, test_mpc_12state.cpp:193-197).
struct PlantDev {
    int nx, nu;
    double A[12 * 12], Bm[12 * 4];   // row-major, model-level (12-state: block diagonal)
};

// An instance whose last solve did not reach "solved" in every stage received no new plan (the
// reference publishes nothing then, mpc_ros.cpp:428-432): it keeps following the last plan it got,
// i.e. its U moves one step up (the last step becomes zero: beyond the plan's end it coasts) and the
// new first control is applied.  status: [nstat][ld] or null (every instance counts as solved).
__global__ __launch_bounds__(256) void plant_step_kernel(const PlantDev P, int B, int ld, int N,
                                                          double *__restrict__ x0,
                                                          double *__restrict__ U,
                                                          const int *__restrict__ status, int nstat)
{
#pragma clang fp contract(off)
    const int inst = blockIdx.x * blockDim.x + threadIdx.x;
    if (inst >= B) return;
    bool ok = true;
    if (status)
        for (int k = 0; k < nstat; k++) ok = ok && status[(size_t)k * ld + inst] == TG_DEV_SOLVED;
    if (!ok) {
        for (int i = 0; i + 1 < N; i++)
            for (int k = 0; k < P.nu; k++)
                U[(size_t)(i * P.nu + k) * ld + inst] = U[(size_t)((i + 1) * P.nu + k) * ld + inst];
        for (int k = 0; k < P.nu; k++) U[(size_t)((N - 1) * P.nu + k) * ld + inst] = 0.0;
    }
    double x[12], u[4], y[12];
    for (int c = 0; c < P.nx; c++) x[c] = x0[(size_t)c * ld + inst];
    for (int k = 0; k < P.nu; k++) u[k] = U[(size_t)k * ld + inst];
    for (int r = 0; r < P.nx; r++) {
        double s = 0.0;
        for (int c = 0; c < P.nx; c++) s += P.A[r * P.nx + c] * x[c];
        for (int k = 0; k < P.nu; k++) s += P.Bm[r * P.nu + k] * u[k];
        y[r] = s;
    }
    for (int r = 0; r < P.nx; r++) x0[(size_t)r * ld + inst] = y[r];
}

// Receding horizon: the reference block of step i+1 moves to step i; the last block is continued
// linearly (ref(N) + (ref(N) - ref(N-1))), which is exact for the constant-velocity references of
// the synthetic workloads.  One thread per (state component, instance).
__global__ __launch_bounds__(256) void ref_shift_kernel(int B, int ld, int nx, int N, double *__restrict__ ref)
{
#pragma clang fp contract(off)
    const int inst = blockIdx.x * blockDim.x + threadIdx.x;
    const int a = blockIdx.y;
    if (inst >= B || a >= nx) return;
    double prev = ref[(size_t)a * ld + inst], cur = prev;
    for (int i = 0; i < N; i++) {
        prev = cur;
        cur = ref[(size_t)((i + 1) * nx + a) * ld + inst];
        ref[(size_t)(i * nx + a) * ld + inst] = cur;
    }
    if (N >= 1) ref[(size_t)(N * nx + a) * ld + inst] = cur + (cur - prev);
}

// ---------------------------------------------------------------- static scratch
// Device memory of the entry point that may not allocate (tg_mpc12_solve_batch_static): the
// inter-stage arrays, the stage status rows and the queue counters live in this array instead of
// in hipMalloc'ed buffers.  Every GPU holds its own copy at its own address; in host code the name
// `scratch` is only a handle for hipGetSymbolAddress, never a device address: the address is looked
// up once per device (static_scratch_init, called with that device current) and everything else --
// kernel arguments, hipMemsetAsync -- uses the looked-up pointer.
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];
static std::mutex g_scratch_mu;
static void *g_scratch_ptr[64] = {nullptr};

hipError_t static_scratch_init(int dev)
{
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (g_scratch_ptr[dev]) return hipSuccess;
    void *ptr = nullptr;
    hipError_t e = hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (!ptr) return hipErrorInvalidSymbol;
    g_scratch_ptr[dev] = ptr;
    return hipSuccess;
}

// the device's copy of the array, or null before static_scratch_init(dev)
void *static_scratch_ptr(int dev)
{
    if (dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    return g_scratch_ptr[dev];
}

// ---------------------------------------------------------------- host-callable launchers
// ---------------------------------------------------------------------------------------------
// Order in which a refilled grid takes the instances.  The batch ends when its slowest instance
// does, so an instance that needs many iterations should start early; which ones do is not known
// beforehand, but on these problems they are among those whose reference strays farthest from
// where the initial state coasts to (score below: of 16384 seeded N = 50 instances every one that
// needed >= 125 iterations is in the top 5.4 % by score).  The top sixteenth by score, cut at a
// histogram bin, is taken first, the rest after it; within either part blocks of 256 instances
// keep the batch order, so neighbouring slots still share cache lines.  An instance's result does
// not depend on when it is solved.
__global__ __launch_bounds__(256) void order_score_kernel(const StageDev P, const BatchArgs a, double *__restrict__ score,
                                                          unsigned long long *__restrict__ hdr)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    double sc = 0.0;
    if (i < a.B) {
        const bool gated = a.gate && a.gate[i] != TG_DEV_SOLVED;
        if (!gated) {
            for (int ax = 0; ax < P.nax; ax++) {
                const size_t x0r = (size_t)P.x0_row[ax] * a.ld + i;
                const double p0 = a.x0[x0r], v0 = a.x0[x0r + (size_t)P.x0_slot * a.ld];
                const double a0 = P.D == 3 ? a.x0[x0r + (size_t)2 * P.x0_slot * a.ld] : 0.0;
                for (int k = 0; k <= P.N; k++) {
                    const double tk = k * P.dt;
                    const double coast = p0 + tk * (v0 + 0.5 * tk * a0);
                    sc = fmax(sc, fabs(a.ref[(size_t)(k * P.ref_stride + P.ref_off[ax]) * a.ref_ld + i] - coast));
                }
            }
        }
        score[i] = sc;
    }
    __shared__ double wmax[4];
    for (int o = 32; o > 0; o >>= 1) sc = fmax(sc, __shfl_xor(sc, o));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = sc;
    __syncthreads();
    // bit patterns of non-negative doubles order like unsigned integers
    if (threadIdx.x == 0)
        atomicMax(hdr, (unsigned long long)__double_as_longlong(fmax(fmax(wmax[0], wmax[1]), fmax(wmax[2], wmax[3]))));
}

__device__ __forceinline__ int order_bin(double sc, double top)
{
    if (!(top > 0.0)) return 0;
    const double r = sc / top;
    return r >= 1.0 ? kOrderBins - 1 : (r > 0.0 ? (int)(r * kOrderBins) : 0);
}

__global__ __launch_bounds__(256) void order_hist_kernel(int B, const double *__restrict__ score, unsigned long long *__restrict__ hdr)
{
    __shared__ unsigned int h[kOrderBins];
    h[threadIdx.x] = 0u;
    __syncthreads();
    const double top = __longlong_as_double((long long)hdr[0]);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < B) atomicAdd(&h[order_bin(score[i], top)], 1u);
    __syncthreads();
    unsigned int *hist = (unsigned int *)(hdr + 1);
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

__global__ __launch_bounds__(256) void order_partition_kernel(int B, const double *__restrict__ score,
                                                              unsigned long long *__restrict__ hdr, int *__restrict__ list)
{
    __shared__ unsigned int suf[kOrderBins];
    __shared__ int cut, base[2], wcnt[2][4];
    const unsigned int *hist = (const unsigned int *)(hdr + 1);
    unsigned int *ctr = (unsigned int *)(hdr + 1) + kOrderBins;
    suf[threadIdx.x] = hist[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        // first bin from which on at most a sixteenth of the batch lies
        unsigned int run = 0;
        int c = kOrderBins;
        for (int b = kOrderBins - 1; b >= 0; b--) {
            run += suf[b];
            if (run > (unsigned int)B / 16u) break;
            c = b;
        }
        cut = c;
        unsigned int hard = 0;
        for (int b = c; b < kOrderBins; b++) hard += suf[b];
        base[1] = (int)hard;        // the rest starts after the early part
    }
    __syncthreads();
    const double top = __longlong_as_double((long long)hdr[0]);
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool in = i < B;
    const bool early = in && order_bin(score[i], top) >= cut;
    const unsigned long long me = __ballot(early), ml = __ballot(in && !early);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { wcnt[0][w] = (int)__popcll(me); wcnt[1][w] = (int)__popcll(ml); }
    __syncthreads();
    if (threadIdx.x < 2) {
        const int k = threadIdx.x;
        const int n = wcnt[k][0] + wcnt[k][1] + wcnt[k][2] + wcnt[k][3];
        const int at = n ? (int)atomicAdd(&ctr[k], (unsigned int)n) : 0;
        base[k] = (k ? base[1] : 0) + at;
    }
    __syncthreads();
    if (in) {
        const int k = early ? 0 : 1;
        int pos = base[k] + (int)__popcll((early ? me : ml) & ((1ull << lane) - 1ull));
        for (int v = 0; v < w; v++) pos += wcnt[k][v];
        list[pos] = i;
    }
}

static hipError_t launch_order(const StageDev &P, const BatchArgs &a, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(a.order_hdr, 0, kOrderHdrBytes, s);
    if (e != hipSuccess) return e;
    const int blocks = (a.B + 255) / 256;
    hipLaunchKernelGGL(order_score_kernel, dim3(blocks), dim3(256), 0, s, P, a, a.order_score, a.order_hdr);
    hipLaunchKernelGGL(order_hist_kernel, dim3(blocks), dim3(256), 0, s, a.B, a.order_score, a.order_hdr);
    hipLaunchKernelGGL(order_partition_kernel, dim3(blocks), dim3(256), 0, s, a.B, a.order_score, a.order_hdr, a.order_list);
    return hipGetLastError();
}

size_t admm_lds_bytes(const StageDev &P)
{
    const int S = P.D + 1, ncell = P.nax * (P.N + 1);
    size_t rows = (size_t)(P.separate_xt ? 2 + (P.kpart == 2 ? 1 : 0) : 1) * (P.n_int + S) + (size_t)ncell * P.D + P.D +
                  (P.mixed ? (size_t)(P.N + 1) * 6 + 6 : 0) + 8 + (size_t)P.nax * P.D +
                  (P.lr ? (size_t)P.lr_np * (P.n_int / P.nb) * P.sb_rows : 0) + (P.ql ? (size_t)ncell + 1 : 0);
    return rows * 16 * sizeof(double) + ((size_t)P.nax * S * 3 + 2) * sizeof(double) + (9 + kAdmmWaves + 1) * 16 * sizeof(double);
}

template <int NT, int D, int NAX, bool MIXED, int MAXC, int MAXIT, bool PAIR, int KSREG, bool SEP, bool LR = false,
          bool SM = false, bool QL = false, int UK = 2, bool SMJ = false, int PDT = kRingDepth, int KPART = 1, int UNI = 0>
static hipError_t launch_admm_t(const StageDev &P, const BatchArgs &a, hipStream_t s, const char **why)
{
    auto kern = admm_kernel<NT, D, NAX, MIXED, MAXC, MAXIT, PAIR, KSREG, SEP, LR, SM, QL, UK, SMJ, PDT, KPART, UNI>;
    if (KSREG == 0 && P.ring != PDT) { *why = "internal: ring depth of the stage and of the kernel variant differ"; return hipErrorInvalidValue; }
    if ((P.kpart == 2) != (KPART == 2)) { *why = "internal: k-range parts of the stage and of the kernel variant differ"; return hipErrorInvalidValue; }
    if ((P.sm != 0) != SM || (P.ql != 0) != QL) { *why = "internal: box-row form of the stage and of the kernel variant differ"; return hipErrorInvalidValue; }   // the LDS layout follows P.ql
    if (KSREG == 0 && (!P.Wfrag || (LR && (P.lr_np > 2 || P.lr_ukp != UK)))) { *why = "internal: fragment streams of the stage do not match the kernel variant"; return hipErrorInvalidValue; }
    const size_t lds_bytes = admm_lds_bytes(P);
    // the dynamic LDS limit of a kernel is per device and sticky: raise it when it has to grow
    // (per-variant caches, shared by every handle and host thread of the process)
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    static int lds_set[64] = {0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || lds_set[dev] < (int)lds_bytes) {
        e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) lds_set[dev] = (int)lds_bytes;
    }
    const int ntiles = (a.B + kTileDev - 1) / kTileDev;
    int grid = ntiles;
    BatchArgs b = a;
    b.queue = nullptr;
    b.queue_base = 0;
    // a.list from the caller: the instances a previous phase of the step-size schedule handed over
    // (a.count of them); such a launch takes them in that order
    if (a.count == nullptr) b.list = nullptr;
    // A grid no larger than what the device holds at once, its slots refilled from a queue, keeps
    // every slot busy until the batch runs out; a tile per workgroup makes each tile wait for its
    // slowest instance.  The kernel's slot schedule needs the checks aligned (see admm_kernel).
    const bool aligned = P.check > 0 && P.max_iter % P.check == 0 &&
                         (P.check_fine <= 0 || (P.check % P.check_fine == 0 && P.check / P.check_fine <= 32));
    if (a.queue_mem && aligned) {
        int cap = a.grid_cap;
        if (cap <= 0) {
            static int resident[64] = {0}, resident_lds[64] = {0};
            if (dev < 0 || dev >= 64 || resident[dev] == 0 || resident_lds[dev] != (int)lds_bytes) {
                int per_cu = 0, cus = 0;
                e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)kern, NT, lds_bytes);
                if (e != hipSuccess) return e;
                e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
                if (e != hipSuccess) return e;
                cap = (per_cu > 0 ? per_cu : 1) * (cus > 0 ? cus : 1);
                if (dev >= 0 && dev < 64) { resident[dev] = cap; resident_lds[dev] = (int)lds_bytes; }
            } else {
                cap = resident[dev];
            }
        }
        if (ntiles > cap) {
            grid = cap;
            b.queue = a.queue_mem;
            b.queue_base = grid * kTileDev;
            e = hipMemsetAsync(a.queue_mem, 0, sizeof(int), s);
            if (e != hipSuccess) return e;
            if (a.count == nullptr && a.order_list && a.order_score && a.order_hdr) {
                e = launch_order(P, a, s);
                if (e != hipSuccess) return e;
                b.list = a.order_list;
            }
        }
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds_bytes, s, P, b);
    return hipGetLastError();
}

hipError_t launch_admm(const StageDev &P, const BatchArgs &a, hipStream_t s, const char **why)
{
    const int ncell = P.nax * (P.N + 1);
    const int stride = P.nwaves * 64 / kTileDev;
    // cells per thread; the two-axis stage keeps the two cells of a step together
    const int needc = P.mixed ? 2 * ((P.N + 1 + stride - 1) / stride) : (ncell + stride - 1) / stride;
    int needi = 0;
    for (int w = 0; w < P.nwaves; w++) needi = P.wi_cnt[w] > needi ? P.wi_cnt[w] : needi;
    if (admm_lds_bytes(P) > kLdsLimit) { *why = "horizon too long: iterates do not fit the 160 KiB LDS"; return hipErrorInvalidValue; }
    // register-resident variants read 4 * KSREG B rows from the start of every block: they must stay
    // inside the zero-initialised part of LDS
    const int S = P.D + 1;
    const int rows_init = (P.separate_xt ? 2 + (P.kpart == 2 ? 1 : 0) : 1) * (P.n_int + S) + ncell * P.D + P.D +
                          (P.mixed ? (P.N + 1) * 6 + 6 : 0) + 8 + P.nax * P.D;
    // the variants with few cells per thread keep the box bounds in registers as (hi, -hi) for
    // every slot but the first
    bool sym = true;
    for (int a = 0; a < P.nax; a++)
        for (int k = 1; k <= P.D; k++) sym = sym && P.lb[a][k] == -P.ub[a][k];
    auto fits = [&](int ksreg) { return P.ks <= ksreg && (P.n_int - P.nb) + 4 * ksreg <= rows_init; };
    if (P.nwaves != 8) { *why = "only 8-wave workgroups are compiled"; return hipErrorInvalidValue; }
    const bool sep = P.separate_xt != 0;
    // the most general variant of each family keeps rhs and xt in one LDS vector; it also takes what
    // the faster variants decline
    StageDev single = P;
    single.separate_xt = 0;
    int needj = 0;
    for (int w = 0; w < P.nwaves; w++) needj = P.wj_cnt[w] > needj ? P.wj_cnt[w] : needj;
    // Development builds compile a subset of the variants (-DTG_VARIANTS=mask, bit = TG_HAS index).
    StageDev plain = P;          // variants without the s form / q in LDS
    plain.sm = 0; plain.ql = 0;
    if (admm_lds_bytes(plain) > kLdsLimit) { *why = "horizon too long: iterates do not fit the 160 KiB LDS"; return hipErrorInvalidValue; }
    single.sm = 0; single.ql = 0;
    // step sizes and boxes past the position slot equal on all axes (the usual case: one speed and
    // one acceleration limit): the UNI variants read them as scalars instead of per-lane values
    bool uni = true, uni0 = true;   // uni0: on the position slot only the last axis's step size and lower bound differ
    for (int ax = 1; ax < P.nax; ax++) {
        for (int k = 1; k <= P.D; k++)
            uni = uni && P.rho_box[ax][k] == P.rho_box[0][k] && P.lb[ax][k] == P.lb[0][k] && P.ub[ax][k] == P.ub[0][k];
        uni0 = uni0 && P.ub[ax][0] == P.ub[0][0] &&
               (ax == P.nax - 1 || (P.rho_box[ax][0] == P.rho_box[0][0] && P.lb[ax][0] == P.lb[0][0]));
    }
    if (P.lr && P.lr_np == 4) {
        // low-rank operators with register-resident fragments (6-DOF, short horizons)
        if (P.D == 2 && P.nax == 3 && !P.mixed && sym && sep && needc <= 2 && needi <= 1 && needj <= 1 && fits(16) &&
            P.lr_crb == 1 && P.lr_uk == 1) {
#if TG_HAS(1)
            if (uni) return launch_admm_t<512, 2, 3, false, 2, 1, true, 16, true, true, false, false, 2, false, kRingDepth, 1, 1>(plain, a, s, why);
            return launch_admm_t<512, 2, 3, false, 2, 1, true, 16, true, true>(plain, a, s, why);
#endif
        }
        *why = "internal: no register-resident low-rank variant for this stage";
        return hipErrorInvalidValue;
    }
    if (P.lr) {
        // block-row low-rank operators: rhs and xt share one vector
        if (sep) { *why = "internal: the low-rank form keeps rhs and xt in one vector"; return hipErrorInvalidValue; }
        const int items = needi > needj ? needi : needj;
        if (P.D == 2 && P.nax == 3 && !P.mixed && needc <= 5 && items <= 3) {
#if TG_HAS(2)
            // (step-major cells, SMJ with MAXC = 6, were measured at N = 50: 20350 against 20000 cycles
            // per iteration -- the per-axis constants then travel through spilled scalar registers)
            if (P.sm && P.ql && P.lr_ukp == 1) {
                if (uni && uni0) return launch_admm_t<512, 2, 3, false, 5, 3, true, 0, false, true, true, true, 1, false, kRingDepth, 1, 2>(P, a, s, why);
                if (uni) return launch_admm_t<512, 2, 3, false, 5, 3, true, 0, false, true, true, true, 1, false, kRingDepth, 1, 1>(P, a, s, why);
                return launch_admm_t<512, 2, 3, false, 5, 3, true, 0, false, true, true, true, 1>(P, a, s, why);
            }
#endif
#if TG_HAS(3)
            if (uni) return launch_admm_t<512, 2, 3, false, 5, 3, true, 0, false, true, false, false, 2, false, kRingDepth, 1, 1>(plain, a, s, why);
            return launch_admm_t<512, 2, 3, false, 5, 3, true, 0, false, true>(plain, a, s, why);
#endif
        }
#if TG_HAS(4)
        if (P.D == 3 && P.nax == 2 && P.mixed && needc <= 2 && items <= 2)
            return launch_admm_t<512, 3, 2, true, 2, 2, false, 0, false, true>(plain, a, s, why);
#endif
#if TG_HAS(5)
        if (P.D == 3 && P.nax == 2 && P.mixed && needc <= 4 && items <= 4)
            return launch_admm_t<512, 3, 2, true, 4, 4, false, 0, false, true>(plain, a, s, why);
#endif
#if TG_HAS(6)
        if (P.D == 3 && P.nax == 1 && !P.mixed && needc <= 2 && items <= 2)
            return launch_admm_t<512, 3, 1, false, 2, 2, false, 0, false, true>(plain, a, s, why);
#endif
        *why = "horizon too long for the compiled kernel variants";
        return hipErrorInvalidValue;
    }
    if (P.D == 2 && P.nax == 3 && !P.mixed) {
#if TG_HAS(7)
        if (sym && sep && needc <= 2 && needi <= 1 && fits(16)) return launch_admm_t<512, 2, 3, false, 2, 1, true, 16, true>(plain, a, s, why);
#endif
#if TG_HAS(8)
        if (sep && needc <= 5 && needi <= 3) return launch_admm_t<512, 2, 3, false, 5, 3, true, 0, true>(plain, a, s, why);
#endif
#if TG_HAS(9)
        if (needc <= 8 && needi <= 4) return launch_admm_t<512, 2, 3, false, 8, 4, true, 0, false>(single, a, s, why);
#endif
    } else if (P.D == 3 && P.nax == 2 && P.mixed) {
#if TG_HAS(10)
        // the ring depth that pads the k range least (host choice; N = 20: 21 k-steps = 3 rounds of 7)
        if (sep && needc <= 2 && needi <= 2) {
            if (P.ring == 7) return launch_admm_t<512, 3, 2, true, 2, 2, false, 0, true, false, false, false, 2, false, 7>(plain, a, s, why);
            if (P.ring == 6) return launch_admm_t<512, 3, 2, true, 2, 2, false, 0, true, false, false, false, 2, false, 6>(plain, a, s, why);
            return launch_admm_t<512, 3, 2, true, 2, 2, false, 0, true>(plain, a, s, why);
        }
#endif
#if TG_HAS(11)
        if (sep && needc <= 4 && needi <= 4) return launch_admm_t<512, 3, 2, true, 4, 4, false, 0, true>(plain, a, s, why);
#endif
#if TG_HAS(12)
        if (needc <= 4 && needi <= 4) return launch_admm_t<512, 3, 2, true, 4, 4, false, 0, false>(single, a, s, why);
#endif
    } else if (P.D == 3 && P.nax == 1 && !P.mixed) {
#if TG_HAS(13)
        if (sym && sep && needc <= 1 && needi <= 1 && fits(21)) {
            if (P.kpart == 2) return launch_admm_t<512, 3, 1, false, 1, 1, false, 21, true, false, false, false, 2, false, kRingDepth, 2>(plain, a, s, why);
            return launch_admm_t<512, 3, 1, false, 1, 1, false, 21, true>(plain, a, s, why);
        }
#endif
#if TG_HAS(14)
        if (sym && sep && needc <= 2 && needi <= 2) return launch_admm_t<512, 3, 1, false, 2, 2, false, 0, true>(plain, a, s, why);
#endif
#if TG_HAS(15)
        if (needc <= 4 && needi <= 4) return launch_admm_t<512, 3, 1, false, 4, 4, false, 0, false>(single, a, s, why);
#endif
    }
    *why = "horizon too long for the compiled kernel variants";
    return hipErrorInvalidValue;
}

hipError_t launch_assemble(const AsmDev &P, const BatchArgs &a, double *q, double *l, double *u,
                           hipStream_t s)
{
    const int rows = P.n > P.m ? P.n : P.m;
    dim3 grid((a.B + 255) / 256, (rows + P.rows_per_block - 1) / P.rows_per_block);
    hipLaunchKernelGGL(assemble_kernel, grid, dim3(256), 0, s, P, a, q, l, u);
    return hipGetLastError();
}

hipError_t launch_xy_bounds(int B, int ld, int N, double xy_max_vel, double xy_max_accel,
                            double z_max_accel, const double *X, double *xy_max, double *mixed_max,
                            hipStream_t s)
{
    dim3 grid((B + 255) / 256, N);
    hipLaunchKernelGGL(xy_bounds_kernel, grid, dim3(256), 0, s, B, ld, N, xy_max_vel, xy_max_accel,
                       z_max_accel, X, xy_max, mixed_max);
    return hipGetLastError();
}

hipError_t launch_yaw_ref(int B, int ld, int N, const double *ref, const double *X, double *yaw_ref,
                          hipStream_t s)
{
    dim3 grid((B + 255) / 256, N + 1);
    hipLaunchKernelGGL(yaw_ref_kernel, grid, dim3(256), 0, s, B, ld, N, ref, X, yaw_ref);
    return hipGetLastError();
}

hipError_t launch_merge12(int B, int ld, int N, const double *Xs, const double *Us, const int *status, double *X,
                          double *U, int *clear, hipStream_t s)
{
    const int rowsX = 12 * (N + 1), rowsU = 4 * N, rpb = 32;
    dim3 grid((B + 255) / 256, (rowsX + rowsU + rpb - 1) / rpb);
    hipLaunchKernelGGL(merge12_kernel, grid, dim3(256), 0, s, B, ld, rowsX, rowsU, rpb, Xs, Us, status, X, U, clear);
    return hipGetLastError();
}

hipError_t launch_plant_step(int nx, int nu, int N, const double *A, const double *Bm, int B, int ld, double *x0,
                             double *U, const int *status, int nstat, hipStream_t s)
{
    PlantDev P{};
    P.nx = nx; P.nu = nu;
    for (int i = 0; i < nx * nx; i++) P.A[i] = A[i];
    for (int i = 0; i < nx * nu; i++) P.Bm[i] = Bm[i];
    hipLaunchKernelGGL(plant_step_kernel, dim3((B + 255) / 256), dim3(256), 0, s, P, B, ld, N, x0, U, status, nstat);
    return hipGetLastError();
}

hipError_t launch_ref_shift(int B, int ld, int nx, int N, double *ref, hipStream_t s)
{
    hipLaunchKernelGGL(ref_shift_kernel, dim3((B + 255) / 256, nx), dim3(256), 0, s, B, ld, nx, N, ref);
    return hipGetLastError();
}

} // namespace tg
