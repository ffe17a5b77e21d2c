// Plain structs shared by the host code and the gfx950 kernels.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace tg {

constexpr int kTileDev = 16;            // instances per workgroup = MFMA f64 16x16x4 columns
constexpr double TG_DEV_INFTY = 1e30;   // OsqpEigen::INFTY
constexpr int TG_DEV_SOLVED = 1, TG_DEV_MAX_ITER = -2, TG_DEV_PRIMAL_INFEASIBLE = -3, TG_DEV_UNSOLVED = -10;
// OSQP's verdicts of the check with tenfold tolerances that it makes when max_iter is reached
constexpr int TG_DEV_SOLVED_INACCURATE = 2, TG_DEV_PRIMAL_INFEASIBLE_INACCURATE = 3;
constexpr double kInfBound = 1e26;      // OSQP_INFTY * MIN_SCALING: bounds beyond it count as infinite
constexpr double kDivisionTol = 1e-30;  // 1 / OSQP_INFTY
constexpr int kMaxAxes = 3, kMaxSlots = 4;
constexpr int kAdmmWaves = 8;           // waves of an ADMM workgroup
constexpr int kMaxItems = 8;                                        // MFMA work items per wave
constexpr int kRingDepth = 8;   // operator fragments (k steps) a wave keeps in flight from L2
constexpr size_t kStaticScratchBytes = (size_t)32 << 20;   // per-GPU static scratch of the entry point that may not allocate

// One QP stage as the ADMM kernel sees it.  Every axis of the model is a chain of
// integrators of order D (6-DOF: D = 2, states p,v, input a; 12-state stages: D = 3,
// states p,v,a, input j).  A "cell" is one horizon step of one axis: slots 0..D-1 are its
// states, slot D its input (absent at step N).  Internal variable index of
// (axis a, step i, slot k) = axis_base[a] + i*(D+1) + k.
struct StageDev {
    int N, D, nax, mixed;        // horizon, chain order, axes, XY mixed rows present
    int n_int, nb, nrb, ks;      // rows of all blocks, rows of one block (= block stride), 16-row blocks, k steps
    int ks_alloc;                // k steps stored per (operator, row block): ks + kRingDepth zero steps
    int kpart;                   // 2: register-resident dense variant with some row blocks split in two k-range items (wi_k0, wi_nr = k-steps, wi_part)
    int ring;                    // k steps a wave keeps in flight from L2 (streamed variants): kRingDepth, or less where that pads ks less
    int ngrp, grp_first[4], grp_cnt[4], grp_op[4]; // block groups sharing an operator
    int axis_base[kMaxAxes];
    int separate_xt;
    int sm;                      // box rows kept as s = argument of the last projection (see admm_kernel)
    int ql;                      // q of the position slots in LDS instead of registers
    int nwaves;                  // waves per workgroup the MFMA items are dealt to
    // MFMA work items (block group, row block) of every wave
    int wi_cnt[kAdmmWaves];
    unsigned char wi_grp[kAdmmWaves][kMaxItems], wi_rb[kAdmmWaves][kMaxItems];
    const double *Mfrag;         // [nops][nrb][ks_alloc][64] operator fragments, MFMA A-operand order
    // Block-row low-rank form (lr != 0): x_I = D_I r_I + U_I (C_I r) per 16-row tile I.  Mfrag then
    // holds the stacked C ([nops][lr_crb][ks_alloc][64]; wi_* items are (group, row block of C)),
    // DUfrag the tiles' D_I and U_I ([nops][nrb][4 + lr_uk][64]; wj_* items are (group, tile)).
    int lr, lr_q, lr_crb, lr_uk;
    int sb_rows;                 // LDS rows per block for C r: lr_crb * 16 + 8
    int lr_np;                   // partial sums C r is formed as: 1 (streamed form) or 4 (register-resident form)
    const double *DUfrag;
    int lr_ukp;                  // k-steps of U per tile as stored in the wave streams (>= lr_uk, zero padded)
    // Streamed variants: wave w reads the fragments of its items in the order of use from
    // Wfrag + wave_off[w] * 64: every first-stage item (wi_*) as wi_nr ring rounds of `ring`
    // k-steps starting at k-step wi_k0 (zero fragments past ks), then (low-rank form) 4 + lr_ukp
    // fragments per tile (wj_*), then zero fragments for the ring to run out into.  In the low-rank
    // form the k range of a row block of C is cut in two items whose results are partial sums
    // wi_part = 0 / 1 of C r.
    const double *Wfrag;
    int wave_off[kAdmmWaves];
    unsigned char wi_k0[kAdmmWaves][kMaxItems], wi_nr[kAdmmWaves][kMaxItems], wi_part[kAdmmWaves][kMaxItems];
    int wj_cnt[kAdmmWaves];
    unsigned char wj_grp[kAdmmWaves][kMaxItems], wj_rb[kAdmmWaves][kMaxItems];
    double dt, c2;               // c2 = dt^2/2 (position <- acceleration coupling, D = 3)
    double qw[kMaxAxes];         // q(position slot) = qw * ref            (qw = -state weight)
    double pdiag[kMaxAxes];      // P diagonal on the position slot
    double r, smooth;            // P on inputs: r*I + smooth * D'D (smooth = 0: off)
    double rho_eq;               // dynamics rows
    double rho_box[kMaxAxes][kMaxSlots], lb[kMaxAxes][kMaxSlots], ub[kMaxAxes][kMaxSlots];
    double rho_mx[6], ub_mx[6];  // XY mixed rows (symmetric bounds)
    // where the stage reads / writes the model-level arrays
    int x0_row[kMaxAxes];        // row of slot 0 of the axis in x0 (slots are consecutive rows)
    int ref_stride, ref_off[kMaxAxes]; // ref row of (step i, axis a) position = i*ref_stride + ref_off[a]
    int X_stride, X_off[kMaxAxes];     // X row of (step i, axis a, slot k) = i*X_stride + X_off[a] + k*X_slot
    int X_slot;                  // row distance between slots of one axis in X (6-DOF: 3, else 1)
    int x0_slot;                 // same for x0
    int U_stride, U_off[kMaxAxes];
    double sigma, alpha, eps_abs, eps_rel, eps_prim_inf;
    int max_iter, check, check_fine;
    // Step-size schedule: a solve is a sequence of launches, one per operator (phase).  This launch
    // takes its instances at iteration phase_start of their run (0: fresh problems; > 0: iterates
    // carried over from the previous phase) and hands those still running at the first check at or
    // after iteration phase_end to the next phase (BatchArgs::next_list); INT_MAX: the last phase.
    int phase_start, phase_end;
};

// reference-ordered tables for the assembly kernel
struct AsmDev {
    int n, m, rows_per_block;
    const double *q_w, *l_t, *u_t;
    const int32_t *q_ref_row, *x0_row, *ov_box, *ov_mixed;
};

// per-call arrays (device, structure-of-arrays, instance-minor)
struct BatchArgs {
    int B, ld, ref_ld, ov_ld, ldw;
    const double *x0, *ref, *ov_box, *ov_mixed;
    double *X, *U;
    int *status, *iters;
    const int *gate;         // optional: status of the upstream stage, instances != solved are skipped
    double *res;             // optional [2][res_ld]: primal / dual residual at exit
    int res_ld;
    int warm_in, warm_out;
    double *warm;            // [warm_rows][ldw] iterates kept between calls
    // Work queue for grids smaller than the batch: *queue counts the instances handed out beyond
    // the first queue_base (= 16 per workgroup launched); null when every tile has a workgroup.
    // launch_admm fills both in from queue_mem.
    int *queue;
    int queue_base;
    int *queue_mem;          // one int of device memory owned by the stage (may be null)
    int grid_cap;            // 0: as many workgroups as fit on the device at once; > 0: at most this many
    // Order in which a refilled grid takes the instances: list[ticket] (null: batch order).
    // launch_admm builds it in order_list when the batch goes through the queue (order_list,
    // order_score [B] and order_hdr [kOrderHdrBytes] are the stage's own device memory; may be null).
    const int *list;
    int *order_list;
    double *order_score;
    unsigned long long *order_hdr;
    // Phases of the step-size schedule.  count (device, may be null): number of entries of `list`
    // that make up this launch's batch, in place of B (the previous phase's hand-over count).
    // next_list / next_count (device, may be null): where this launch appends the instances it
    // hands over to the next phase; their iterates go to `warm` (x, z, y with y unscaled, so that
    // the rows mean the same under every operator).
    // zero_counts (device, may be null): two hand-over counters that no launch of this solve uses; the
    // first launch clears them for the next solve of the handle, which saves a memset per solve.
    const int *count;
    int *next_list;
    int *next_count;
    int *zero_counts;
    // done_list / done_count (device, may be null): a launch appends the instances it ends as
    // solved (the 12-state chain starts the next stage for the late finishers of this one from it).
    int *done_list;
    int *done_count;
};

// dynamic LDS a workgroup may ask for: the CU's 160 KiB less the kernel's static share (the block-wide vote helpers keep a few words there)
constexpr size_t kLdsLimit = (size_t)160 * 1024 - 512;
constexpr int kOrderBins = 256;
constexpr size_t kOrderHdrBytes = 8 + kOrderBins * 4 + 2 * 4 + 8;

} // namespace tg
