// C ABI of the batched MPC engine (include/tg_mpc.h): handles, device residency of the
// shared QP data, kernel sequencing for MPC::mpcLoop (mpc_6dof.cpp:616-649) and
// MPC12STATE::mpcLoop (mpc_12state.cpp:1374-1453).
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "stage_problem.hpp"

namespace tg {
hipError_t launch_admm(const StageDev &P, const BatchArgs &a, hipStream_t s, const char **why);
size_t admm_lds_bytes(const StageDev &P);
hipError_t launch_assemble(const AsmDev &P, const BatchArgs &a, double *q, double *l, double *u,
                           hipStream_t s);
hipError_t launch_xy_bounds(int B, int ld, int N, double xy_max_vel, double xy_max_accel,
                            double z_max_accel, const double *X, double *xy_max, double *mixed_max,
                            hipStream_t s);
hipError_t launch_plant_step(int nx, int nu, int N, const double *A, const double *Bm, int B, int ld, double *x0,
                             double *U, const int *status, int nstat, hipStream_t s);
hipError_t launch_ref_shift(int B, int ld, int nx, int N, double *ref, hipStream_t s);
hipError_t launch_yaw_ref(int B, int ld, int N, const double *ref, const double *X, double *yaw_ref,
                          hipStream_t s);
hipError_t launch_merge12(int B, int ld, int N, const double *Xs, const double *Us, const int *status, double *X,
                          double *U, int *clear, hipStream_t s);
hipError_t static_scratch_init(int dev);
void *static_scratch_ptr(int dev);
} // namespace tg

using namespace tg;

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t e__ = (expr);                                                          \
        if (e__ != hipSuccess)                                                            \
            return fail(TG_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));    \
    } while (0)

// ---------------------------------------------------------------- kernel timing
// Process-wide totals over every handle and device; g_time_mu guards them, so solves issued from
// several host threads may be timed together.  Events remember the device they were created on.
static std::mutex g_time_mu;
static bool g_timing = false;
static double g_admm_ms = 0.0;
static int g_admm_launches = 0;
struct PendingEvent { hipEvent_t a, b; int dev; };
static std::vector<PendingEvent> g_pending;

static void drain_events()   // caller holds g_time_mu
{
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    for (auto &p : g_pending) {
        float ms = 0.f;
        hipSetDevice(p.dev);
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            g_admm_ms += ms; g_admm_launches++;
        }
        hipEventDestroy(p.a); hipEventDestroy(p.b);
    }
    g_pending.clear();
    if (have_cur) hipSetDevice(cur);
}

// ---------------------------------------------------------------- device residency of one stage
template <class T> static int upload(const std::vector<T> &h, T **d)
{
    *d = nullptr;
    size_t bytes = std::max<size_t>(h.size(), 1) * sizeof(T);
    HIP_TRY(hipMalloc((void **)d, bytes));
    if (!h.empty()) HIP_TRY(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return TG_OK;
}

struct DevStage {
    StageProblem sp;
    StageDev dev{};
    AsmDev adev{};
    double *d_Mfrag = nullptr, *d_DUfrag = nullptr, *d_Wfrag = nullptr;
    bool regs_dense = false;  // short horizon: the dense fragments fit the registers
    int mfma_streamed = 0;    // MFMAs per tile-iteration of the streamed variants (zero-padding rounds included)
    int lowrank_mode = 0;     // 0: when it pays, 1: whenever the operators allow it, -1: never
    int *d_queue = nullptr;   // ticket counter of the kernel's instance queue
    int grid_cap = 0;
    int hard_first = 0;       // 0: a refilled grid takes the likely slow instances first, -1: batch order
    int *d_olist = nullptr; double *d_oscore = nullptr; unsigned long long *d_ohdr = nullptr; int order_cap = 0;
    double *d_qw = nullptr, *d_lt = nullptr, *d_ut = nullptr;
    int32_t *d_qrr = nullptr, *d_x0r = nullptr, *d_ovb = nullptr, *d_ovm = nullptr;
    // warm-start iterates of the instance slots
    double *d_warm = nullptr; int warm_cap = 0; bool warm_valid = false; int warm_B = 0;
    double *d_res = nullptr; int res_cap = 0, res_B = 0;

    // Step-size schedule (tg_solver_opts::phase_iters / phase_scale): the operators of the phases
    // after the first (each a stage of its own of which only the operator data is used), and the
    // iteration counts at which an instance still running moves on.  The lists of instances a phase
    // hands over (two, used in turn) and their counters are the stage's own device memory.
    std::vector<std::unique_ptr<DevStage>> later;
    int phase_iters[2] = {-1, -1};
    double phase_scale[2][4] = {{1, 1, 1, 1}, {1, 1, 1, 1}};
    int *d_plist[2] = {nullptr, nullptr}; int plist_cap = 0; int *d_pcount = nullptr; int pcount_set = 0;

    int init(const StageModel &mdl, const tg_solver_opts &o, double rho)
    {
        std::string err;
        if (!build_stage(mdl, rho, o.sigma, o.rho_scale, sp, err)) return fail(TG_E_NUMERIC, err);
        int rc;
        // later phases of the step-size schedule: the same stage with the class multipliers of the
        // phase (relative to this one's), built by a recursive call with the schedule switched off
        {
            int it[2]; double sc[2][4];
            default_phases(mdl, it, sc);
            for (int p = 0; p < 2; p++) {
                if (o.phase_iters[p] != 0) it[p] = o.phase_iters[p];
                for (int k = 0; k < 4; k++) if (o.phase_scale[p][k] > 0) sc[p][k] = o.phase_scale[p][k];
            }
            if (it[0] <= 0) it[1] = -1;                          // no second phase, no third
            if (it[1] > 0 && it[1] <= it[0]) return fail(TG_E_ARG, "phase_iters must increase");
            for (int p = 0; p < 2; p++) {
                phase_iters[p] = it[p] > 0 ? it[p] : -1;
                for (int k = 0; k < 4; k++) phase_scale[p][k] = sc[p][k];
                if (it[p] <= 0) continue;
                tg_solver_opts op = o;
                op.phase_iters[0] = op.phase_iters[1] = -1;
                for (int k = 0; k < 4; k++) op.rho_scale[k] = sp.rho_scale[k] * sc[p][k];
                later.emplace_back(new DevStage());
                if ((rc = later.back()->init(mdl, op, rho))) return rc;
            }
        }
        if ((rc = upload(sp.Mfrag, &d_Mfrag))) return rc;
        HIP_TRY(hipMalloc((void **)&d_queue, sizeof(int)));
        grid_cap = o.grid_cap;
        hard_first = o.hard_first;
        lowrank_mode = o.lowrank;
        if ((rc = upload(sp.q_w, &d_qw))) return rc;
        if ((rc = upload(sp.l_t, &d_lt))) return rc;
        if ((rc = upload(sp.u_t, &d_ut))) return rc;
        if ((rc = upload(sp.q_ref_row, &d_qrr))) return rc;
        if ((rc = upload(sp.asm_x0_row, &d_x0r))) return rc;
        if ((rc = upload(sp.asm_ov_box, &d_ovb))) return rc;
        if ((rc = upload(sp.asm_ov_mixed, &d_ovm))) return rc;
        const bool six = mdl.family == TG_FAM_6DOF, xy = mdl.family == TG_FAM_XY;
        dev.N = mdl.N; dev.D = sp.D; dev.nax = sp.nax; dev.mixed = mdl.n_mixed ? 1 : 0;
        dev.n_int = sp.n_int; dev.nb = sp.nb; dev.nrb = (sp.nb + 15) / 16; dev.ks = (sp.nb + 3) / 4; dev.ks_alloc = dev.ks + kRingDepth;
        dev.nwaves = 8;
        // groups of up to two adjacent blocks that share an operator
        dev.ngrp = 0;
        for (int b = 0; b < sp.nblk;) {
            int cnt = (b + 1 < sp.nblk && sp.op_of_block[b + 1] == sp.op_of_block[b]) ? 2 : 1;
            dev.grp_first[dev.ngrp] = b; dev.grp_cnt[dev.ngrp] = cnt; dev.grp_op[dev.ngrp] = sp.op_of_block[b];
            dev.ngrp++; b += cnt;
        }
        // (second attempt: the low-rank form was chosen or asked for but its C r vectors do not fit
        // the LDS beside the iterates -- the dense form is taken instead)
        for (int attempt = 0; attempt < 2; attempt++) {
            const int lowrank_mode = attempt == 0 ? this->lowrank_mode : -1;
            // Long horizons apply the operators in block-row low-rank form (fewer MFMAs and fragment
            // bytes per iteration); short ones keep the dense fragments in registers.
            {
                const int stride = dev.nwaves * 64 / kTileDev, ncell = sp.nax * (mdl.N + 1);
                const int cells = (ncell + stride - 1) / stride;
                const bool in_registers = (six && cells <= 2 && dev.ks <= 16) || (!six && !xy && cells <= 1 && dev.ks <= 21);
                regs_dense = in_registers;
                const int dense_mfma = dev.nrb * dev.ks, lr_mfma = sp.lr_crb * dev.ks + dev.nrb * (4 + sp.lr_uk);
                // (the streamed low-rank 6-DOF variant is compiled for five cells per thread: N <= 52;
                // longer horizons take the dense variant with eight)
                const bool lr_compiled = !six || cells <= 5;
                dev.lr = (sp.lr_q > 0 && lr_compiled && (lowrank_mode > 0 || (lowrank_mode == 0 && !in_registers && 5 * lr_mfma <= 3 * dense_mfma))) ? 1 : 0;
                dev.lr_q = sp.lr_q; dev.lr_crb = sp.lr_crb; dev.lr_uk = sp.lr_uk; dev.sb_rows = sp.lr_crb * 16 + 8;
                // streamed low-rank form: C r as two partial sums (the k range of a row block of C is
                // cut in two items, so that the first stage spreads evenly over the waves)
                dev.lr_np = 2;
                // short 6-DOF horizons: the low-rank form with register-resident fragments and C r as
                // four partial sums (one first-stage item per wave)
                // (its B-operand reads run up to 64 rows past the start of the last block: they must
                // stay inside the zero-initialised LDS, as launch_admm checks)
                const int S = sp.D + 1;
                const int rows_init = 2 * (sp.n_int + S) + ncell * sp.D + sp.D + 8 + sp.nax * sp.D;
                if (lowrank_mode >= 0 && six && in_registers && sp.lr_q == 4 && sp.lr_crb == 1 && sp.lr_uk == 1 &&
                    (sp.n_int - sp.nb) + 64 <= rows_init) { dev.lr = 1; dev.lr_np = 4; }
            }
            const bool streamed_lr = dev.lr && dev.lr_np == 2;
            // ring depth of the fragment streams: an item is a whole number of ring rounds, so the
            // depth that pads the k range least wins where a variant is compiled for it (dense XY)
            dev.ring = kRingDepth;
            if (xy && !dev.lr && mdl.N + 1 <= dev.nwaves * 64 / kTileDev && 2 * dev.nrb <= 2 * kAdmmWaves) {   // the two-cell, two-item variant
                int best = (dev.ks + kRingDepth - 1) / kRingDepth * kRingDepth;
                for (int r = kRingDepth - 1; r >= 6; r--) {
                    const int padded = (dev.ks + r - 1) / r * r;
                    if (padded < best) { best = padded; dev.ring = r; }
                }
            }
            if (streamed_lr) {   // the second partial sum needs LDS of its own: dropped where the horizon leaves none
                dev.separate_xt = 0; dev.sm = 0; dev.ql = 0;
                if (admm_lds_bytes(dev) > kLdsLimit) dev.lr_np = 1;
            }
            // a second on-chip vector for xt: the MFMA results are stored as soon as an item is done
            // (no accumulators held across a barrier, one barrier less per iteration); taken whenever
            // the iterates of the 16 instances still fit the 160 KiB of LDS
            dev.separate_xt = streamed_lr ? 0 : 1;   // the streamed low-rank form writes xt over the rhs, tile by tile
            dev.sm = 0; dev.ql = 0; dev.lr_ukp = 2;
            if (admm_lds_bytes(dev) > kLdsLimit) dev.separate_xt = 0;
            // 6-DOF: the boxes are the same for every instance and every call of the handle; when each
            // contains zero the kernel keeps a box row's (z, y) as the one number they are projected from.
            // Long horizons also move q of the position slots to LDS (five cells per thread: registers).
            if (six) {
                bool zero_inside = true;
                for (int a = 0; a < sp.nax; a++)
                    for (int k = 0; k <= sp.D; k++) zero_inside = zero_inside && sp.lb[a][k] <= 0.0 && sp.ub[a][k] >= 0.0;
                dev.sm = (zero_inside && streamed_lr) ? 1 : 0;   // (measured slower with register-resident fragments: 5650 against 5240 cycles per iteration at N = 20)
                if (dev.sm && streamed_lr) {
                    // the long-horizon variant is compiled with the s form, q in LDS and one k-step of
                    // U_I per tile (no input smoothing), or with none of them
                    const int sb_rows = dev.sb_rows;
                    dev.ql = 1;
                    if (dev.lr_uk == 1 && dev.lr_q == 4) { dev.lr_ukp = 1; dev.sb_rows = dev.lr_crb * 16; }   // no read past a tile's own four coefficients
                    if (dev.lr_ukp != 1 || admm_lds_bytes(dev) > kLdsLimit) { dev.ql = 0; dev.sm = 0; dev.lr_ukp = 2; dev.sb_rows = sb_rows; }
                }
            }
            // Single-axis stages with register-resident dense fragments have fewer row blocks than waves
            // (N = 20: six): as many row blocks as there are idle waves are split in two items of half the
            // k range each, so that every SIMD carries the same number of MFMAs; the halves' products
            // land in two vectors the projection adds.  (The conditions are those of launch_admm's
            // register-resident single-axis variant.)
            dev.kpart = 0;
            if (!six && !xy && !dev.lr && regs_dense && dev.separate_xt && dev.nrb < kAdmmWaves) {
                bool sym = true;
                for (int a = 0; a < sp.nax; a++)
                    for (int k = 1; k <= sp.D; k++) sym = sym && sp.lb[a][k] == -sp.ub[a][k];
                dev.kpart = 2;
                const int S = sp.D + 1, ncell = sp.nax * (mdl.N + 1);
                const int rows_init = 3 * (sp.n_int + S) + ncell * sp.D + sp.D + 8 + sp.nax * sp.D;
                if (!sym || (sp.n_int - sp.nb) + 4 * 21 > rows_init || admm_lds_bytes(dev) > kLdsLimit) dev.kpart = 0;
            }
            if (!(dev.lr && admm_lds_bytes(dev) > kLdsLimit)) break;
        }
        if (dev.lr && admm_lds_bytes(dev) > kLdsLimit) return fail(TG_E_ARG, "horizon too long: iterates do not fit the 160 KiB LDS");
        // deal the MFMA items: longest first onto the least loaded SIMD (waves w, w + 4, ... of a
        // workgroup share a SIMD), then onto its least loaded wave
        struct Item { int grp, rb, cost, k0, nr, part; };
        std::vector<Item> dealt1[kAdmmWaves], dealt2[kAdmmWaves];
        // A wave's items of one stage are all pairs of blocks that share an operator or all single
        // blocks (the kernel then runs one code path per wave and stage).  For every split of the
        // eight waves into pair waves and single waves the items go longest first onto the least
        // loaded wave of their kind; the waves are then matched to SIMDs heaviest with lightest
        // (waves w and w + 4 share SIMD w % 4); the split with the lightest heaviest SIMD wins.
        auto deal = [&](const std::vector<Item> &items, std::vector<Item> *out, int limit) -> bool {
            std::vector<Item> kind[2];
            for (const Item &it : items) kind[dev.grp_cnt[it.grp] == 2 ? 1 : 0].push_back(it);
            for (auto &v : kind) std::stable_sort(v.begin(), v.end(), [](const Item &x, const Item &y) { return x.cost > y.cost; });
            long best = -1;
            std::vector<Item> best_bins[kAdmmWaves];
            for (int npw = 0; npw <= kAdmmWaves; npw++) {
                if ((npw == 0) != kind[1].empty() || (npw == kAdmmWaves) != kind[0].empty()) {
                    if (!(kind[0].empty() && kind[1].empty())) continue;
                }
                std::vector<Item> bins[kAdmmWaves];
                int load[kAdmmWaves] = {0};
                bool ok = true;
                for (int k = 0; k < 2 && ok; k++) {
                    const int lo = k == 1 ? 0 : npw, hi = k == 1 ? npw : kAdmmWaves;
                    for (const Item &it : kind[k]) {
                        int w = lo;
                        for (int c = lo + 1; c < hi; c++) if (load[c] < load[w]) w = c;
                        if ((int)bins[w].size() >= limit) { ok = false; break; }
                        bins[w].push_back(it); load[w] += it.cost;
                    }
                }
                if (!ok) continue;
                int order[kAdmmWaves];
                std::iota(order, order + kAdmmWaves, 0);
                std::stable_sort(order, order + kAdmmWaves, [&](int x, int y) { return load[x] > load[y]; });
                long worst = 0, most = 0;
                for (int k = 0; k < 4; k++) worst = std::max<long>(worst, load[order[k]] + load[order[7 - k]]);
                for (int w = 0; w < kAdmmWaves; w++) most = std::max<long>(most, (long)bins[w].size());
                const long score = worst * 16 + most;
                if (best < 0 || score < best) {
                    best = score;
                    for (int k = 0; k < 4; k++) { best_bins[k] = bins[order[k]]; best_bins[k + 4] = bins[order[7 - k]]; }
                }
            }
            if (best < 0) return false;
            for (int w = 0; w < kAdmmWaves; w++) out[w] = best_bins[w];
            return true;
        };
        const int PD = dev.ring, rounds = (dev.ks + PD - 1) / PD;
        {
            // dense: (group, row block); low-rank: (group, row block of the stacked C[, part of the k
            // range]), then (group, tile)
            std::vector<Item> items, tiles;
            for (int gi = 0; gi < dev.ngrp; gi++) {
                const int cnt = dev.grp_cnt[gi];
                if (dev.lr && dev.lr_np == 4) {   // (group, quarter of the k range)
                    for (int pt = 0; pt < 4; pt++) items.push_back({gi, pt, cnt * 4, 0, 0, 0});
                } else if (dev.lr) {
                    const int ra = dev.lr_np == 2 ? (rounds + 1) / 2 : rounds, rb2 = rounds - ra;
                    for (int rb = 0; rb < dev.lr_crb; rb++) {
                        items.push_back({gi, rb, cnt * ra * PD, 0, ra, 0});
                        if (rb2 > 0) items.push_back({gi, rb, cnt * rb2 * PD, ra * PD, rb2, 1});
                    }
                } else if (dev.kpart == 2) {
                    // the last row blocks in halves of the k range (nr: k-steps of the item)
                    const int nsplit = std::min(dev.nrb, kAdmmWaves - dev.nrb), kh = (dev.ks + 1) / 2;
                    for (int rb = 0; rb < dev.nrb; rb++) {
                        if (rb < dev.nrb - nsplit) { items.push_back({gi, rb, cnt * dev.ks, 0, dev.ks, 0}); continue; }
                        items.push_back({gi, rb, cnt * kh, 0, kh, 0});
                        items.push_back({gi, rb, cnt * (dev.ks - kh), kh, dev.ks - kh, 1});
                    }
                } else {
                    for (int rb = 0; rb < dev.nrb; rb++) items.push_back({gi, rb, cnt * rounds * PD, 0, rounds, 0});
                }
                for (int rb = 0; rb < (dev.lr ? dev.nrb : 0); rb++) tiles.push_back({gi, rb, cnt * (4 + dev.lr_ukp), 0, 0, 0});
            }
            // no compiled kernel variant takes more than four items per wave and stage: a split within
            // that is preferred to a better balanced one beyond it
            if (!(deal(items, dealt1, 4) || deal(items, dealt1, kMaxItems)) || !(deal(tiles, dealt2, 4) || deal(tiles, dealt2, kMaxItems)))
                return fail(TG_E_ARG, "horizon too long: too many MFMA row blocks");
        }
        mfma_streamed = 0;
        for (int w = 0; w < kAdmmWaves; w++) {
            dev.wi_cnt[w] = (int)dealt1[w].size(); dev.wj_cnt[w] = (int)dealt2[w].size();
            for (size_t i = 0; i < dealt1[w].size(); i++) {
                const Item &it = dealt1[w][i];
                dev.wi_grp[w][i] = (unsigned char)it.grp; dev.wi_rb[w][i] = (unsigned char)it.rb;
                dev.wi_k0[w][i] = (unsigned char)it.k0; dev.wi_nr[w][i] = (unsigned char)it.nr; dev.wi_part[w][i] = (unsigned char)it.part;
                mfma_streamed += dev.grp_cnt[it.grp] * it.nr * PD;
            }
            for (size_t i = 0; i < dealt2[w].size(); i++) {
                const Item &it = dealt2[w][i];
                dev.wj_grp[w][i] = (unsigned char)it.grp; dev.wj_rb[w][i] = (unsigned char)it.rb;
                mfma_streamed += dev.grp_cnt[it.grp] * (4 + dev.lr_ukp);
            }
        }
        // fragment stream of every wave (streamed variants; the register-resident ones read Mfrag)
        std::vector<double> W;
        if (!(dev.lr && dev.lr_np == 4)) {
            const std::vector<double> &src = dev.lr ? sp.Cfrag : sp.Mfrag;
            const int rbs = dev.lr ? dev.lr_crb : dev.nrb;
            const size_t ksa = (size_t)dev.ks_alloc;
            const std::vector<double> zero(64, 0.0);
            for (int w = 0; w < kAdmmWaves; w++) {
                dev.wave_off[w] = (int)(W.size() / 64);
                for (const Item &it : dealt1[w])
                    for (int kk = it.k0; kk < it.k0 + it.nr * PD; kk++) {
                        const double *f = kk < dev.ks ? &src[(((size_t)dev.grp_op[it.grp] * rbs + it.rb) * ksa + kk) * 64] : zero.data();
                        W.insert(W.end(), f, f + 64);
                    }
                for (const Item &it : dealt2[w])
                    for (int kk = 0; kk < 4 + dev.lr_ukp; kk++) {
                        const double *f = kk < 4 + dev.lr_uk
                            ? &sp.DUfrag[(((size_t)dev.grp_op[it.grp] * dev.nrb + it.rb) * (4 + dev.lr_uk) + kk) * 64] : zero.data();
                        W.insert(W.end(), f, f + 64);
                    }
                for (int kk = 0; kk < kMaxItems * 6 + PD; kk++) W.insert(W.end(), zero.begin(), zero.end());   // the ring runs out into zeros
            }
            if ((rc = upload(W, &d_Wfrag))) return rc;
        }
        dev.Wfrag = d_Wfrag;
        if (dev.lr && dev.lr_np == 4) {
            hipFree(d_Mfrag); d_Mfrag = nullptr;
            if ((rc = upload(sp.Cfrag, &d_Mfrag))) return rc;
            if ((rc = upload(sp.DUfrag, &d_DUfrag))) return rc;
        }
        dev.DUfrag = d_DUfrag;
        dev.Mfrag = d_Mfrag;
        dev.dt = mdl.Ad[1] != 0.0 ? mdl.Ad[1] : mdl.Ad[3];   // Ad(0,1) (triple integrator) or Ad(0,3) (6-DOF)
        dev.c2 = six ? 0.0 : mdl.Ad[2];
        dev.r = mdl.r; dev.smooth = mdl.smoothing ? mdl.smooth_w : 0.0;
        dev.rho_eq = sp.rho_eq;
        dev.X_slot = six ? 3 : 1; dev.x0_slot = six ? 3 : 1;
        dev.ref_stride = mdl.ref_stride; dev.X_stride = mdl.X_stride; dev.U_stride = mdl.U_stride;
        for (int a = 0; a < sp.nax; a++) {
            const int pc = six ? a : (xy ? 3 * a : 0);  // position component of the axis
            dev.axis_base[a] = sp.axis_base[a];
            dev.qw[a] = -1.0 * mdl.qdiag[pc]; dev.pdiag[a] = mdl.qdiag[pc];
            dev.x0_row[a] = mdl.x0_row[pc];
            dev.ref_off[a] = mdl.ref_off + (mdl.ref_pos_only ? 0 : pc);
            dev.X_off[a] = mdl.X_off + pc; dev.U_off[a] = mdl.U_off + a;
            for (int k = 0; k <= sp.D; k++) {
                dev.rho_box[a][k] = sp.rho_box[a][k]; dev.lb[a][k] = sp.lb[a][k]; dev.ub[a][k] = sp.ub[a][k];
            }
        }
        for (int k = 0; k < 6; k++) { dev.rho_mx[k] = sp.rho_mx[k]; dev.ub_mx[k] = sp.ub_mx[k]; }
        dev.sigma = o.sigma; dev.alpha = o.alpha; dev.eps_abs = o.eps_abs; dev.eps_rel = o.eps_rel; dev.eps_prim_inf = o.eps_prim_inf;
        dev.max_iter = o.max_iter; dev.check = o.check_termination; dev.check_fine = o.check_fine;
        adev.n = sp.n; adev.m = sp.m; adev.rows_per_block = 32;
        adev.q_w = d_qw; adev.l_t = d_lt; adev.u_t = d_ut; adev.q_ref_row = d_qrr;
        adev.x0_row = d_x0r; adev.ov_box = d_ovb; adev.ov_mixed = d_ovm;
        return TG_OK;
    }
    void release()
    {
        for (auto &l : later) l->release();
        later.clear();
        hipFree(d_plist[0]); hipFree(d_plist[1]); hipFree(d_pcount);
        hipFree(d_Mfrag); hipFree(d_DUfrag); hipFree(d_Wfrag); hipFree(d_queue);
        hipFree(d_qw); hipFree(d_lt); hipFree(d_ut); hipFree(d_qrr); hipFree(d_x0r); hipFree(d_ovb);
        hipFree(d_ovm); hipFree(d_warm); hipFree(d_res);
        hipFree(d_olist); hipFree(d_oscore); hipFree(d_ohdr);
    }
    void kernel_info(int *mfma, int *lowrank) const
    {
        int blocks = 0;
        for (int gi = 0; gi < dev.ngrp; gi++) blocks += dev.grp_cnt[gi];
        int n = mfma_streamed;                                             // streamed: zero-padded ring rounds included
        if (dev.lr && dev.lr_np == 4) n = blocks * (16 + dev.nrb * (4 + dev.lr_uk));   // register-resident low-rank form
        else if (!dev.lr && regs_dense) n = blocks * dev.nrb * dev.ks;     // register-resident dense fragments
        if (mfma) *mfma = n;
        if (lowrank) *lowrank = dev.lr;
    }
    size_t warm_rows() const
    {
        return (size_t)3 * sp.n_int + (size_t)2 * sp.nax * (sp.mdl.N + 1) * sp.D + (size_t)12 * (sp.mdl.N + 1);
    }
    // The kernel writes a slot's warm rows only when the slot's instance ran (a gated instance, or
    // one beyond a smaller batch, leaves them alone), so a fresh array starts as all zeros: zero
    // iterates are a cold start.
    int ensure_warm(int B, hipStream_t s)
    {
        int cap = (B + 15) / 16 * 16;
        if (cap <= warm_cap) return TG_OK;
        hipFree(d_warm); d_warm = nullptr; warm_cap = 0; warm_valid = false;
        size_t rows = warm_rows();
        HIP_TRY(hipMalloc((void **)&d_warm, rows * cap * sizeof(double)));
        warm_cap = cap;
        HIP_TRY(hipMemsetAsync(d_warm, 0, rows * cap * sizeof(double), s));
        return TG_OK;
    }
    // work arrays of the instance ordering (launch_admm uses them when the batch goes through the queue)
    int ensure_order(int B)
    {
        if (B <= order_cap) return TG_OK;
        hipFree(d_olist); hipFree(d_oscore); d_olist = nullptr; d_oscore = nullptr; order_cap = 0;
        const int cap = (B + 255) / 256 * 256;
        HIP_TRY(hipMalloc((void **)&d_olist, (size_t)cap * sizeof(int)));
        HIP_TRY(hipMalloc((void **)&d_oscore, (size_t)cap * sizeof(double)));
        if (!d_ohdr) HIP_TRY(hipMalloc((void **)&d_ohdr, kOrderHdrBytes));
        order_cap = cap;
        return TG_OK;
    }
    // hand-over lists of the step-size schedule: two lists of instance numbers and a counter per phase boundary
    int ensure_plist(int B)
    {
        // two sets of two counters: a solve counts in one set and its first launch clears the other
        // for the solve after it (stream order makes that safe; solves of one handle follow each other)
        if (!d_pcount) {
            HIP_TRY(hipMalloc((void **)&d_pcount, 4 * sizeof(int)));
            HIP_TRY(hipMemset(d_pcount, 0, 4 * sizeof(int)));
            pcount_set = 0;
        }
        if (B <= plist_cap) return TG_OK;
        hipFree(d_plist[0]); hipFree(d_plist[1]); d_plist[0] = d_plist[1] = nullptr; plist_cap = 0;
        const int cap = (B + 255) / 256 * 256;
        HIP_TRY(hipMalloc((void **)&d_plist[0], (size_t)cap * sizeof(int)));
        HIP_TRY(hipMalloc((void **)&d_plist[1], (size_t)cap * sizeof(int)));
        plist_cap = cap;
        return TG_OK;
    }
    // primal / dual residual of every instance at its exit, [2][res_cap]
    int ensure_res(int ld)
    {
        if (ld <= res_cap) return TG_OK;
        hipFree(d_res); d_res = nullptr; res_cap = 0; res_B = 0;
        HIP_TRY(hipMalloc((void **)&d_res, (size_t)2 * ld * sizeof(double)));
        res_cap = ld;
        return TG_OK;
    }
    // What the 12-state chain adds to a stage's solve when it overlaps two stages.
    struct SolveExtra {
        const int *first_list = nullptr, *first_count = nullptr;   // the first launch takes this list instead of the batch
        int *done_list = nullptr, *done_count = nullptr;           // launches after the first append what they solve
        int warm_in = -1;                                          // >= 0: whether warm rows are read, decided by the caller
        std::function<int()> after_first;                          // called when the first launch has been issued
    };
    // Phases a solve will run and the checks they end at (the step-size schedule: phase p ends at the
    // first check at or after phase_iters[p]; an end at or beyond max_iter never comes; two boundaries
    // that fall on the same check leave the operator between them unused).
    int plan_phases(int bound[2], int oper[3]) const
    {
        int nphase = 1;
        bound[0] = bound[1] = 0; oper[0] = oper[1] = oper[2] = 0;
        for (int p = 0; p < (int)later.size(); p++) {
            const int b = (phase_iters[p] + dev.check - 1) / dev.check * dev.check;
            if (b >= dev.max_iter) break;
            if (nphase > 1 && b <= bound[nphase - 2]) { oper[nphase - 1] = p + 1; continue; }
            bound[nphase - 1] = b; oper[nphase] = p + 1; nphase++;
        }
        return nphase;
    }
    bool warm_rows_valid(int B) const { return warm_valid && warm_B >= B; }
    // One batched ADMM solve of this stage.  no_alloc: the caller guarantees that nothing here may
    // allocate (no warm start, no residual record, no timing events; a.queue_mem is the caller's).
    int solve(BatchArgs a, int warm, hipStream_t s, bool no_alloc = false, const SolveExtra *ex = nullptr)
    {
        a.warm_in = 0; a.warm_out = 0; a.ldw = 0;
        a.warm = nullptr;
        if (warm) {
            if (no_alloc) return fail(TG_E_ARG, "warm start is not available without allocation");
            int rc = ensure_warm(a.B, s);
            if (rc) return rc;
            a.ldw = warm_cap;
            a.warm = d_warm;
            a.warm_in = (ex && ex->warm_in >= 0) ? ex->warm_in : (warm_rows_valid(a.B) ? 1 : 0);
            a.warm_out = 1;
        }
        a.res = nullptr; a.res_ld = 0;
        if (!no_alloc) {
            int rc = ensure_res(a.ld);
            if (rc) return rc;
            a.res = d_res; a.res_ld = res_cap;
            a.queue_mem = grid_cap >= 0 ? d_queue : nullptr;
        }
        a.list = nullptr; a.order_list = nullptr; a.order_score = nullptr; a.order_hdr = nullptr;
        a.grid_cap = grid_cap;
        a.count = nullptr; a.next_list = nullptr; a.next_count = nullptr; a.zero_counts = nullptr;
        a.done_list = nullptr; a.done_count = nullptr;
        // The step-size schedule.  Without allocation there is one phase only.
        int bound[2] = {0, 0}, oper[3] = {0, 0, 0}, nphase = 1;
        if (!no_alloc) nphase = plan_phases(bound, oper);
        // Likely slow instances first (hard_first): 1 always, 0 where it pays.  With a schedule the slow
        // instances leave the first launch after phase_iters[0] iterations wherever they started, and
        // ordering a large batch costs more than it saves (N = 50, 262144 instances: 14.5 against
        // 14.2 ms); a batch of a few rounds of the grid still gains (32768: 2.65 against 2.75 ms).
        // Short horizons never gain: their tails are shorter than the three ordering kernels take
        // (N = 20, 8192 vehicles in closed loop: 0.25 against 0.21 ms per step).
        const bool pays = sp.mdl.N > 30 && (nphase == 1 || a.B <= 48 * 1024);
        if (!no_alloc && grid_cap >= 0 && (hard_first > 0 || (hard_first == 0 && pays)) && a.B > 16 * 256) {   // smaller batches never go through the queue
            int rc = ensure_order(a.B);
            if (rc) return rc;
            a.order_list = d_olist; a.order_score = d_oscore; a.order_hdr = d_ohdr;
        }
        if (nphase > 1) {
            int rc = ensure_warm(a.B, s);         // the hand-over travels through the warm rows
            if (rc) return rc;
            if ((rc = ensure_plist(a.B))) return rc;
            a.ldw = warm_cap; a.warm = d_warm;
        }
        const char *why = "";
        hipEvent_t e0 = nullptr, e1 = nullptr;
        bool timing = false;
        if (!no_alloc) {
            std::lock_guard<std::mutex> lk(g_time_mu);
            timing = g_timing;
        }
        if (timing) {
            if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess || hipEventRecord(e0, s) != hipSuccess) {
                if (e0) hipEventDestroy(e0);
                if (e1) hipEventDestroy(e1);
                return fail(TG_E_HIP, "timing events could not be created");
            }
        }
        // the chain of launches: every operator of the schedule in turn, each taking what the one
        // before handed over
        auto chain = [&](const BatchArgs &first, int *const plist[2], int *pcount, hipStream_t st) -> int {
            for (int p = 0; p < nphase; p++) {
                StageDev d = oper[p] == 0 ? dev : later[oper[p] - 1]->dev;
                if (p > 0) {   // the run-time settings are those of the stage
                    d.alpha = dev.alpha; d.eps_abs = dev.eps_abs; d.eps_rel = dev.eps_rel; d.eps_prim_inf = dev.eps_prim_inf;
                    d.max_iter = dev.max_iter; d.check = dev.check; d.check_fine = dev.check_fine;
                }
                d.phase_start = p > 0 ? bound[p - 1] : 0;
                d.phase_end = p + 1 < nphase ? bound[p] : INT_MAX;
                BatchArgs ap = first;
                if (p > 0) {
                    ap.count = pcount + (p - 1); ap.list = plist[(p - 1) & 1]; ap.warm_in = 1;
                    ap.order_list = nullptr; ap.order_score = nullptr; ap.order_hdr = nullptr;
                }
                if (p + 1 < nphase) { ap.next_list = plist[p & 1]; ap.next_count = pcount + p; }
                ap.zero_counts = (p == 0 && nphase > 1) ? d_pcount + 2 * (pcount_set ^ 1) : nullptr;
                if (ex) {
                    if (p == 0 && ex->first_count) { ap.count = ex->first_count; ap.list = ex->first_list; }
                    if (p > 0) { ap.done_list = ex->done_list; ap.done_count = ex->done_count; }
                }
                hipError_t e = launch_admm(d, ap, st, &why);
                if (e != hipSuccess)
                    return fail(e == hipErrorInvalidValue ? TG_E_ARG : TG_E_HIP,
                                std::string("admm launch: ") + (why[0] ? why : hipGetErrorString(e)));
                if (p == 0 && ex && ex->after_first) {
                    const int rc = ex->after_first();
                    if (rc != TG_OK) return rc;
                }
            }
            return TG_OK;
        };
        const int crc = chain(a, d_plist, d_pcount + 2 * pcount_set, s);
        if (crc != TG_OK) {
            if (timing) { hipEventDestroy(e0); hipEventDestroy(e1); }
            if (nphase > 1) {       // part of the chain may have run: both counter sets start over
                hipStreamSynchronize(s);
                hipMemset(d_pcount, 0, 4 * sizeof(int));
            }
            return crc;
        }
        if (nphase > 1) pcount_set ^= 1;
        if (timing) {
            if (hipEventRecord(e1, s) != hipSuccess) {
                hipEventDestroy(e0); hipEventDestroy(e1);
                return fail(TG_E_HIP, "timing event could not be recorded");
            }
            int dv = 0;
            hipGetDevice(&dv);
            std::lock_guard<std::mutex> lk(g_time_mu);
            g_pending.push_back({e0, e1, dv});
        }
        if (warm) { warm_valid = true; warm_B = a.B; }
        res_B = no_alloc ? 0 : a.B;   // a solve without a residual record leaves none of an earlier one behind
        return TG_OK;
    }
    // residuals of the last solve, host arrays of B doubles each (either may be null)
    int residuals(int B, double *prim, double *dual) const
    {
        if (B < 0 || B > res_B) return fail(TG_E_STATE, "no residuals recorded for that many instances");
        if (B == 0) return TG_OK;
        HIP_TRY(hipDeviceSynchronize());
        if (prim) HIP_TRY(hipMemcpy(prim, d_res, sizeof(double) * B, hipMemcpyDeviceToHost));
        if (dual) HIP_TRY(hipMemcpy(dual, d_res + res_cap, sizeof(double) * B, hipMemcpyDeviceToHost));
        return TG_OK;
    }
    // effective options: automatic choices resolved
    void effective_opts(const tg_solver_opts &asked, tg_solver_opts *out) const
    {
        *out = asked;
        out->rho = sp.rho; out->sigma = dev.sigma; out->alpha = dev.alpha;
        out->eps_abs = dev.eps_abs; out->eps_rel = dev.eps_rel; out->eps_prim_inf = dev.eps_prim_inf;
        out->max_iter = dev.max_iter; out->check_termination = dev.check; out->check_fine = dev.check_fine;
        out->grid_cap = grid_cap; out->hard_first = hard_first; out->lowrank = dev.lr;
        for (int k = 0; k < 4; k++) out->rho_scale[k] = sp.rho_scale[k];
        for (int p = 0; p < 2; p++) {
            out->phase_iters[p] = phase_iters[p];
            for (int k = 0; k < 4; k++) out->phase_scale[p][k] = phase_scale[p][k];
        }
    }
    // options that do not enter the factorised operator can change on a live handle
    void set_run_opts(const tg_solver_opts &o)
    {
        dev.alpha = o.alpha; dev.eps_abs = o.eps_abs; dev.eps_rel = o.eps_rel; dev.eps_prim_inf = o.eps_prim_inf;
        dev.max_iter = o.max_iter; dev.check = o.check_termination; dev.check_fine = o.check_fine;
        grid_cap = o.grid_cap;
        hard_first = o.hard_first;
    }
    int get(int which, double *out) const
    {
        const int n = sp.n, m = sp.m, nx = sp.mdl.nx, nu = sp.mdl.nu;
        auto copy = [&](const std::vector<double> &v) { std::memcpy(out, v.data(), v.size() * sizeof(double)); };
        switch (which) {
        case TG_MAT_P: copy(sp.P); break;
        case TG_MAT_A: copy(sp.A); break;
        case TG_MAT_KINV: copy(sp.Kinv); break;
        case TG_VEC_L: copy(sp.l_t); break;
        case TG_VEC_U: copy(sp.u_t); break;
        case TG_VEC_RHO: copy(sp.rho_vec); break;
        case TG_MAT_AD: std::memcpy(out, sp.mdl.Ad, sizeof(double) * nx * nx); break;
        case TG_MAT_BD: std::memcpy(out, sp.mdl.Bd, sizeof(double) * nx * nu); break;
        case TG_MAT_Q:
            std::memset(out, 0, sizeof(double) * nx * nx);
            for (int a = 0; a < nx; a++) out[a * nx + a] = sp.mdl.qdiag[a];
            break;
        case TG_MAT_R:
            std::memset(out, 0, sizeof(double) * nu * nu);
            for (int a = 0; a < nu; a++) out[a * nu + a] = sp.mdl.r;
            break;
        default: return fail(TG_E_ARG, "unknown matrix id");
        }
        (void)n; (void)m;
        return TG_OK;
    }
};

// growable device scratch for the host-pointer entry points
struct HostStaging {
    double *d = nullptr; size_t cap = 0;
    int *di = nullptr; size_t capi = 0;
    int ensure(size_t doubles, size_t ints)
    {
        if (doubles > cap) { hipFree(d); d = nullptr; cap = 0; HIP_TRY(hipMalloc((void **)&d, doubles * sizeof(double))); cap = doubles; }
        if (ints > capi) { hipFree(di); di = nullptr; capi = 0; HIP_TRY(hipMalloc((void **)&di, ints * sizeof(int))); capi = ints; }
        return TG_OK;
    }
    void release() { hipFree(d); hipFree(di); }
};

static void to_soa(const double *aos, int B, int rows, int ld, std::vector<double> &soa)
{
    soa.assign((size_t)rows * ld, 0.0);
    for (int b = 0; b < B; b++)
        for (int r = 0; r < rows; r++) soa[(size_t)r * ld + b] = aos[(size_t)b * rows + r];
}
static void from_soa(const std::vector<double> &soa, int B, int rows, int ld, double *aos)
{
    for (int b = 0; b < B; b++)
        for (int r = 0; r < rows; r++) aos[(size_t)b * rows + r] = soa[(size_t)r * ld + b];
}

static int check_opts(const tg_solver_opts &o)
{
    if (!(o.sigma > 0) || !(o.alpha > 0 && o.alpha < 2) || !(o.eps_abs >= 0) || !(o.eps_rel >= 0) || !(o.eps_prim_inf >= 0) ||
        o.max_iter < 1 || o.check_termination < 1 || o.check_fine < 0)
        return fail(TG_E_ARG, "invalid solver options");
    for (int k = 0; k < 4; k++)
        if (!(o.rho_scale[k] >= 0)) return fail(TG_E_ARG, "rho_scale entries must be >= 0");
    for (int p = 0; p < 2; p++)
        for (int k = 0; k < 4; k++)
            if (!(o.phase_scale[p][k] >= 0)) return fail(TG_E_ARG, "phase_scale entries must be >= 0");
    return TG_OK;
}

extern "C" {

void tg_default_opts(tg_solver_opts *o)
{
    o->rho = 0.0; o->sigma = 1e-6; o->alpha = 1.6; o->eps_abs = 1e-3; o->eps_rel = 1e-3;
    o->max_iter = 4000; o->check_termination = 25; o->eps_prim_inf = 1e-4; o->check_fine = 0; o->grid_cap = 0; o->lowrank = 0; o->hard_first = 0;
    for (int k = 0; k < 4; k++) o->rho_scale[k] = 0.0;
    for (int p = 0; p < 2; p++) {
        o->phase_iters[p] = 0;
        for (int k = 0; k < 4; k++) o->phase_scale[p][k] = 0.0;
    }
    o->overlap = 0;
}
const char *tg_last_error(void) { return g_err.c_str(); }
int tg_device_count(void) { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }

int tg_copy_to_host(void *dst, const void *d_src, unsigned long long bytes)
{
    if (!dst || !d_src) return fail(TG_E_ARG, "null pointer");
    HIP_TRY(hipMemcpy(dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost));
    return TG_OK;
}

void tg_timing_enable(int on)
{
    std::lock_guard<std::mutex> lk(g_time_mu);
    g_timing = on != 0;
}
int tg_timing_read(double *ms, int *launches, int reset)
{
    std::lock_guard<std::mutex> lk(g_time_mu);
    drain_events();
    if (ms) *ms = g_admm_ms;
    if (launches) *launches = g_admm_launches;
    if (reset) { g_admm_ms = 0; g_admm_launches = 0; }
    return TG_OK;
}

// ================================================================ 6-DOF
struct tg_mpc6 {
    tg_mpc6_params p; tg_solver_opts o; int device;
    DevStage st; HostStaging hs;
};

int tg_mpc6_create(const tg_mpc6_params *p, const tg_solver_opts *o, int device, tg_mpc6 **out)
{
    if (!p || !out) return fail(TG_E_ARG, "null argument");
    *out = nullptr;
    if (p->N < 1 || !(p->dt > 0)) return fail(TG_E_ARG, "N must be >= 1 and dt > 0");
    if (p->state_weight < 0 || p->input_weight < 0 || p->smooth_input_weight < 0)
        return fail(TG_E_ARG, "weights must be >= 0");
    for (int a = 0; a < 3; a++)
        if (p->max_vel[a] < 0 || p->max_accel[a] < 0) return fail(TG_E_ARG, "limits must be >= 0");
    tg_solver_opts oo;
    if (o) oo = *o; else tg_default_opts(&oo);
    int rc = check_opts(oo);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(device));
    tg_mpc6 *h = new tg_mpc6();
    h->p = *p; h->o = oo; h->device = device;
    rc = h->st.init(model_6dof(*p), oo, oo.rho);
    if (rc) { h->st.release(); delete h; return rc; }
    *out = h;
    return TG_OK;
}

void tg_mpc6_destroy(tg_mpc6 *h)
{
    if (!h) return;
    hipSetDevice(h->device);
    h->st.release(); h->hs.release();
    delete h;
}

int tg_mpc6_solve_batch(tg_mpc6 *h, int B, int ld, const double *d_x0, const double *d_ref,
                        double *d_X, double *d_U, int *d_status, int *d_iters, int warm, void *stream)
{
    if (!h) return fail(TG_E_ARG, "null handle");
    if (B < 0 || ld < B) return fail(TG_E_ARG, "need 0 <= B <= ld");
    if (B == 0) return TG_OK;
    if (!d_x0 || !d_ref || !d_X || !d_U) return fail(TG_E_ARG, "null device array");
    HIP_TRY(hipSetDevice(h->device));
    BatchArgs a{};
    a.B = B; a.ld = ld; a.ref_ld = ld; a.ov_ld = 0;
    a.x0 = d_x0; a.ref = d_ref; a.ov_box = nullptr; a.ov_mixed = nullptr;
    a.X = d_X; a.U = d_U; a.status = d_status; a.iters = d_iters;
    return h->st.solve(a, warm, (hipStream_t)stream);
}

int tg_mpc6_solve_host(tg_mpc6 *h, int B, const double *x0, const double *ref, double *X, double *U,
                       int *status, int *iters, int warm)
{
    if (!h) return fail(TG_E_ARG, "null handle");
    if (B < 0) return fail(TG_E_ARG, "B < 0");
    if (B == 0) return TG_OK;
    if (!x0 || !ref || !X || !U) return fail(TG_E_ARG, "null host array");
    HIP_TRY(hipSetDevice(h->device));
    const int N = h->p.N, ld = (B + 15) / 16 * 16;
    const int rx = 6, rr = 6 * (N + 1), rX = 6 * (N + 1), rU = 3 * N;
    int rc = h->hs.ensure((size_t)(rx + rr + rX + rU) * ld, (size_t)2 * ld);
    if (rc) return rc;
    double *dx0 = h->hs.d, *dref = dx0 + (size_t)rx * ld, *dX = dref + (size_t)rr * ld, *dU = dX + (size_t)rX * ld;
    std::vector<double> tmp;
    to_soa(x0, B, rx, ld, tmp);
    HIP_TRY(hipMemcpy(dx0, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice));
    to_soa(ref, B, rr, ld, tmp);
    HIP_TRY(hipMemcpy(dref, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice));
    // an instance that does not reach "solved" gets no X / U from the kernel: zeros, not what an
    // earlier call left in the staging buffer
    HIP_TRY(hipMemset(dX, 0, (size_t)(rX + rU) * ld * sizeof(double)));
    rc = tg_mpc6_solve_batch(h, B, ld, dx0, dref, dX, dU, h->hs.di, h->hs.di + ld, warm, nullptr);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    tmp.resize((size_t)rX * ld);
    HIP_TRY(hipMemcpy(tmp.data(), dX, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
    from_soa(tmp, B, rX, ld, X);
    tmp.resize((size_t)rU * ld);
    HIP_TRY(hipMemcpy(tmp.data(), dU, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
    from_soa(tmp, B, rU, ld, U);
    if (status) HIP_TRY(hipMemcpy(status, h->hs.di, sizeof(int) * B, hipMemcpyDeviceToHost));
    if (iters) HIP_TRY(hipMemcpy(iters, h->hs.di + ld, sizeof(int) * B, hipMemcpyDeviceToHost));
    return TG_OK;
}

int tg_mpc6_assemble_batch(tg_mpc6 *h, int B, int ld, const double *d_x0, const double *d_ref,
                           double *d_q, double *d_l, double *d_u, void *stream)
{
    if (!h) return fail(TG_E_ARG, "null handle");
    if (B < 0 || ld < B) return fail(TG_E_ARG, "need 0 <= B <= ld");
    if (B == 0) return TG_OK;
    if (!d_x0 || !d_ref) return fail(TG_E_ARG, "null device array");
    if ((d_l == nullptr) != (d_u == nullptr)) return fail(TG_E_ARG, "l and u go together");
    HIP_TRY(hipSetDevice(h->device));
    BatchArgs a{};
    a.B = B; a.ld = ld; a.ref_ld = ld; a.x0 = d_x0; a.ref = d_ref;
    HIP_TRY(launch_assemble(h->st.adev, a, d_q, d_l, d_u, (hipStream_t)stream));
    return TG_OK;
}

int tg_mpc6_assemble_host(tg_mpc6 *h, int B, const double *x0, const double *ref, double *q, double *l, double *u)
{
    if (!h) return fail(TG_E_ARG, "null handle");
    if (B < 0) return fail(TG_E_ARG, "B < 0");
    if (B == 0) return TG_OK;
    if (!x0 || !ref) return fail(TG_E_ARG, "null host array");
    if ((l == nullptr) != (u == nullptr)) return fail(TG_E_ARG, "l and u go together");
    HIP_TRY(hipSetDevice(h->device));
    const int N = h->p.N, ld = (B + 15) / 16 * 16, n = h->st.sp.n, m = h->st.sp.m;
    const int rx = 6, rr = 6 * (N + 1);
    int rc = h->hs.ensure((size_t)(rx + rr + n + 2 * m) * ld, 0);
    if (rc) return rc;
    double *dx0 = h->hs.d, *dref = dx0 + (size_t)rx * ld, *dq = dref + (size_t)rr * ld;
    double *dl = dq + (size_t)n * ld, *du = dl + (size_t)m * ld;
    std::vector<double> tmp;
    to_soa(x0, B, rx, ld, tmp);
    HIP_TRY(hipMemcpy(dx0, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice));
    to_soa(ref, B, rr, ld, tmp);
    HIP_TRY(hipMemcpy(dref, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice));
    rc = tg_mpc6_assemble_batch(h, B, ld, dx0, dref, q ? dq : nullptr, l ? dl : nullptr, u ? du : nullptr, nullptr);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    struct Out { double *host; double *dev; int rows; } outs[3] = {{q, dq, n}, {l, dl, m}, {u, du, m}};
    for (const Out &o : outs) {
        if (!o.host) continue;
        tmp.resize((size_t)o.rows * ld);
        HIP_TRY(hipMemcpy(tmp.data(), o.dev, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
        from_soa(tmp, B, o.rows, ld, o.host);
    }
    return TG_OK;
}

int tg_mpc6_plant_step(tg_mpc6 *h, int B, int ld, double *d_x0, double *d_U, const int *d_status, void *stream)
{
    if (!h) return fail(TG_E_ARG, "null handle");
    if (B < 0 || ld < B) return fail(TG_E_ARG, "need 0 <= B <= ld");
    if (B == 0) return TG_OK;
    if (!d_x0 || !d_U) return fail(TG_E_ARG, "null device array");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(launch_plant_step(6, 3, h->p.N, h->st.sp.mdl.Ad, h->st.sp.mdl.Bd, B, ld, d_x0, d_U, d_status, 1, (hipStream_t)stream));
    return TG_OK;
}

int tg_ref_shift(int B, int ld, int nx, int N, double *d_ref, void *stream)
{
    if (B < 0 || ld < B || nx < 1 || nx > 12 || N < 1) return fail(TG_E_ARG, "bad argument");
    if (B == 0) return TG_OK;
    if (!d_ref) return fail(TG_E_ARG, "null device array");
    HIP_TRY(launch_ref_shift(B, ld, nx, N, d_ref, (hipStream_t)stream));
    return TG_OK;
}

int tg_mpc6_reset_warm(tg_mpc6 *h) { if (!h) return fail(TG_E_ARG, "null handle"); h->st.warm_valid = false; return TG_OK; }
int tg_mpc6_dims(const tg_mpc6 *h, int *n, int *m)
{
    if (!h) return fail(TG_E_ARG, "null handle");
    if (n) *n = h->st.sp.n;
    if (m) *m = h->st.sp.m;
    return TG_OK;
}
int tg_mpc6_get(const tg_mpc6 *h, int which, double *out)
{
    if (!h || !out) return fail(TG_E_ARG, "null argument");
    return h->st.get(which, out);
}
double tg_mpc6_rho(const tg_mpc6 *h) { return h ? h->st.sp.rho : 0.0; }
int tg_mpc6_kernel_info(const tg_mpc6 *h, int *mfma, int *lowrank)
{
    if (!h) return fail(TG_E_ARG, "null handle");
    h->st.kernel_info(mfma, lowrank);
    return TG_OK;
}
int tg_mpc6_get_opts(const tg_mpc6 *h, tg_solver_opts *out)
{
    if (!h || !out) return fail(TG_E_ARG, "null argument");
    h->st.effective_opts(h->o, out);
    return TG_OK;
}
// true when `o` asks for the factorised operator the handle already has
static bool same_operator(const tg_solver_opts &have, const tg_solver_opts &o)
{
    bool same = (o.rho <= 0 || o.rho == have.rho) && o.sigma == have.sigma;
    for (int k = 0; k < 4; k++) same = same && (o.rho_scale[k] <= 0 || o.rho_scale[k] == have.rho_scale[k]);
    return same;
}
int tg_mpc6_set_opts(tg_mpc6 *h, const tg_solver_opts *o)
{
    if (!h || !o) return fail(TG_E_ARG, "null argument");
    int rc = check_opts(*o);
    if (rc) return rc;
    tg_solver_opts have;
    h->st.effective_opts(h->o, &have);
    if (!same_operator(have, *o)) return fail(TG_E_STATE, "rho, sigma and rho_scale are frozen at create: make a new handle");
    h->st.set_run_opts(*o);
    h->o.alpha = o->alpha; h->o.eps_abs = o->eps_abs; h->o.eps_rel = o->eps_rel; h->o.eps_prim_inf = o->eps_prim_inf;
    h->o.max_iter = o->max_iter; h->o.check_termination = o->check_termination; h->o.check_fine = o->check_fine;
    h->o.grid_cap = o->grid_cap;
    h->o.hard_first = o->hard_first;
    return TG_OK;
}
int tg_mpc6_last_residuals(tg_mpc6 *h, int B, double *prim, double *dual)
{
    if (!h) return fail(TG_E_ARG, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    return h->st.residuals(B, prim, dual);
}

// ================================================================ 12-state
// Where a 12-state solve keeps what passes between its stages: the handle's own hipMalloc'ed
// arrays (ensure_inter) or, for the entry point that may not allocate, a carve-up of the device's
// static scratch array.
struct InterView {
    double *xy_max = nullptr, *mixed_max = nullptr, *yaw_ref = nullptr;
    double *Xs = nullptr, *Us = nullptr;           // the stages' solutions until all three have solved: [12(N+1)][ld], [4N][ld]
    int ld = 0;
    int *queue[3] = {nullptr, nullptr, nullptr};   // per stage slot (z, xy, yaw); null: the stage's own counter
    bool no_alloc = false;
};

struct tg_mpc12 {
    tg_mpc12_params p; tg_solver_opts o; int device;
    DevStage st[4]; // indexed by TG_FAM_*; [0] unused
    HostStaging hs;
    // xy_max [6N] | mixed_max [6N] | yaw_ref [N+1] | stage status (2 rows) | Xs [12(N+1)] | Us [4N]
    double *d_inter = nullptr; int inter_ld = 0;
    // Overlap of the XY stage's continuation launches with the yaw stage of the vehicles whose XY
    // problem the first launch solved: a second stream, two events, and [gate: cap ints | late
    // list: cap ints | late count] of device memory.
    hipStream_t s2 = nullptr; hipEvent_t ev_xy0 = nullptr, ev_yaw = nullptr;
    int *d_ovl = nullptr; int ovl_cap = 0;
    int ensure_overlap(int ld)
    {
        if (ld <= ovl_cap) return TG_OK;
        hipFree(d_ovl); d_ovl = nullptr; ovl_cap = 0;
        const int cap = (ld + 63) / 64 * 64;
        HIP_TRY(hipMalloc((void **)&d_ovl, ((size_t)2 * cap + 64) * sizeof(int)));
        HIP_TRY(hipMemset(d_ovl + (size_t)2 * cap, 0, 64 * sizeof(int)));     // the count; every chain's merge clears it again
        ovl_cap = cap;
        return TG_OK;
    }
    int ensure_inter(int ld)
    {
        if (ld <= inter_ld) return TG_OK;
        hipFree(d_inter); d_inter = nullptr; inter_ld = 0;
        // 2 rows hold the [3][ld] stage status used to gate the later stages when the caller
        // passes no status array; Xs / Us take the stages' solutions, which reach the caller's
        // X / U only for instances whose three stages all solved (launch_merge12)
        HIP_TRY(hipMalloc((void **)&d_inter, (size_t)(13 * p.N + 1 + 2 + 12 * (p.N + 1) + 4 * p.N) * ld * sizeof(double)));
        inter_ld = ld;
        return TG_OK;
    }
    int *own_status() { return (int *)(d_inter + (size_t)(13 * p.N + 1) * inter_ld); }
    double *xy_max() { return d_inter; }
    double *mixed_max() { return d_inter + (size_t)6 * p.N * inter_ld; }
    double *yaw_ref() { return d_inter + (size_t)12 * p.N * inter_ld; }
    double *plan_X() { return d_inter + (size_t)(13 * p.N + 1 + 2) * inter_ld; }
    double *plan_U() { return plan_X() + (size_t)12 * (p.N + 1) * inter_ld; }
    InterView view();
};

InterView tg_mpc12::view()
{
    InterView iv;
    iv.xy_max = xy_max(); iv.mixed_max = mixed_max(); iv.yaw_ref = yaw_ref(); iv.ld = inter_ld;
    iv.Xs = plan_X(); iv.Us = plan_U();
    return iv;
}

int tg_mpc12_create(const tg_mpc12_params *p, const tg_solver_opts *o, const double *rho_xyz,
                    int device, tg_mpc12 **out)
{
    if (!p || !out) return fail(TG_E_ARG, "null argument");
    *out = nullptr;
    if (p->N < 1 || !(p->dt > 0)) return fail(TG_E_ARG, "N must be >= 1 and dt > 0");
    const double nonneg[] = {p->xy_state_weight, p->z_state_weight, p->yaw_state_weight, p->xy_input_weight,
                             p->z_input_weight, p->yaw_input_weight, p->xy_smooth_input_weight,
                             p->z_smooth_input_weight, p->yaw_smooth_input_weight, p->xy_max_vel, p->z_max_vel,
                             p->yaw_max_vel, p->xy_max_accel, p->z_max_accel, p->yaw_max_accel, p->xy_max_jerk,
                             p->z_max_jerk, p->yaw_max_jerk};
    for (double v : nonneg) if (v < 0) return fail(TG_E_ARG, "weights and limits must be >= 0");
    tg_solver_opts oo;
    if (o) oo = *o; else tg_default_opts(&oo);
    int rc = check_opts(oo);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(device));
    // address of this device's static scratch array (tg_mpc12_solve_batch_static), looked up once per device
    HIP_TRY(static_scratch_init(device));
    tg_mpc12 *h = new tg_mpc12();
    h->p = *p; h->o = oo; h->device = device;
    const int fams[3] = {TG_FAM_XY, TG_FAM_Z, TG_FAM_YAW};
    for (int k = 0; k < 3; k++) {
        double rho = rho_xyz ? rho_xyz[k] : oo.rho;
        rc = h->st[fams[k]].init(model_12(*p, fams[k]), oo, rho);
        if (rc) { tg_mpc12_destroy(h); return rc; }
    }
    if (oo.overlap > 0) {
        if (hipStreamCreateWithFlags(&h->s2, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_xy0, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_yaw, hipEventDisableTiming) != hipSuccess) {
            tg_mpc12_destroy(h);
            return fail(TG_E_HIP, "stream / events of the stage overlap could not be created");
        }
    }
    *out = h;
    return TG_OK;
}

void tg_mpc12_destroy(tg_mpc12 *h)
{
    if (!h) return;
    hipSetDevice(h->device);
    if (h->s2) { hipStreamSynchronize(h->s2); hipStreamDestroy(h->s2); }
    if (h->ev_xy0) hipEventDestroy(h->ev_xy0);
    if (h->ev_yaw) hipEventDestroy(h->ev_yaw);
    for (int f = 1; f <= 3; f++) h->st[f].release();
    h->hs.release(); hipFree(h->d_inter); hipFree(h->d_ovl);
    delete h;
}

// The three stages of one MPC12STATE::mpcLoop and the merge.  The stages write their solutions to
// iv.Xs / iv.Us; the caller's X / U receive them only for instances whose three stages all
// solved: mpcLoop fills _x_opt / _u_opt in extractSolution, after the last stage
// (mpc_12state.cpp:1405-1449, 1528-1551), so a vehicle whose XY or yaw stage fails keeps the whole
// plan of its last successful call, Z rows included.
static int mpc12_stage(tg_mpc12 *h, const InterView &iv, int stage, int B, int ld, const double *d_x0, const double *d_ref,
                       double *d_X, double *d_U, int *d_status, int *d_iters, int warm, hipStream_t s,
                       bool gated, const DevStage::SolveExtra *ex = nullptr, const int *gate = nullptr);
static int mpc12_chain(tg_mpc12 *h, const InterView &iv, int B, int ld, const double *d_x0, const double *d_ref,
                       double *d_X, double *d_U, int *d_status, int *d_iters, int warm, hipStream_t s)
{
    int rc;
    if ((rc = mpc12_stage(h, iv, TG_FAM_Z, B, ld, d_x0, d_ref, iv.Xs, iv.Us, d_status, d_iters, warm, s, true))) return rc;
    int bound[2], oper[3];
    const bool overlap = !iv.no_alloc && h->s2 != nullptr && h->st[TG_FAM_XY].plan_phases(bound, oper) > 1;
    if (!overlap) {
        if ((rc = mpc12_stage(h, iv, TG_FAM_XY, B, ld, d_x0, d_ref, iv.Xs, iv.Us, d_status, d_iters, warm, s, true))) return rc;
        if ((rc = mpc12_stage(h, iv, TG_FAM_YAW, B, ld, d_x0, d_ref, iv.Xs, iv.Us, d_status, d_iters, warm, s, true))) return rc;
        HIP_TRY(launch_merge12(B, ld, h->p.N, iv.Xs, iv.Us, d_status, d_X, d_U, nullptr, s));
        return TG_OK;
    }
    // The XY stage's first launch ends with most vehicles solved and a few handed over to its
    // continuation launches, which a slow or infeasible problem can keep busy for a long time on a
    // handful of CUs.  The yaw stage of the vehicles that are done does not wait for them: it runs
    // on a second stream beside the continuation launches, and the vehicles those solve get their
    // yaw stage afterwards, from the list the continuation launches write.  Every vehicle's three
    // solves are what they are in the sequential chain, bit for bit.
    // (The early pass forms the yaw reference of every vehicle, also of those whose XY rows the
    // continuation launches are still writing: such a vehicle is gated out of that pass by the copy of
    // the XY statuses taken before, and its reference is formed again for the late pass, after the
    // continuation launches and after the early pass has ended.  The two passes never run at once,
    // so the yaw stage's own buffers -- warm rows, hand-over lists, queue counter -- see one at a time.)
    if ((rc = h->ensure_overlap(ld))) return rc;
    int *gate = h->d_ovl, *late = gate + h->ovl_cap, *late_n = late + h->ovl_cap;
    const int yaw_warm_in = warm ? (h->st[TG_FAM_YAW].warm_rows_valid(B) ? 1 : 0) : -1;
    DevStage::SolveExtra yx;           // both yaw passes read the warm rows, or neither
    yx.warm_in = yaw_warm_in;
    DevStage::SolveExtra xx;
    xx.done_list = late; xx.done_count = late_n;
    xx.after_first = [&]() -> int {
        // the XY statuses as the first launch left them (handed-over vehicles: not attempted yet) gate the early yaw pass
        HIP_TRY(hipMemcpyAsync(gate, d_status + (size_t)ld, (size_t)B * sizeof(int), hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipEventRecord(h->ev_xy0, s));
        HIP_TRY(hipStreamWaitEvent(h->s2, h->ev_xy0, 0));
        const int r = mpc12_stage(h, iv, TG_FAM_YAW, B, ld, d_x0, d_ref, iv.Xs, iv.Us, d_status, d_iters, warm, h->s2, true, &yx, gate);
        if (r != TG_OK) return r;
        HIP_TRY(hipEventRecord(h->ev_yaw, h->s2));
        return TG_OK;
    };
    rc = mpc12_stage(h, iv, TG_FAM_XY, B, ld, d_x0, d_ref, iv.Xs, iv.Us, d_status, d_iters, warm, s, true, &xx);
    if (rc != TG_OK) {                 // part of the chain may be in flight on the second stream: let it end, start the count over
        hipStreamSynchronize(h->s2); hipStreamSynchronize(s);
        hipMemset(late_n, 0, sizeof(int));
        return rc;
    }
    HIP_TRY(hipStreamWaitEvent(s, h->ev_yaw, 0));
    DevStage::SolveExtra lx = yx;
    lx.first_list = late; lx.first_count = late_n;
    if ((rc = mpc12_stage(h, iv, TG_FAM_YAW, B, ld, d_x0, d_ref, iv.Xs, iv.Us, d_status, d_iters, warm, s, true, &lx))) {
        hipStreamSynchronize(s);
        hipMemset(late_n, 0, sizeof(int));
        return rc;
    }
    HIP_TRY(launch_merge12(B, ld, h->p.N, iv.Xs, iv.Us, d_status, d_X, d_U, late_n, s));
    return TG_OK;
}

// gated: instances whose upstream stage did not solve are skipped (status TG_UNSOLVED), as
// MPC12STATE::mpcLoop returns at the first failing stage (mpc_12state.cpp:1374-1453); gate: the
// statuses to go by, if not the upstream stage's row of d_status
static int mpc12_stage(tg_mpc12 *h, const InterView &iv, int stage, int B, int ld, const double *d_x0, const double *d_ref,
                       double *d_X, double *d_U, int *d_status, int *d_iters, int warm, hipStream_t s,
                       bool gated, const DevStage::SolveExtra *ex, const int *gate)
{
    const tg_mpc12_params &p = h->p;
    if (iv.ld != ld) return fail(TG_E_STATE, "leading dimension changed between stages");
    BatchArgs a{};
    a.B = B; a.ld = ld; a.ref_ld = ld; a.ov_ld = iv.ld;
    a.x0 = d_x0; a.ref = d_ref; a.X = d_X; a.U = d_U;
    const int slot = stage == TG_FAM_Z ? 0 : stage == TG_FAM_XY ? 1 : 2;
    a.status = d_status ? d_status + (size_t)slot * ld : nullptr;
    a.iters = d_iters ? d_iters + (size_t)slot * ld : nullptr;
    if (gated && slot > 0 && d_status) a.gate = gate ? gate : d_status + (size_t)(slot - 1) * ld;
    if (stage == TG_FAM_XY) {
        // computeXYBounds from the Z solution already in X, then updateXYQP + solve
        HIP_TRY(launch_xy_bounds(B, ld, p.N, p.xy_max_vel, p.xy_max_accel, p.z_max_accel, d_X,
                                 iv.xy_max, iv.mixed_max, s));
        a.ov_box = iv.xy_max; a.ov_mixed = iv.mixed_max;
    } else if (stage == TG_FAM_YAW) {
        HIP_TRY(launch_yaw_ref(B, ld, p.N, d_ref, d_X, iv.yaw_ref, s));
        a.ref = iv.yaw_ref; a.ref_ld = iv.ld;
    }
    if (iv.no_alloc) a.queue_mem = h->st[stage].grid_cap >= 0 ? iv.queue[slot] : nullptr;
    return h->st[stage].solve(a, warm, s, iv.no_alloc, ex);
}

static int mpc12_check(tg_mpc12 *h, int B, int ld, const void *a, const void *b, const void *c, const void *d)
{
    if (!h) return fail(TG_E_ARG, "null handle");
    if (B < 0 || ld < B) return fail(TG_E_ARG, "need 0 <= B <= ld");
    if (B > 0 && (!a || !b || !c || !d)) return fail(TG_E_ARG, "null device array");
    return TG_OK;
}

int tg_mpc12_solve_batch(tg_mpc12 *h, int B, int ld, const double *d_x0, const double *d_ref,
                         double *d_X, double *d_U, int *d_status, int *d_iters, int warm, void *stream)
{
    int rc = mpc12_check(h, B, ld, d_x0, d_ref, d_X, d_U);
    if (rc || B == 0) return rc;
    HIP_TRY(hipSetDevice(h->device));
    // keep one leading dimension for the intermediates of a handle
    if (h->inter_ld != 0 && h->inter_ld != ld) { hipFree(h->d_inter); h->d_inter = nullptr; h->inter_ld = 0; }
    hipStream_t s = (hipStream_t)stream;
    if ((rc = h->ensure_inter(ld))) return rc;
    if (!d_status) d_status = h->own_status();
    return mpc12_chain(h, h->view(), B, ld, d_x0, d_ref, d_X, d_U, d_status, d_iters, warm, s);
}

// The same three stages with every piece of device memory beyond the caller's arrays taken from
// the device's static scratch array: [3 queue counters, 64 B apart | status 3 x ld ints | xy_max 6N x ld |
// mixed_max 6N x ld | yaw_ref (N+1) x ld | Xs 12(N+1) x ld | Us 4N x ld doubles].  Nothing here allocates:
// no hipMalloc / hipFree, no warm start, no residual record, no timing events.
unsigned long long tg_mpc12_static_scratch_bytes(int N, int ld)
{
    if (N < 1 || ld < 1) return 0;
    const unsigned long long status = ((unsigned long long)3 * ld * sizeof(int) + 255) / 256 * 256;
    return 256 + status + (unsigned long long)(13 * (unsigned long long)N + 1 + 12 * ((unsigned long long)N + 1) + 4 * (unsigned long long)N) * ld * sizeof(double);
}

int tg_mpc12_solve_batch_static(tg_mpc12 *h, int B, int ld, const double *d_x0, const double *d_ref,
                                double *d_X, double *d_U, int *d_status, int *d_iters, void *stream)
{
    int rc = mpc12_check(h, B, ld, d_x0, d_ref, d_X, d_U);
    if (rc || B == 0) return rc;
    const unsigned long long need = tg_mpc12_static_scratch_bytes(h->p.N, ld);
    if (need > kStaticScratchBytes)
        return fail(TG_E_ARG, "batch too large for the static scratch: N = " + std::to_string(h->p.N) + ", ld = " +
                                  std::to_string(ld) + " need " + std::to_string(need) + " bytes of " +
                                  std::to_string(kStaticScratchBytes) + " (use tg_mpc12_solve_batch, or split the batch)");
    HIP_TRY(hipSetDevice(h->device));
    unsigned char *base = (unsigned char *)static_scratch_ptr(h->device);
    if (!base) return fail(TG_E_STATE, "static scratch of this device was not initialised");
    InterView iv;
    iv.no_alloc = true;
    iv.ld = ld;
    for (int k = 0; k < 3; k++) iv.queue[k] = (int *)(base + 64 * k);
    int *own_status = (int *)(base + 256);
    const size_t status_bytes = ((size_t)3 * ld * sizeof(int) + 255) / 256 * 256;
    iv.xy_max = (double *)(base + 256 + status_bytes);
    iv.mixed_max = iv.xy_max + (size_t)6 * h->p.N * ld;
    iv.yaw_ref = iv.mixed_max + (size_t)6 * h->p.N * ld;
    iv.Xs = iv.yaw_ref + (size_t)(h->p.N + 1) * ld;
    iv.Us = iv.Xs + (size_t)12 * (h->p.N + 1) * ld;
    if (!d_status) d_status = own_status;
    return mpc12_chain(h, iv, B, ld, d_x0, d_ref, d_X, d_U, d_status, d_iters, 0, (hipStream_t)stream);
}

int tg_mpc12_solve_stage(tg_mpc12 *h, int stage, int B, int ld, const double *d_x0, const double *d_ref,
                         double *d_X, double *d_U, int *d_status, int *d_iters, void *stream)
{
    int rc = mpc12_check(h, B, ld, d_x0, d_ref, d_X, d_U);
    if (rc || B == 0) return rc;
    if (stage < TG_FAM_XY || stage > TG_FAM_YAW) return fail(TG_E_ARG, "stage must be XY, Z or YAW");
    HIP_TRY(hipSetDevice(h->device));
    if (h->inter_ld != 0 && h->inter_ld != ld) { hipFree(h->d_inter); h->d_inter = nullptr; h->inter_ld = 0; }
    if ((rc = h->ensure_inter(ld))) return rc;
    return mpc12_stage(h, h->view(), stage, B, ld, d_x0, d_ref, d_X, d_U, d_status, d_iters, 0, (hipStream_t)stream, false);
}

int tg_mpc12_solve_host(tg_mpc12 *h, int B, const double *x0, const double *ref, double *X, double *U,
                        int *status, int *iters, int warm)
{
    if (!h) return fail(TG_E_ARG, "null handle");
    if (B < 0) return fail(TG_E_ARG, "B < 0");
    if (B == 0) return TG_OK;
    if (!x0 || !ref || !X || !U) return fail(TG_E_ARG, "null host array");
    HIP_TRY(hipSetDevice(h->device));
    const int N = h->p.N, ld = (B + 15) / 16 * 16;
    const int rx = 12, rr = 12 * (N + 1), rX = 12 * (N + 1), rU = 4 * N;
    int rc = h->hs.ensure((size_t)(rx + rr + rX + rU) * ld, (size_t)6 * ld);
    if (rc) return rc;
    double *dx0 = h->hs.d, *dref = dx0 + (size_t)rx * ld, *dX = dref + (size_t)rr * ld, *dU = dX + (size_t)rX * ld;
    std::vector<double> tmp;
    to_soa(x0, B, rx, ld, tmp);
    HIP_TRY(hipMemcpy(dx0, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice));
    to_soa(ref, B, rr, ld, tmp);
    HIP_TRY(hipMemcpy(dref, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(dX, 0, (size_t)(rX + rU) * ld * sizeof(double)));
    rc = tg_mpc12_solve_batch(h, B, ld, dx0, dref, dX, dU, h->hs.di, h->hs.di + (size_t)3 * ld, warm, nullptr);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    tmp.resize((size_t)rX * ld);
    HIP_TRY(hipMemcpy(tmp.data(), dX, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
    from_soa(tmp, B, rX, ld, X);
    tmp.resize((size_t)rU * ld);
    HIP_TRY(hipMemcpy(tmp.data(), dU, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
    from_soa(tmp, B, rU, ld, U);
    std::vector<int> ti((size_t)6 * ld);
    HIP_TRY(hipMemcpy(ti.data(), h->hs.di, ti.size() * sizeof(int), hipMemcpyDeviceToHost));
    for (int b = 0; b < B; b++)
        for (int k = 0; k < 3; k++) {
            if (status) status[3 * b + k] = ti[(size_t)k * ld + b];
            if (iters) iters[3 * b + k] = ti[(size_t)(3 + k) * ld + b];
        }
    return TG_OK;
}

int tg_mpc12_intermediates(tg_mpc12 *h, const double **d_xy_max, const double **d_mixed_max,
                           const double **d_yaw_ref, int *ld)
{
    if (!h) return fail(TG_E_ARG, "null handle");
    if (!h->d_inter) return fail(TG_E_STATE, "no solve has run on this handle yet");
    if (d_xy_max) *d_xy_max = h->xy_max();
    if (d_mixed_max) *d_mixed_max = h->mixed_max();
    if (d_yaw_ref) *d_yaw_ref = h->yaw_ref();
    if (ld) *ld = h->inter_ld;
    return TG_OK;
}

int tg_mpc12_assemble_stage(tg_mpc12 *h, int stage, int B, int ld, const double *d_x0, const double *d_ref,
                            const double *d_X, double *d_q, double *d_l, double *d_u, void *stream)
{
    if (!h) return fail(TG_E_ARG, "null handle");
    if (B < 0 || ld < B) return fail(TG_E_ARG, "need 0 <= B <= ld");
    if (B == 0) return TG_OK;
    if (stage < TG_FAM_XY || stage > TG_FAM_YAW) return fail(TG_E_ARG, "stage must be XY, Z or YAW");
    if (!d_x0 || !d_ref) return fail(TG_E_ARG, "null device array");
    if ((d_l == nullptr) != (d_u == nullptr)) return fail(TG_E_ARG, "l and u go together");
    if (stage != TG_FAM_Z && !d_X) return fail(TG_E_ARG, "XY / Yaw assembly needs the upstream solution in X");
    HIP_TRY(hipSetDevice(h->device));
    if (h->inter_ld != 0 && h->inter_ld != ld) { hipFree(h->d_inter); h->d_inter = nullptr; h->inter_ld = 0; }
    int rc = h->ensure_inter(ld);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const tg_mpc12_params &p = h->p;
    BatchArgs a{};
    a.B = B; a.ld = ld; a.ref_ld = ld; a.ov_ld = h->inter_ld; a.x0 = d_x0; a.ref = d_ref;
    if (stage == TG_FAM_XY) {
        HIP_TRY(launch_xy_bounds(B, ld, p.N, p.xy_max_vel, p.xy_max_accel, p.z_max_accel, d_X, h->xy_max(),
                                 h->mixed_max(), s));
        a.ov_box = h->xy_max(); a.ov_mixed = h->mixed_max();
    } else if (stage == TG_FAM_YAW) {
        HIP_TRY(launch_yaw_ref(B, ld, p.N, d_ref, d_X, h->yaw_ref(), s));
        a.ref = h->yaw_ref(); a.ref_ld = h->inter_ld;
    }
    HIP_TRY(launch_assemble(h->st[stage].adev, a, d_q, d_l, d_u, s));
    return TG_OK;
}

int tg_mpc12_assemble_host(tg_mpc12 *h, int stage, int B, const double *x0, const double *ref, const double *X,
                           double *q, double *l, double *u)
{
    if (!h) return fail(TG_E_ARG, "null handle");
    if (B < 0) return fail(TG_E_ARG, "B < 0");
    if (B == 0) return TG_OK;
    if (stage < TG_FAM_XY || stage > TG_FAM_YAW) return fail(TG_E_ARG, "stage must be XY, Z or YAW");
    if (!x0 || !ref) return fail(TG_E_ARG, "null host array");
    if (stage != TG_FAM_Z && !X) return fail(TG_E_ARG, "XY / Yaw assembly needs the upstream solution in X");
    if ((l == nullptr) != (u == nullptr)) return fail(TG_E_ARG, "l and u go together");
    HIP_TRY(hipSetDevice(h->device));
    const int N = h->p.N, ld = (B + 15) / 16 * 16, n = h->st[stage].sp.n, m = h->st[stage].sp.m;
    const int rx = 12, rr = 12 * (N + 1);
    int rc = h->hs.ensure((size_t)(rx + 2 * rr + n + 2 * m) * ld, (size_t)6 * ld);
    if (rc) return rc;
    double *dx0 = h->hs.d, *dref = dx0 + (size_t)rx * ld, *dX = dref + (size_t)rr * ld, *dq = dX + (size_t)rr * ld;
    double *dl = dq + (size_t)n * ld, *du = dl + (size_t)m * ld;
    std::vector<double> tmp;
    to_soa(x0, B, rx, ld, tmp);
    HIP_TRY(hipMemcpy(dx0, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice));
    to_soa(ref, B, rr, ld, tmp);
    HIP_TRY(hipMemcpy(dref, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice));
    if (X) {
        to_soa(X, B, rr, ld, tmp);
        HIP_TRY(hipMemcpy(dX, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    rc = tg_mpc12_assemble_stage(h, stage, B, ld, dx0, dref, X ? dX : nullptr, q ? dq : nullptr,
                                 l ? dl : nullptr, u ? du : nullptr, nullptr);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    struct Out { double *host; double *dev; int rows; } outs[3] = {{q, dq, n}, {l, dl, m}, {u, du, m}};
    for (const Out &o : outs) {
        if (!o.host) continue;
        tmp.resize((size_t)o.rows * ld);
        HIP_TRY(hipMemcpy(tmp.data(), o.dev, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
        from_soa(tmp, B, o.rows, ld, o.host);
    }
    return TG_OK;
}

int tg_mpc12_plant_step(tg_mpc12 *h, int B, int ld, double *d_x0, double *d_U, const int *d_status, void *stream)
{
    if (!h) return fail(TG_E_ARG, "null handle");
    if (B < 0 || ld < B) return fail(TG_E_ARG, "need 0 <= B <= ld");
    if (B == 0) return TG_OK;
    if (!d_x0 || !d_U) return fail(TG_E_ARG, "null device array");
    HIP_TRY(hipSetDevice(h->device));
    // getTransitionMatrix / getInputMatrix (mpc_12state.cpp:2001-2025): blkdiag(XY, Z, Yaw)
    double A[144] = {0}, Bm[48] = {0};
    const int fam[3] = {TG_FAM_XY, TG_FAM_Z, TG_FAM_YAW};
    int r0 = 0, u0 = 0;
    for (int f : fam) {
        const StageModel &m = h->st[f].sp.mdl;
        for (int i = 0; i < m.nx; i++) {
            for (int j = 0; j < m.nx; j++) A[(r0 + i) * 12 + r0 + j] = m.Ad[i * m.nx + j];
            for (int k = 0; k < m.nu; k++) Bm[(r0 + i) * 4 + u0 + k] = m.Bd[i * m.nu + k];
        }
        r0 += m.nx; u0 += m.nu;
    }
    HIP_TRY(launch_plant_step(12, 4, h->p.N, A, Bm, B, ld, d_x0, d_U, d_status, 3, (hipStream_t)stream));
    return TG_OK;
}

int tg_mpc12_reset_warm(tg_mpc12 *h)
{
    if (!h) return fail(TG_E_ARG, "null handle");
    for (int f = 1; f <= 3; f++) h->st[f].warm_valid = false;
    return TG_OK;
}
int tg_mpc12_dims(const tg_mpc12 *h, int stage, int *n, int *m)
{
    if (!h || stage < 1 || stage > 3) return fail(TG_E_ARG, "bad argument");
    if (n) *n = h->st[stage].sp.n;
    if (m) *m = h->st[stage].sp.m;
    return TG_OK;
}
int tg_mpc12_get(const tg_mpc12 *h, int stage, int which, double *out)
{
    if (!h || !out || stage < 1 || stage > 3) return fail(TG_E_ARG, "bad argument");
    return h->st[stage].get(which, out);
}
double tg_mpc12_rho(const tg_mpc12 *h, int stage) { return (h && stage >= 1 && stage <= 3) ? h->st[stage].sp.rho : 0.0; }
int tg_mpc12_kernel_info(const tg_mpc12 *h, int stage, int *mfma, int *lowrank)
{
    if (!h || stage < 1 || stage > 3) return fail(TG_E_ARG, "bad argument");
    h->st[stage].kernel_info(mfma, lowrank);
    return TG_OK;
}
int tg_mpc12_get_opts(const tg_mpc12 *h, int stage, tg_solver_opts *out)
{
    if (!h || !out || stage < 1 || stage > 3) return fail(TG_E_ARG, "bad argument");
    h->st[stage].effective_opts(h->o, out);
    return TG_OK;
}
int tg_mpc12_set_opts(tg_mpc12 *h, const tg_solver_opts *o)
{
    if (!h || !o) return fail(TG_E_ARG, "null argument");
    int rc = check_opts(*o);
    if (rc) return rc;
    for (int f = 1; f <= 3; f++) {
        tg_solver_opts have;
        h->st[f].effective_opts(h->o, &have);
        tg_solver_opts want = *o;
        // rho and the per-class scales are per stage (rho_xyz / the built-in choices, which differ
        // between the stages) and frozen at create: what tg_mpc12_get_opts returned for any one
        // stage is accepted for all three, and neither is looked at
        want.rho = 0.0;
        for (int k = 0; k < 4; k++) want.rho_scale[k] = 0.0;
        if (!same_operator(have, want)) return fail(TG_E_STATE, "sigma is frozen at create: make a new handle");
    }
    for (int f = 1; f <= 3; f++) h->st[f].set_run_opts(*o);
    h->o.alpha = o->alpha; h->o.eps_abs = o->eps_abs; h->o.eps_rel = o->eps_rel; h->o.eps_prim_inf = o->eps_prim_inf;
    h->o.max_iter = o->max_iter; h->o.check_termination = o->check_termination; h->o.check_fine = o->check_fine;
    h->o.grid_cap = o->grid_cap;
    h->o.hard_first = o->hard_first;
    return TG_OK;
}
int tg_mpc12_last_residuals(tg_mpc12 *h, int stage, int B, double *prim, double *dual)
{
    if (!h || stage < 1 || stage > 3) return fail(TG_E_ARG, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    return h->st[stage].residuals(B, prim, dual);
}

} // extern "C"
