"""Random model parameters x solver options x inputs, both families, against the CPU oracle in the
engine's mode: statuses and iteration counts must be equal and solved trajectories agree to 1e-8
(development aid; GPU box).  Options drawn: eps, alpha, sigma, max_iter, check interval, finer
checks, infeasibility tolerance, step-size class scales, operator form, grid size (slot refill),
instance order; inputs: seeded vehicles with a random share in the stress geometry; cold starts.
usage: fuzz_vs_oracle.py [seed [trials]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch          # before the engine: both bring a HIP runtime, torch's has to be the one that initialises
import trajectory_generation_amd as tg
from trajectory_generation_amd import synth
from oracle import oracle as O
O.build()
SEED = int(sys.argv[1]) if len(sys.argv) > 1 else 1
TRIALS = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rng = np.random.default_rng(SEED)
rng_sched = np.random.default_rng(SEED + 7777)   # step-size schedules from a stream of their own: the other draws of a seed stay what they were
rng_tie = np.random.default_rng(SEED + 991)     # one-ulp input changes that tell a numerical tie from a finding
bad = 0
DUMP = os.environ.get("FUZZ_DUMP")               # directory: inputs, parameters and options of every 12-state trial run (with FUZZ_ONLY)
WILD = bool(os.environ.get("FUZZ_WILD"))      # parameters far outside the usual ranges
PROGRESS = bool(os.environ.get("FUZZ_PROGRESS"))   # a line before every engine / oracle call: a stall names its place
ONLY = [int(v) for v in os.environ.get("FUZZ_ONLY", "").split(",") if v]     # run only these trials (all are drawn)


def mark(msg):
    if PROGRESS:
        print("   ... " + msg, flush=True)


def draw_opts(N):
    chk = int(rng.choice([5, 10, 25, 25, 25, 50]))
    fine = int(rng.choice([0, 0, 5])) if chk % 5 == 0 and chk > 5 else 0
    mi = int(rng.choice([1, 2, 4, 160])) * chk
    eps = float(rng.choice([1e-2, 1e-3, 1e-3, 1e-4, 1e-5, 1e-7]))      # 1e-7: the step sizes of the tight tiers
    o = dict(eps_abs=eps, eps_rel=eps * float(rng.choice([1.0, 1.0, 0.1])), alpha=float(rng.choice([1.0, 1.4, 1.6, 1.6, 1.8])),
             sigma=float(rng.choice([1e-6, 1e-6, 1e-4])), max_iter=mi, check_termination=chk, check_fine=fine,
             eps_prim_inf=float(rng.choice([1e-4, 1e-4, 1e-3, 0.0])))
    e = dict(o)
    if rng.integers(0, 3) == 0:
        e["rho_scale"] = tuple(float(v) for v in 10 ** rng.uniform(-0.7, 0.7, 4))
    e["lowrank"] = int(rng.choice([0, 0, 1, -1]))
    e["grid_cap"] = int(rng.choice([0, 0, -1, 1, 2, 3]))
    e["hard_first"] = int(rng.choice([0, -1]))
    # step-size schedule: the built-in one, none, or a drawn one (boundaries anywhere, not only on the check grid)
    kind = int(rng_sched.integers(0, 4))
    if os.environ.get("FUZZ_NO_SCHEDULE"):
        kind = 1
    if kind == 1:
        e["phase_iters"] = (-1, -1)
    elif kind >= 2:
        k1 = int(rng_sched.integers(1, 8)) * int(rng_sched.choice([5, 10, 25, 33]))
        k2 = k1 + int(rng_sched.integers(1, 8)) * int(rng_sched.choice([5, 10, 25, 40])) if kind == 3 else -1
        e["phase_iters"] = (k1, k2)
        e["phase_scale"] = tuple(tuple(float(v) for v in 10 ** rng_sched.uniform(-0.5, 1.5, 4)) for _ in range(2))
    e["overlap"] = int(rng_sched.integers(0, 2))      # 12-state chain: yaw stage of the early XY finishers beside the XY continuation launches
    import json
    e.update(json.loads(os.environ.get("FUZZ_ENGINE_ONLY", "{}")))      # overrides for bisecting a finding
    for k, v in json.loads(os.environ.get("FUZZ_BOTH", "{}")).items():
        o[k] = v; e[k] = v
    return o, e


def report(tag, trial, desc, r, ro, solved, tol):
    global bad
    mism = int((r["status"] != ro["status"]).sum() + (r["iters"] != ro["iters"]).sum())
    d = float(np.abs(r["X"][solved] - ro["X"][solved]).max() / max(1.0, np.abs(ro["X"][solved]).max())) if solved.any() else 0.0   # relative to the largest entry
    flag = "" if mism == 0 and d < tol else "  <-- MISMATCH"
    bad += bool(flag)
    print("%s trial %3d %s solved %d/%d statuses %s itmax %d mism %d d %.1e%s" % (
        tag, trial, desc, solved.sum(), len(solved), sorted(set(int(v) for v in np.unique(ro["status"]))), ro["iters"].max(), mism, d, flag), flush=True)
    if flag:
        idx = np.argwhere((r["status"] != ro["status"]) | (r["iters"] != ro["iters"]))[:6]
        for ix in idx:
            ix = tuple(ix)
            print("      at %s: engine %d after %d, oracle %d after %d" % (ix, r["status"][ix], r["iters"][ix], ro["status"][ix], ro["iters"][ix]))
        if r["X"].shape[1] % 12 == 0 and r["status"].ndim == 2 and solved.any():
            # 12-state chain: where the trajectories part (columns x, y | z | yaw of every step).  The yaw
            # reference is atan2 of (target - planned position) in y and x (mpc_12state.cpp:544-548): where
            # that vector is next to zero, or points along the +-pi cut, it amplifies rounding-level
            # differences of the XY solution
            dX = np.abs(r["X"][solved] - ro["X"][solved]).reshape(int(solved.sum()), -1, 12)
            print("      max-abs difference of the solved instances: xy %.1e  z %.1e  yaw %.1e" % (
                dX[:, :, :6].max(), dX[:, :, 6:9].max(), dX[:, :, 9:].max()), flush=True)


for trial in range(TRIALS):
    B = int(rng.integers(1, 100))
    big = rng.integers(0, 12) == 0           # now and then a batch larger than the GPU holds at once: the default grid refills
    if big:
        B = int(rng.integers(4200, 9000))
    dt = float(rng.choice([0.05, 0.1, 0.1, 0.2]))
    smooth = bool(rng.integers(0, 2))
    share = float(rng.choice([0.0, 0.0, 0.2, 1.0]))
    # ---------------------------------------------------------------- 6-DOF
    N = int(rng.integers(1, 82)) if not big else int(rng.integers(1, 16))
    o, e = draw_opts(N)
    if big:
        e["grid_cap"] = 0
        o["max_iter"] = e["max_iter"] = min(o["max_iter"], 400)    # the oracle runs these thousands of instances on one core
    kw6 = dict(N=N, dt=dt, state_weight=float(10 ** rng.uniform(2, 4.3)), input_weight=float(10 ** rng.uniform(1, 3)),
               smooth_input_weight=float(10 ** rng.uniform(0, 2)), enable_control_smoothing=smooth,
               min_altitude=float(rng.uniform(-2, 1)), max_vel=tuple(float(v) for v in rng.uniform(3, 15, 3)),
               max_accel=tuple(float(v) for v in rng.uniform(2, 10, 3)))
    if WILD:        # far outside what a vehicle would be configured with
        dt = float(rng.choice([0.01, 0.02, 0.5, 1.0])); kw6["dt"] = dt
        kw6.update(state_weight=float(10 ** rng.uniform(-1, 6)), input_weight=float(10 ** rng.uniform(-2, 5)),
                   smooth_input_weight=float(10 ** rng.uniform(-2, 4)), min_altitude=float(rng.uniform(-50, 4)),
                   max_vel=tuple(float(v) for v in 10 ** rng.uniform(-0.5, 2.5, 3)),
                   max_accel=tuple(float(v) for v in 10 ** rng.uniform(-0.5, 2.5, 3)))
    desc = "N %2d B %2d dt %.2f sm %d %s" % (N, B, dt, smooth, {k: e[k] for k in ("eps_abs", "alpha", "max_iter", "check_termination", "check_fine", "lowrank", "grid_cap", "phase_iters") if k in e})
    try:
        if ONLY and trial not in ONLY:
            raise KeyError("skipped")
        eng = tg.BatchedMPC6(tg.mpc6_params(**kw6), tg.make_opts(**e))
        x0, ref = O.synth_mpc6(1000 * SEED + trial, N, dt, B)
        ns = int(share * B)
        if ns:
            x0[:ns], ref[:ns] = synth.stress("6dof", x0[:ns], ref[:ns], N)
        if rng.integers(0, 4) == 0:            # a few vehicles report a speed outside the limits: infeasible QPs
            for i in rng.choice(B, size=min(B, 3), replace=False):
                ax = int(rng.integers(0, 3))
                x0[i, 3 + ax] = float(rng.choice([-1, 1])) * kw6["max_vel"][ax] * float(rng.uniform(1.05, 3.0))
        O.match_engine(eng)
        op = O.mpc6_params(**kw6)
        mark("6dof  trial %3d %s: engine" % (trial, desc))
        r = eng.solve_host(x0, ref)
        mark("6dof  trial %3d engine done (iters max %d), oracle" % (trial, r["iters"].max()))
        ro = O.mpc6_batch(op, O.fixed_rho_settings(eng.rho, warm_start=0, **o), x0, ref)
        before = bad
        report("6dof ", trial, desc, r, ro, ro["status"] == 1, 1e-8)
        diff = np.where((r["status"] != ro["status"]) | (r["iters"] != ro["iters"]))[0]
        if bad > before and 0 < len(diff) <= 4 and np.abs(r["X"][ro["status"] == 1] - ro["X"][ro["status"] == 1]).max() < 1e-6:
            # A residual within rounding noise of its threshold at a check?  Then the oracle's own count moves
            # when the inputs move by one ulp, and takes the engine's value among others: a tie, not a finding.
            tie = True
            for b in diff:
                seen = set()
                for _ in range(24):
                    xa = x0[b:b + 1] * (1 + 2.2e-16 * rng_tie.integers(-1, 2, (1, 6)))
                    ra = ref[b:b + 1] * (1 + 2.2e-16 * rng_tie.integers(-1, 2, ref[b:b + 1].shape))
                    rp = O.mpc6_batch(op, O.fixed_rho_settings(eng.rho, warm_start=0, **o), xa, ra)
                    seen.add((int(rp["status"][0]), int(rp["iters"][0])))
                tie = tie and (int(r["status"][b]), int(r["iters"][b])) in seen and len(seen) > 1
                print("      instance %d with its inputs moved by one ulp, oracle alone: %s" % (b, sorted(seen)), flush=True)
            if tie:
                bad -= 1
                print("      -> numerical tie at a termination check (not counted)", flush=True)
        if DUMP and ((r["status"] != ro["status"]) | (r["iters"] != ro["iters"])).any():
            # the instances that differ, with everything needed to run both sides again
            import json
            os.makedirs(DUMP, exist_ok=True)
            ix = np.where((r["status"] != ro["status"]) | (r["iters"] != ro["iters"]))[0]
            eo = eng.get_opts()
            np.savez_compressed(os.path.join(DUMP, "fuzz%d_trial%d_6dof.npz" % (SEED, trial)), index=ix, x=x0, ref=ref,
                                params=json.dumps(kw6), oracle_opts=json.dumps(o), engine_opts=json.dumps(e), rho=eng.rho,
                                rho_scale=np.array(eo.rho_scale[:]), phase_iters=np.array(eo.phase_iters[:]),
                                phase_scale=np.array([list(eo.phase_scale[0]), list(eo.phase_scale[1])]),
                                status_engine=r["status"], iters_engine=r["iters"], status_oracle=ro["status"], iters_oracle=ro["iters"],
                                X_engine=r["X"][ix], X_oracle=ro["X"][ix], residuals_engine=np.array(eng.last_residuals(B)))
        if rng.integers(0, 4) == 0 and N <= 40:
            # QP data of these parameters, bit for bit: P, A, l, u, and q / l / u of a few instances
            from trajectory_generation_amd import _capi as capi
            qp = O.build_qp(op)
            same = all(np.array_equal(eng.get(k), qp[n]) for k, n in ((capi.TG_MAT_P, "P"), (capi.TG_MAT_A, "A"), (capi.TG_VEC_L, "l"), (capi.TG_VEC_U, "u")))
            nb = min(B, 5)
            buf = eng.alloc(nb)
            buf["x0"][:, :nb] = torch.from_numpy(np.ascontiguousarray(x0[:nb].T)).cuda()
            buf["ref"][:, :nb] = torch.from_numpy(np.ascontiguousarray(ref[:nb].T)).cuda()
            q, l, u = [t.cpu().numpy() for t in eng.assemble(buf)]
            for b in range(nb):
                qo, lo, uo = O.qp6_vectors(op, x0[b], ref[b])
                same = same and np.array_equal(q[:, b], qo) and np.array_equal(l[:, b], lo) and np.array_equal(u[:, b], uo)
            bad += not same
            print("6dofq trial %3d QP data bit for bit: %s%s" % (trial, same, "" if same else "  <-- MISMATCH"), flush=True)
        if rng.integers(0, 3) == 0:
            # three warm-started solves in a row (the state and the reference move in between), a few
            # instances one at a time on the oracle's side
            nb = min(B, 8)
            eng.reset_warm()
            seq = [(x0[:nb], ref[:nb])]
            for k in range(2):
                seq.append((seq[-1][0] + rng.uniform(-0.05, 0.05, (nb, 6)), seq[-1][1] + rng.uniform(-0.03, 0.03, (nb, 1))))
            mark("6dofw trial %3d: engine" % trial)
            got = [eng.solve_host(a, b, warm=True) for a, b in seq]
            mark("6dofw trial %3d engine done, oracle" % trial)
            mism, worst = 0, 0.0
            for i in range(nb):
                m = O.Mpc6(op, O.fixed_rho_settings(eng.rho, warm_start=1, **o))
                for k, (a, b) in enumerate(seq):
                    res = m.loop(a[i], b[i])
                    mism += int(res["iters"] != got[k]["iters"][i]) + int(res["status"] != got[k]["status"][i])
                    if res["status"] == 1:
                        worst = max(worst, float(np.abs(res["X"] - got[k]["X"][i]).max()))
            flag = "" if mism == 0 and worst < 1e-8 else "  <-- MISMATCH"
            bad += bool(flag)
            print("6dofw trial %3d warm triple on %d instances: mism %d d %.1e%s" % (trial, nb, mism, worst, flag), flush=True)
        if rng.integers(0, 5) == 0 and N <= 30 and N >= 2:
            # closed loop, warm-started: solve, gated plant step, reference shift, for a few vehicles
            # and steps; one vehicle reports a speed outside its limits at one step (a failed solve:
            # it keeps following the plan it last received).  The same loop on the host with the oracle.
            from trajectory_generation_amd import _capi as capi
            nb, steps = min(B, 5), 7
            A, Bm = eng.get(capi.TG_MAT_AD), eng.get(capi.TG_MAT_BD)
            xs, rs = O.synth_mpc6(3000 * SEED + trial, N, dt, nb)
            fault = (int(rng.integers(1, steps)), int(rng.integers(0, nb)), 3 + int(rng.integers(0, 3)))
            eng.reset_warm()
            buf = eng.alloc(nb)
            buf["x0"][:, :nb] = torch.from_numpy(np.ascontiguousarray(xs.T)).cuda()
            buf["ref"][:, :nb] = torch.from_numpy(np.ascontiguousarray(rs.T)).cuda()
            hist, stat = [], []
            for k in range(steps):
                if k == fault[0]:
                    buf["x0"][fault[2], fault[1]] = 4.0 * kw6["max_vel"][fault[2] - 3]
                eng.solve(buf, warm=True)
                stat.append(buf["status"][:nb].cpu().numpy().copy())
                eng.plant_step(buf); eng.ref_shift(buf)
                hist.append(buf["x0"][:, :nb].T.cpu().numpy().copy())
            worst, mism = 0.0, 0
            for b in range(nb):
                m = O.Mpc6(op, O.fixed_rho_settings(eng.rho, warm_start=1, **o))
                x, r, plan = xs[b].copy(), rs[b].copy(), np.zeros(3 * N)
                for k in range(steps):
                    if k == fault[0] and b == fault[1]:
                        x[fault[2]] = 4.0 * kw6["max_vel"][fault[2] - 3]
                    res = m.loop(x, r)
                    mism += int(res["status"] != stat[k][b])
                    if res["status"] == 1:
                        plan = res["U"].copy()
                    else:
                        plan = np.concatenate([plan[3:], np.zeros(3)])
                    x = A @ x + Bm @ plan[:3]
                    rr = r.reshape(N + 1, 6).copy(); last = rr[N] + (rr[N] - rr[N - 1]); rr[:-1] = rr[1:]; rr[N] = last
                    r = rr.ravel()
                    worst = max(worst, float(np.abs(hist[k][b] - x).max()))
            flag = "" if mism == 0 and worst < 1e-6 else "  <-- MISMATCH"
            bad += bool(flag)
            print("6dofc trial %3d closed loop %d vehicles x %d steps, fault %s: status mism %d, state d %.1e%s" % (
                trial, nb, steps, fault, mism, worst, flag), flush=True)
    except KeyError:
        pass
    except Exception as ex:
        msg = str(ex)
        if "horizon too long" in msg and N > 52:
            print("6dof  trial %3d N %d declined: %s" % (trial, N, msg[:80]), flush=True)
        else:
            print("6dof  trial", trial, desc, "FAIL", msg[:200], flush=True); bad += 1
    # ---------------------------------------------------------------- 12-state chain
    N = int(rng.integers(1, 61)) if not big else int(rng.integers(1, 12))
    o, e = draw_opts(N)
    if big:
        e["grid_cap"] = 0
        o["max_iter"] = e["max_iter"] = min(o["max_iter"], 400)    # the oracle runs these thousands of instances on one core
    kw12 = dict(N=N, dt=dt, enable_control_smoothing=smooth,
                state_weight=tuple(float(10 ** v) for v in rng.uniform(2.5, 4, 3)),
                input_weight=tuple(float(10 ** v) for v in rng.uniform(1, 2.7, 3)),
                smooth_input_weight=tuple(float(10 ** v) for v in rng.uniform(1, 2.3, 3)),
                max_vel=(float(rng.uniform(5, 15)), float(rng.uniform(3, 10)), float(rng.uniform(1, 3))),
                max_accel=(float(rng.uniform(3, 10)), float(rng.uniform(5, 12)), float(rng.uniform(5, 12))),
                max_jerk=(float(rng.uniform(10, 40)), float(rng.uniform(10, 50)), float(rng.uniform(5, 20))))
    if WILD:
        kw12.update(dt=dt, state_weight=tuple(float(10 ** v) for v in rng.uniform(-1, 6, 3)),
                    input_weight=tuple(float(10 ** v) for v in rng.uniform(-2, 5, 3)),
                    smooth_input_weight=tuple(float(10 ** v) for v in rng.uniform(-2, 4, 3)),
                    max_vel=tuple(float(v) for v in 10 ** rng.uniform(-0.5, 2, 3)),
                    max_accel=tuple(float(v) for v in 10 ** rng.uniform(-0.5, 2, 3)),
                    max_jerk=tuple(float(v) for v in 10 ** rng.uniform(0, 3, 3)))
    desc = "N %2d B %2d dt %.2f sm %d %s" % (N, B, dt, smooth, {k: e[k] for k in ("eps_abs", "alpha", "max_iter", "check_termination", "check_fine", "lowrank", "grid_cap", "phase_iters", "overlap") if k in e})
    try:
        if ONLY and trial not in ONLY:
            raise KeyError("skipped")
        if ONLY:
            print("options", o, e, kw12, flush=True)
        eng = tg.BatchedMPC12(tg.mpc12_params(**kw12), tg.make_opts(**e))
        x, ref = O.synth_mpc12(2000 * SEED + trial, N, dt, B)
        ns = int(share * B)
        if ns:
            x[:ns], ref[:ns] = synth.stress("12state", x[:ns], ref[:ns], N)
        if rng.integers(0, 4) == 0:            # speeds outside the limits in one of the three stages
            for i in rng.choice(B, size=min(B, 3), replace=False):
                col, lim = [(1, kw12["max_vel"][0]), (4, kw12["max_vel"][0]), (7, kw12["max_vel"][1]), (10, kw12["max_vel"][2])][int(rng.integers(0, 4))]
                x[i, col] = float(rng.choice([-1, 1])) * lim * float(rng.uniform(1.05, 3.0))
        mark("12st  trial %3d %s: engine" % (trial, desc))
        r = eng.solve_host(x, ref)
        mark("12st  trial %3d engine done (iters max %d), oracle" % (trial, r["iters"].max()))
        O.match_engine(eng)
        s = {k: O.fixed_rho_settings(eng.rho(k), warm_start=0, **o) for k in ("xy", "z", "yaw")}
        ro = O.mpc12_batch(O.mpc12_params(**kw12), s, x, ref)
        before = bad
        okc = (ro["status"] == 1).all(1)
        report("12st ", trial, desc, r, ro, okc, 1e-7)
        if bad > before and okc.any() and (r["status"] == ro["status"]).all() and (r["iters"][:, :2] == ro["iters"][:, :2]).all():
            # Only the yaw columns apart?  yaw_ref = atan2(target - plan) (mpc_12state.cpp:544-548) of a vector
            # of next to no length turns rounding-level differences of the XY solutions into different yaw
            # problems.  Then the engine's yaw stage, given the oracle's XY solution, must reproduce the
            # oracle's yaw stage: if it does, the draw shows that sensitivity and nothing else.
            dX = np.abs(r["X"][okc] - ro["X"][okc]).reshape(int(okc.sum()), -1, 12)
            if dX[:, :, :9].max() < 1e-9 * max(1.0, np.abs(ro["X"][okc]).max()):
                buf = eng.alloc(B)
                buf["x0"][:, :B] = torch.from_numpy(np.ascontiguousarray(x.T)).cuda()
                buf["ref"][:, :B] = torch.from_numpy(np.ascontiguousarray(ref.T)).cuda()
                buf["X"][:, :B] = torch.from_numpy(np.ascontiguousarray(np.where(okc[:, None], ro["X"], 0.0).T)).cuda()
                eng.solve_stage("yaw", buf)
                torch.cuda.synchronize()
                Xy = buf["X"][:, :B].cpu().numpy().T
                st2, it2 = buf["status"][2, :B].cpu().numpy(), buf["iters"][2, :B].cpu().numpy()
                same = (st2[okc] == ro["status"][okc, 2]).all() and (it2[okc] == ro["iters"][okc, 2]).all()
                dy = float(np.abs(Xy[okc] - ro["X"][okc]).max() / max(1.0, np.abs(ro["X"][okc]).max()))
                print("      yaw stage on the oracle's XY solution: statuses and iterations equal %s, d %.1e" % (same, dy), flush=True)
                if same and dy < 1e-7:
                    bad -= 1
                    print("      -> sensitivity of the yaw reference to the last bits of the XY solution (not counted)", flush=True)
        if DUMP and (not os.environ.get("FUZZ_DUMP_TRIALS") or str(trial) in os.environ["FUZZ_DUMP_TRIALS"].split(",")):
            import json
            os.makedirs(DUMP, exist_ok=True)
            np.savez_compressed(os.path.join(DUMP, "fuzz%d_trial%d.npz" % (SEED, trial)), x=x, ref=ref,
                                params=json.dumps(kw12), oracle_opts=json.dumps(o), engine_opts=json.dumps(e),
                                X_engine=r["X"], status_engine=r["status"], iters_engine=r["iters"],
                                X_oracle=ro["X"], status_oracle=ro["status"], iters_oracle=ro["iters"])
        if rng.integers(0, 4) == 0 and N <= 30:
            from trajectory_generation_amd import _capi as capi
            op12 = O.mpc12_params(**kw12)
            same = True
            for stage, fam in (("xy", O.FAM_XY), ("z", O.FAM_Z), ("yaw", O.FAM_YAW)):
                qp = O.build_qp(op12, fam)
                same = same and all(np.array_equal(eng.get(stage, k), qp[n]) for k, n in ((capi.TG_MAT_P, "P"), (capi.TG_MAT_A, "A"), (capi.TG_VEC_L, "l"), (capi.TG_VEC_U, "u")))
            bad += not same
            print("12stq trial %3d stage matrices bit for bit: %s%s" % (trial, same, "" if same else "  <-- MISMATCH"), flush=True)
        if rng.integers(0, 4) == 0:
            # the chain warm-started (an option the reference does not use): two solves in a row
            nb = min(B, 5)
            eng.reset_warm()
            x1 = x[:nb] + rng.uniform(-0.03, 0.03, (nb, 12)); r1 = ref[:nb] + rng.uniform(-0.02, 0.02, (nb, 1))
            mark("12stw trial %3d: engine" % trial)
            got = [eng.solve_host(x[:nb], ref[:nb], warm=True), eng.solve_host(x1, r1, warm=True)]
            mark("12stw trial %3d engine done, oracle" % trial)
            mism, worst = 0, 0.0
            for i in range(nb):
                m = O.Mpc12(O.mpc12_params(**kw12), {k: O.fixed_rho_settings(eng.rho(k), warm_start=1, **o) for k in ("xy", "z", "yaw")})
                for k, (a, b) in enumerate(((x[:nb], ref[:nb]), (x1, r1))):
                    res = m.loop(a[i], b[i])
                    mism += int((res["iters"] != got[k]["iters"][i]).sum()) + int((res["status"] != got[k]["status"][i]).sum())
                    if (res["status"] == 1).all():
                        worst = max(worst, float(np.abs(res["X"] - got[k]["X"][i]).max() / max(1.0, np.abs(res["X"]).max())))
            flag = "" if mism == 0 and worst < 1e-7 else "  <-- MISMATCH"
            bad += bool(flag)
            print("12stw trial %3d warm pair on %d instances: mism %d d %.1e%s" % (trial, nb, mism, worst, flag), flush=True)
    except KeyError:
        pass
    except Exception as ex:
        print("12st  trial", trial, desc, "FAIL", str(ex)[:200], flush=True); bad += 1
print("problems:", bad)
sys.exit(1 if bad else 0)
