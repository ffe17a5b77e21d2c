"""Random model parameters (dt, weights, limits, smoothing) x horizons, both families: the solve
must launch and agree with the CPU oracle in the engine's mode (development aid; GPU box).
usage: param_sweep.py [seed [trials [long]]]   (default 7 30; "long": horizons to 60 as well)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import trajectory_generation_amd as tg
from oracle import oracle as O
O.build()
SEED = int(sys.argv[1]) if len(sys.argv) > 1 else 7
TRIALS = int(sys.argv[2]) if len(sys.argv) > 2 else 30
HORIZONS = [3, 5, 10, 16, 20, 27, 35, 50] + ([43, 52, 57, 60] if len(sys.argv) > 3 else [])
rng = np.random.default_rng(SEED)
B = 24
bad = 0
for trial in range(TRIALS):
    N = int(rng.choice(HORIZONS))
    dt = float(rng.choice([0.05, 0.1, 0.2]))
    smooth = bool(rng.integers(0, 2))
    kw6 = dict(N=N, dt=dt, state_weight=float(10 ** rng.uniform(2, 4.3)), input_weight=float(10 ** rng.uniform(1, 3)),
               smooth_input_weight=float(10 ** rng.uniform(0, 2)), enable_control_smoothing=smooth,
               min_altitude=float(rng.uniform(-2, 1)), max_vel=tuple(float(v) for v in rng.uniform(3, 15, 3)),
               max_accel=tuple(float(v) for v in rng.uniform(2, 10, 3)))
    try:
        e = tg.BatchedMPC6(tg.mpc6_params(**kw6))
        x0, ref = O.synth_mpc6(100 + trial, N, dt, B)
        r = e.solve_host(x0, ref)
        O.match_engine(e)
        ro = O.mpc6_batch(O.mpc6_params(**kw6), O.fixed_rho_settings(e.rho, warm_start=0), x0, ref)
        ok = ro["status"] == 1
        mism = int((r["status"] != ro["status"]).sum() + (r["iters"] != ro["iters"]).sum())
        d = float(np.abs(r["X"][ok] - ro["X"][ok]).max()) if ok.any() else 0.0
        flag = "" if mism == 0 and d < 1e-8 else "  <-- MISMATCH"
        bad += bool(flag)
        print("6dof  trial %2d N %2d dt %.2f smooth %d rho %g info %s solved %d/%d itmax %d mism %d d %.1e%s" % (
            trial, N, dt, smooth, e.rho, e.kernel_info(), ok.sum(), B, ro["iters"].max(), mism, d, flag), flush=True)
        for i in np.nonzero((r["status"] != ro["status"]) | (r["iters"] != ro["iters"]))[0]:
            print("      instance %d: engine status %d after %d, oracle status %d after %d" % (i, r["status"][i], r["iters"][i], ro["status"][i], ro["iters"][i]))
    except Exception as ex:
        print("6dof  trial", trial, N, kw6, "FAIL", ex, flush=True); bad += 1
    kw12 = dict(N=N, dt=dt, enable_control_smoothing=smooth,
                state_weight=tuple(float(10 ** v) for v in rng.uniform(2.5, 4, 3)),
                input_weight=tuple(float(10 ** v) for v in rng.uniform(1, 2.7, 3)),
                smooth_input_weight=tuple(float(10 ** v) for v in rng.uniform(1, 2.3, 3)),
                max_vel=(float(rng.uniform(5, 15)), float(rng.uniform(3, 10)), float(rng.uniform(1, 3))),
                max_accel=(float(rng.uniform(3, 10)), float(rng.uniform(5, 12)), float(rng.uniform(5, 12))),
                max_jerk=(float(rng.uniform(10, 40)), float(rng.uniform(10, 50)), float(rng.uniform(5, 20))))
    try:
        e = tg.BatchedMPC12(tg.mpc12_params(**kw12))
        x, ref = O.synth_mpc12(200 + trial, N, dt, B)
        r = e.solve_host(x, ref)
        O.match_engine(e)
        s = {k: O.fixed_rho_settings(e.rho(k), warm_start=0) for k in ("xy", "z", "yaw")}
        ro = O.mpc12_batch(O.mpc12_params(**kw12), s, x, ref)
        ok = (ro["status"] == 1).all(1)
        mism = int((r["status"] != ro["status"]).sum() + (r["iters"] != ro["iters"]).sum())
        d = float(np.abs(r["X"][ok] - ro["X"][ok]).max()) if ok.any() else 0.0
        flag = "" if mism == 0 and d < 1e-7 else "  <-- MISMATCH"
        bad += bool(flag)
        print("12st  trial %2d N %2d dt %.2f smooth %d rho %s info %s solved %d/%d itmax %d mism %d d %.1e%s" % (
            trial, N, dt, smooth, [e.rho(k) for k in ("xy", "z", "yaw")], [e.kernel_info(k)[0] for k in ("z", "xy", "yaw")],
            ok.sum(), B, ro["iters"].max(), mism, d, flag), flush=True)
        for i, k in zip(*np.nonzero((r["status"] != ro["status"]) | (r["iters"] != ro["iters"]))):
            print("      instance %d stage %d: engine status %d after %d, oracle status %d after %d" % (
                i, k, r["status"][i, k], r["iters"][i, k], ro["status"][i, k], ro["iters"][i, k]))
    except Exception as ex:
        print("12st  trial", trial, N, "FAIL", ex, flush=True); bad += 1
print("problems:", bad)
sys.exit(1 if bad else 0)
