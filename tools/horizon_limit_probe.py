"""Horizons past 50: which ones each family still solves (and whether the results match the CPU
oracle in the engine's mode), and what the engine says where it declines (development aid; GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import trajectory_generation_amd as tg
from oracle import oracle as O
O.build()
B = 37
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (51, 100)
for N in range(lo, hi + 1):
    t0 = time.time()
    line = "N %3d " % N
    try:
        eng = tg.BatchedMPC6(tg.mpc6_params(N=N))
        x0, ref = O.synth_mpc6(11, N, 0.1, B)
        r = eng.solve_host(x0, ref)
        O.match_engine(eng)
        ro = O.mpc6_batch(O.mpc6_params(N=N), O.fixed_rho_settings(eng.rho, warm_start=0), x0, ref)
        ok = ro["status"] == 1
        e6 = (r["status"] != ro["status"]).sum() + (r["iters"] != ro["iters"]).sum()
        d6 = np.abs(r["X"][ok] - ro["X"][ok]).max() if ok.any() else 0.0
        line += "6dof %s mism %d d %.1e" % (eng.kernel_info(), e6, d6)
    except Exception as ex:
        line += "6dof DECLINED: %s" % str(ex)[:90]
    try:
        e12 = tg.BatchedMPC12(tg.mpc12_params(N=N))
        x12, ref12 = O.synth_mpc12(12, N, 0.1, B)
        r12 = e12.solve_host(x12, ref12)
        O.match_engine(e12)
        s = dict(xy=O.fixed_rho_settings(e12.rho("xy"), warm_start=0), z=O.fixed_rho_settings(e12.rho("z"), warm_start=0),
                 yaw=O.fixed_rho_settings(e12.rho("yaw"), warm_start=0))
        ro12 = O.mpc12_batch(O.mpc12_params(N=N), s, x12, ref12)
        st, it = r12["status"], r12["iters"]
        e12n = (st != ro12["status"]).sum() + (it != ro12["iters"]).sum()
        ok12 = (ro12["status"] == 1).all(1)
        d12 = np.abs(r12["X"][ok12] - ro12["X"][ok12]).max() if ok12.any() else 0.0
        line += " | 12st %s mism %d d %.1e" % ([e12.kernel_info(k) for k in ("z", "xy", "yaw")], e12n, d12)
    except Exception as ex:
        line += " | 12st DECLINED: %s" % str(ex)[:90]
    print(line + "  (%.1fs)" % (time.time() - t0), flush=True)
