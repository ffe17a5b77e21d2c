"""Every horizon 1..50, both families, a ragged batch: statuses and iteration counts against the
CPU oracle in the engine's mode, solutions to 1e-9 (development aid; run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import trajectory_generation_amd as tg
from oracle import oracle as O
O.build()
B = 37
bad = 0
for N in range(1, 51):
    t0 = time.time()
    eng = tg.BatchedMPC6(tg.mpc6_params(N=N))
    x0, ref = O.synth_mpc6(11, N, 0.1, B)
    r = eng.solve_host(x0, ref)
    O.match_engine(eng)
    ro = O.mpc6_batch(O.mpc6_params(N=N), O.fixed_rho_settings(eng.rho, warm_start=0), x0, ref)
    ok = ro["status"] == 1
    e6 = (r["status"] != ro["status"]).sum() + (r["iters"] != ro["iters"]).sum()
    d6 = np.abs(r["X"][ok] - ro["X"][ok]).max() if ok.any() else 0.0
    e12 = tg.BatchedMPC12(tg.mpc12_params(N=N))
    x12, ref12 = O.synth_mpc12(12, N, 0.1, B)
    r12 = e12.solve_host(x12, ref12)
    O.match_engine(e12)
    s = dict(xy=O.fixed_rho_settings(e12.rho("xy"), warm_start=0), z=O.fixed_rho_settings(e12.rho("z"), warm_start=0),
             yaw=O.fixed_rho_settings(e12.rho("yaw"), warm_start=0))
    ro12 = O.mpc12_batch(O.mpc12_params(N=N), s, x12, ref12)
    st, it = r12["status"], r12["iters"]           # both [B][3], stages z, xy, yaw
    e12n = (st != ro12["status"]).sum() + (it != ro12["iters"]).sum()
    ok12 = (ro12["status"] == 1).all(1)
    d12 = np.abs(r12["X"][ok12] - ro12["X"][ok12]).max() if ok12.any() else 0.0
    flag = "" if (e6 == 0 and e12n == 0 and d6 < 1e-9 and d12 < 1e-8) else "  <-- MISMATCH"
    bad += bool(flag)
    print("N %2d  6dof %s mism %d d %.1e | 12st %s mism %d d %.1e  (%.1fs)%s" % (
        N, eng.kernel_info(), e6, d6, [e12.kernel_info(k) for k in ("z", "xy", "yaw")], e12n, d12, time.time() - t0, flag), flush=True)
print("mismatching horizons:", bad)
sys.exit(1 if bad else 0)
