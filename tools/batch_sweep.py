"""Batch sizes around every boundary (tile of 16, one wave of workgroups, queue on/off, ordering
on/off), both families: a sample of each batch against the CPU oracle (development aid; GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import trajectory_generation_amd as tg
from oracle import oracle as O
O.build()
bad = 0
for fam, N in (("6dof", 20), ("6dof", 50), ("12state", 20)):
    eng = tg.BatchedMPC6(tg.mpc6_params(N=N)) if fam == "6dof" else tg.BatchedMPC12(tg.mpc12_params(N=N))
    O.match_engine(eng)
    for B in (1, 2, 15, 16, 17, 31, 255, 4095, 4096, 4097, 4111, 4113, 8191, 20001):
        if fam == "6dof":
            x, ref = O.synth_mpc6(31, N, 0.1, B); r = eng.solve_host(x, ref)
            idx = np.unique(np.concatenate([np.arange(min(B, 8)), np.arange(max(B - 8, 0), B), np.arange(0, B, max(B // 16, 1))]))
            ro = O.mpc6_batch(O.mpc6_params(N=N), O.fixed_rho_settings(eng.rho, warm_start=0), x[idx], ref[idx])
            ok = ro["status"] == 1
        else:
            x, ref = O.synth_mpc12(32, N, 0.1, B); r = eng.solve_host(x, ref)
            idx = np.unique(np.concatenate([np.arange(min(B, 8)), np.arange(max(B - 8, 0), B), np.arange(0, B, max(B // 16, 1))]))
            s = {k: O.fixed_rho_settings(eng.rho(k), warm_start=0) for k in ("xy", "z", "yaw")}
            ro = O.mpc12_batch(O.mpc12_params(N=N), s, x[idx], ref[idx])
            ok = (ro["status"] == 1).all(1)
        mism = int((r["status"][idx] != ro["status"]).sum() + (r["iters"][idx] != ro["iters"]).sum())
        d = float(np.abs(r["X"][idx][ok] - ro["X"][ok]).max()) if ok.any() else 0.0
        solved = float((r["status"] == 1).mean())
        flag = "" if mism == 0 and d < 1e-8 else "  <-- MISMATCH"
        bad += bool(flag)
        print("%s N %d B %6d: sample %3d mism %d d %.1e solved %.4f%s" % (fam, N, B, len(idx), mism, d, solved, flag), flush=True)
print("problems:", bad)
sys.exit(1 if bad else 0)
