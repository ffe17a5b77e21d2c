"""Launch-coverage sweep (development aid; GPU box): every horizon with input smoothing on/off,
operator form forced dense / low-rank, a tiny refilled grid: each solve must launch and agree with
the default configuration's statuses (same algorithm, same iteration counts)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import trajectory_generation_amd as tg
from trajectory_generation_amd import synth
B = 70
bad = 0
for N in range(1, 51):
    for fam in ("6dof", "12state"):
        for smooth in (False, True):
            base = None
            for name, kw in (("default", {}), ("dense", dict(lowrank=-1)), ("lowrank", dict(lowrank=1)), ("cap2", dict(grid_cap=2)),
                             ("cap2_batch_order", dict(grid_cap=2, hard_first=-1))):
                try:
                    if fam == "6dof":
                        e = tg.BatchedMPC6(tg.mpc6_params(N=N, enable_control_smoothing=smooth), tg.make_opts(**kw))
                        x, ref = synth.mpc6(21, N, 0.1, B)
                    else:
                        e = tg.BatchedMPC12(tg.mpc12_params(N=N, enable_control_smoothing=smooth), tg.make_opts(**kw))
                        x, ref = synth.mpc12(22, N, 0.1, B)
                    r = e.solve_host(x, ref)
                except Exception as ex:
                    print("N %d %s smooth %d %s: FAIL %s" % (N, fam, smooth, name, ex), flush=True); bad += 1; continue
                if base is None:
                    base = r
                elif not (np.array_equal(base["status"], r["status"]) and np.array_equal(base["iters"], r["iters"])):
                    d = int((base["iters"] != r["iters"]).sum())
                    print("N %d %s smooth %d %s: %d iteration counts differ from default" % (N, fam, smooth, name, d), flush=True); bad += 1
    print("N", N, "done", flush=True)
print("problems:", bad)
sys.exit(1 if bad else 0)
