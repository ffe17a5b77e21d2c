"""Closed-loop receding-horizon roll-out on the device: every step solves the batch (warm-started
ADMM), applies the first control to the plant x+ = A x + B u(0) -- the rule of the reference's test
mains (test_mpc_6dof.cpp:115-122, test_mpc_12state.cpp:193-197) -- and moves the reference one
step along.  Nothing returns to the host between steps unless statistics are asked for.

A vehicle whose solve fails at a step gets no new plan -- the reference's node publishes nothing then
(mpc_ros.cpp:428-432) -- and keeps following the last plan it received (plant_step moves its U one
step up); the per-step statistics report how many did.

Warm start for the 12-state model deviates from the reference, which cold-starts its three solvers
(mpc_12state.cpp:1025,1045,1065); pass warm=False to follow it."""
import numpy as np


def closed_loop(eng, buf, steps, warm=True, record=False, stats_every=0, stream=None):
    """Runs `steps` closed-loop steps in place on buf (x0 and ref advance).  Returns a dict:
    states [steps+1, B, nx] (record=True), and per sampled step the mean / max iterations, the
    fraction of instances whose every stage solved, the fraction that failed (any stage primal
    infeasible or out of iterations; later stages of such an instance are not attempted) and the
    fraction with a stage that ran into max_iter."""
    import torch
    B = buf["B"]
    out = {"iters_mean": [], "iters_max": [], "solved_frac": [], "failed_frac": [], "max_iter_hit_frac": [], "step": []}
    traj = []
    if record:
        traj.append(buf["x0"][:, :B].T.cpu().numpy().copy())
    eng.reset_warm()
    for k in range(steps):
        eng.solve(buf, warm=warm, stream=stream)
        if stats_every and (k % stats_every == 0 or k == steps - 1):
            it = buf["iters"].reshape(-1, buf["ld"])[:, :B]
            st = buf["status"].reshape(-1, buf["ld"])[:, :B]
            out["step"].append(k)
            out["iters_mean"].append(float(it.double().mean().item()))
            out["iters_max"].append(int(it.max().item()))
            out["solved_frac"].append(float((st == 1).all(0).double().mean().item()))
            out["failed_frac"].append(float((st != 1).any(0).double().mean().item()))
            out["max_iter_hit_frac"].append(float(((st == -2) | (st == 2) | (st == 3)).any(0).double().mean().item()))
        eng.plant_step(buf, stream=stream)
        eng.ref_shift(buf, stream=stream)
        if record:
            traj.append(buf["x0"][:, :B].T.cpu().numpy().copy())
    torch.cuda.synchronize()
    if record:
        out["states"] = np.stack(traj)
    return out
