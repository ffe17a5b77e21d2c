"""Closed-loop roll-out statistics per step (development aid): rollout_stats.py [6dof|12state] N B steps"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import trajectory_generation_amd as tg
from trajectory_generation_amd import synth
from trajectory_generation_amd.rollout import closed_loop

model, N, B, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
warm = (sys.argv[5] != "cold") if len(sys.argv) > 5 else True
if model == "6dof":
    eng = tg.BatchedMPC6(tg.mpc6_params(N=N)); x0, ref = synth.mpc6(0x4D5043 + 5, N, 0.1, B)
else:
    eng = tg.BatchedMPC12(tg.mpc12_params(N=N)); x0, ref = synth.mpc12(0x4D5043 + 5, N, 0.1, B)
buf = eng.alloc(B)
buf["x0"][:, :B] = torch.from_numpy(np.ascontiguousarray(x0.T)).cuda()
buf["ref"][:, :B] = torch.from_numpy(np.ascontiguousarray(ref.T)).cuda()
t = time.perf_counter()
out = closed_loop(eng, buf, steps, warm=warm, stats_every=1)
dt = time.perf_counter() - t
print("total %.3f s, %.3f ms/step (stats included)" % (dt, dt / steps * 1e3))
for k, im, ix, sf in zip(out["step"], out["iters_mean"], out["iters_max"], out["solved_frac"]):
    if k < 5 or ix > 100 or sf < 1 or k % 50 == 0:
        print("step %4d iters mean %.1f max %d solved %.5f" % (k, im, ix, sf))
x = buf["x0"][:, :B].T.cpu().numpy(); r = buf["ref"][:, :B].T.cpu().numpy()
nx = x.shape[1]
pos = [0, 1, 2] if model == "6dof" else [0, 3, 6]
print("final position error to the reference: mean %.3f max %.3f" % (np.abs(x[:, pos] - r[:, pos]).mean(), np.abs(x[:, pos] - r[:, pos]).max()))
