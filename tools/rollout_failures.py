"""Closed-loop 6-DOF roll-out that records the problems whose solve first fails (development aid).
usage: rollout_failures.py [steps] -> gpurun_out/rollout_failures.npz (x0, ref, status, step, vehicle)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import trajectory_generation_amd as tg
from trajectory_generation_amd import synth
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
N, B = 20, 8192
eng = tg.BatchedMPC6(tg.mpc6_params(N=N), tg.make_opts(max_iter=1000))
x0, ref = synth.mpc6(0x4D5043 + 5, N, 0.1, B)
buf = eng.alloc(B)
buf["x0"][:, :B] = torch.from_numpy(np.ascontiguousarray(x0.T)).cuda()
buf["ref"][:, :B] = torch.from_numpy(np.ascontiguousarray(ref.T)).cuda()
seen = set(); rec = []
for s in range(steps):
    eng.solve(buf, warm=True)
    st = buf["status"][:B].cpu().numpy()
    bad = np.where(st != 1)[0]
    for v in bad:
        if v not in seen and len(rec) < 64:
            seen.add(v)
            rec.append((s, v, st[v], buf["x0"][:, v].cpu().numpy().copy(), buf["ref"][:, v].cpu().numpy().copy(),
                        int(buf["iters"][v])))
    eng.plant_step(buf); eng.ref_shift(buf)
print("failing vehicles", len(seen), "recorded", len(rec))
os.makedirs("gpurun_out", exist_ok=True)
np.savez("gpurun_out/rollout_failures.npz", step=np.array([r[0] for r in rec]), vehicle=np.array([r[1] for r in rec]),
         status=np.array([r[2] for r in rec]), x0=np.array([r[3] for r in rec]), ref=np.array([r[4] for r in rec]),
         iters=np.array([r[5] for r in rec]))
for r in rec[:8]:
    print("step", r[0], "vehicle", r[1], "status", r[2], "iters", r[5], "x0", np.round(r[3], 5))
