"""One real 6-DOF batch (eps 1e-3) under a chosen slot schedule, for rocprofv3 --pmc FETCH_SIZE /
WRITE_SIZE passes that separate the traffic of the slot refill from the rest (development aid).
usage: traffic_probe2.py N B grid_cap hard_first"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import trajectory_generation_amd as tg
from trajectory_generation_amd import synth
N, B, cap, hf = (int(v) for v in sys.argv[1:5])
eng = tg.BatchedMPC6(tg.mpc6_params(N=N), tg.make_opts(grid_cap=cap, hard_first=hf))
x0, ref = synth.mpc6(1, N, 0.1, B)
buf = eng.alloc(B)
buf["x0"][:, :B] = torch.from_numpy(np.ascontiguousarray(x0.T)).cuda()
buf["ref"][:, :B] = torch.from_numpy(np.ascontiguousarray(ref.T)).cuda()
for _ in range(3):
    eng.solve(buf)
torch.cuda.synchronize()
print("done", int(buf["iters"][:B].max()), float(buf["iters"][:B].double().mean()))
